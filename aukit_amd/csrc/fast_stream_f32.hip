// fast_stream_f32.hip — aukit.stream.pcm on f32 sample rows with AUKIT_F32 output: the stream.pcm wave kernel of fast_stream.hip on the rows
// k_pcm_unpack (api_resample.hip) makes of any PCM format other than 16-bit little-endian mono / stereo — 8-bit unsigned at 48 kHz is what most
// ComputerCraft audio is kept in — one row per output channel (the `mono` mix is made while unpacking, in the reference's order, :2368).
#include "fast_stream_body.h"

namespace aukit {

int launch_fast_wave_stream_f32(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    if (interp == AUKIT_INTERP_LINEAR) return launch_nv_stream<SRC_AUDIO_F32, AUKIT_INTERP_LINEAR>(ctx, nv, P, F, lds, grid);
    return launch_nv_stream<SRC_AUDIO_F32, AUKIT_INTERP_CUBIC>(ctx, nv, P, F, lds, grid);
}

}  // namespace aukit
