// fast_stream_u8.hip — aukit.stream.pcm on 8-bit mono strings (signed or unsigned) with AUKIT_F32 output: the stream.pcm wave kernel of
// fast_stream.hip reading the bytes themselves (1 byte in, 4 out per sample; through f32 rows it is 1 + 4 + 4 in and 4 out).  8-bit unsigned
// mono at 48 kHz is the classic pre-converted ComputerCraft speaker file.
#include "fast_stream_body.h"

namespace aukit {

int launch_fast_wave_stream_u8(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    if (interp == AUKIT_INTERP_LINEAR) return launch_nv_stream<SRC_PCM8_MONO, AUKIT_INTERP_LINEAR>(ctx, nv, P, F, lds, grid);
    return launch_nv_stream<SRC_PCM8_MONO, AUKIT_INTERP_CUBIC>(ctx, nv, P, F, lds, grid);
}

}  // namespace aukit
