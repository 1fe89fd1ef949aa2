// fast_stream.hip — the wave-private f32 resampler (fast2.hip) with the stream.pcm epilogue (aukit.lua:2397-2403), for
// aukit.stream.pcm on s16le mono input with AUKIT_F32 output:
//   s = raw interpolated sample (not clamped); ns = ls + alpha * (s - ls) with ls the RAW previous sample of the same iterator
//   call (0 for its first output, Q2); output clamp(ns * (ns < 0 and 128 or 127), -128, 127).
// The previous sample of a row's first lane comes from lane 63 of the row before (one readlane); that of a tile's first output
// is evaluated once more from the window, which therefore reaches one tap further left than the Audio:resample kernel's.
// Its own translation unit: see fast_wave_dev.h.
#include "fast_stream_body.h"

namespace aukit {

int launch_fast_wave_stream(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    if (interp == AUKIT_INTERP_LINEAR) return launch_nv_stream<SRC_PCM_S16LE_MONO, AUKIT_INTERP_LINEAR>(ctx, nv, P, F, lds, grid);
    return launch_nv_stream<SRC_PCM_S16LE_MONO, AUKIT_INTERP_CUBIC>(ctx, nv, P, F, lds, grid);
}

}  // namespace aukit
