// stream_tail.h — the tail that aukit.stream.qoa (aukit.lua:3312-3330) and aukit.stream.flac (aukit.lua:3166-3183) share: a decoded
// table (rows in HBM) → `if x % 1 == 0 then s = t[x] else s = interp(t, x)` → recursive one-pole low-pass `s = ls + lp_alpha * (s - ls); ls = s`
// → chunk sample.  One launch (k_iir_tail, stream_tail.hip) instead of the two passes through a scratch of doubles of round 2.
#pragma once
#include "common.h"

namespace aukit {

struct TailJob {
    unsigned long long src_off;    // element offset of table index 1 (channel 0) in the rows
    unsigned long long last_off;   // element offset of table index 0's value (the history `last[2]`); ~0: 0
    unsigned long long m1_off;     // element offset of table index -1's value (`last[1]`); ~0: 0
    unsigned long long out_off;    // element offset of output 0 (channel 0)
    unsigned src_cstride, last_cstride, out_cstride;  // element strides between the channels of a mixed job (TAIL_QOA with mix)
    int n, nout;                   // #table, outputs
    int pad;
};
static_assert(sizeof(TailJob) == 56, "TailJob layout");

enum { TAIL_QOA = 0, TAIL_FLAC = 1 };
enum { TAIL_ROWS_I8 = 0, TAIL_ROWS_I32 = 1, TAIL_ROWS_F64 = 2 };

// kind: TAIL_QOA — interpolated samples clamped to [-128, 127] (:3323), ls seeded with the raw history sample (:3316), outputs stored as they
// are, or (mix_channels > 1) the mean `(0 + s_1 + ... + s_C) / C` of the job's channels (:3326-3329);
// TAIL_FLAC — rows scaled by 1 / full (:505), ls seeded with `last[2] / (last[2] < 0 and 128 or 127)` (:3172), outputs
// `clamp(s * (s < 0 and 128 or 127), -128, 127)` (:3181).  Returns false (nothing launched) when the shape is not served (very low sample
// rates whose filter memory is longer than a tile's warm-up): the caller keeps its two-pass path.
bool iir_tail_try(aukit_ctx *ctx, int kind, int rows_kind, const void *rows, double full, const std::vector<TailJob> &jobs, int mix_channels, double rate,
                  int interp, int dtype, void *out, uint64_t algorithmic_bytes, const char *name, int *rc);
// the same with the jobs already on the device (written by a kernel: stream.flac's come from the frame records the decoder left there);
// max_nout / sum_nout: the largest and the total output count of the jobs.  iir_tail_served() tells beforehand whether a shape is taken.
bool iir_tail_served(aukit_ctx *ctx, int kind, int rows_kind, int mix_channels, double rate, double full, int interp, int dtype, uint64_t max_nout);
bool iir_tail_try_dev(aukit_ctx *ctx, int kind, int rows_kind, const void *rows, double full, const TailJob *d_jobs, size_t njobs, uint64_t max_nout, uint64_t sum_nout,
                      int mix_channels, double rate, int interp, int dtype, void *out, uint64_t algorithmic_bytes, const char *name, int *rc);

// the same tail for stream.qoa with F32 storage on the tile chain of k_rs_onepole (flac_tail.hip): state carried from tile to tile instead of warmed up per tile
bool rs_onepole_jobs_try(aukit_ctx *ctx, const void *rows_i8, const std::vector<TailJob> &jobs, int mix_channels, double rate, int interp, double lp_alpha, float *out,
                         uint64_t algorithmic_bytes, const char *name, int *rc);
bool rs_onepole_jobs_try_dev(aukit_ctx *ctx, const void *rows_i32, double full, const TailJob *d_jobs, size_t njobs, double rate, int interp, double lp_alpha, float *out,
                             uint64_t algorithmic_bytes, const char *name, int *rc);

}  // namespace aukit
