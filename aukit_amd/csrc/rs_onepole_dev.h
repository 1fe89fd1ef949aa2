// rs_onepole_dev.h — what the two tile-chain kernels of the "resample owed, then a one-pole filter" pass share: k_rs_onepole (flac_tail.hip: any
// ratio, tiles of 512 outputs) and k_rsp (rs_periodic.hip: ratios whose phases repeat every 320 outputs — 44.1 / 22.05 kHz -> 48 kHz —, weights and
// tap offsets in registers).  Parameters, and the DPP moves of the wave scan.
#pragma once
#include "resample.h"
#include "flac_dev.h"
#include "stream_tail.h"

namespace aukit {

struct RsOnepoleParams {
    const void *rows;                              // int32 (FLAC), int16 (IMA / MS-ADPCM / QOA) or int8 (DFPWM) rows: the template's S
    const unsigned long long *row_off, *row_len;   // per (stream, channel): element offset / samples of the decoded row
    const unsigned long long *a_meta;              // the audio's len[n], row_off[n], row_stride[n]
    float *out;
    unsigned long long *rowmax;
    unsigned n;
    int C, cap;
    unsigned fa, fb, fmagic, dq256, dr256;
    float inv_b, scale, scale_neg;                 // sample = (float)v * (v < 0 ? scale_neg : scale): the fast wave kernels' conversion (fast_wave_dev.h)
    double coef;
    const float *wg;   // cubic: the four tap weights of each of the fb output phases (null: the Horner form on fx = rem / fb)
    // round 4: the rows frame by frame where the fused FLAC decoder left them (null: contiguous rows at row_off).  Every stream's frames but its
    // last have bs0[stream] samples, and a tile's window is shorter than that: it lies in one frame or two consecutive ones
    const FrameRec *frames;
    const unsigned long long *fbase;
    const int *bs0;
    // round 4: a row is cut into `segs` runs of tiles, a wave each.  A run starts `warm` tiles early from a zero state and stores nothing there: the
    // recurrence forgets its state at m per output (m^(512 warm) < 2^-40), so the run's own outputs are those of the whole row's chain to far
    // below an f32 ulp — and a row is no longer ONE serial chain of 938 tiles on a chip that can run four times as many chains as config 5 has rows
    int segs, warm;
    int novec;   // AUKIT_RS_NOVEC=1: every window element by element (the first cut; A/B)
    int fr_mul;  // frame-by-frame rows: a record's offset counts int32 slots — 2 when the frames hold int16 finals (k_flac_decode<..., O16>), else 1
    // NW = 2 (round 4, late): a workgroup = the two channels of one stream, a wave each; what leaves is their MEAN (`Audio:mono` :682-687 behind
    // the filter), `out` / `a_meta` describe the MONO audio, rowmax2[stream] receives the larger of the two channels' maxima (what a
    // non-independent effects.normalize in between divides by, :3439-3444); wave_lds = floats of LDS per wave
    unsigned long long *rowmax2;
    int wave_lds;
    // JOBS (round 4, last): a workgroup's work item is not a row of an audio but a JOB of stream.qoa's tail (stream_tail.h: one iterator call's chunk of one
    // channel — or of all its channels, NW = 2, whose mean is stored): its own table (n samples at src_off, the history sample `last[2]` as table index 0,
    // :3255), its own outputs, the low-pass seeded with the history sample (:3316), interpolated samples clamped to [clo, chi] (:3323)
    const TailJob *jobs;
    unsigned njobs;
    int epi;   // 1: stream.flac's seed and store scaling (above)
    float clo, chi;
};

// A double moved between lanes by DPP (two v_mov_b32 with a DPP modifier: VALU latency) instead of ds_bpermute (two LDS round trips): lanes without a
// source lane — and rows outside ROW_MASK — get 0.  CTRL: row_shr:n = 0x110 + n, row_bcast15 = 0x142, row_bcast31 = 0x143, wave_shr:1 = 0x138.
template <int CTRL, int ROW_MASK = 0xF>
AUKIT_DEV double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <int CTRL, int ROW_MASK = 0xF> AUKIT_DEV double dpp_rt(double v) { return dpp_f64<CTRL, ROW_MASK>(v); }
template <int CTRL, int ROW_MASK = 0xF> AUKIT_DEV float dpp_rt(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true)); }
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
typedef unsigned u32x2g __attribute__((ext_vector_type(2)));

// rs_periodic.hip: the same pass for S = int16 rows, cubic, a ratio a / b < 1 with 320 a = 0 (mod b) whose tap pattern is one of the built ones.
// false: not this shape (the caller launches k_rs_onepole).  `P` as lazy_onepole_try fills it (cap, wave_lds, segs, warm are set here)
bool rsp_try(aukit_ctx *ctx, RsOnepoleParams &P, bool highpass, bool r32, int NW, size_t rows, uint64_t min_out_len, int min_frame, int *rc);
bool rsp_jobs_try(aukit_ctx *ctx, const RsOnepoleParams &P, bool r32, int NW, unsigned grid, int *rc);   // stream.qoa's tail (long jobs of int8 rows at 44.1 kHz)

}  // namespace aukit
