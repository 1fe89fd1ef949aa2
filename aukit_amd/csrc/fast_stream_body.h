// fast_stream_body.h — the stream.pcm wave kernel (see fast_stream.hip), a template over its source so that each source gets its own
// translation unit (fast_stream.hip: 16-bit mono strings; fast_stream_f32.hip: f32 rows, what every other PCM format is unpacked to):
// as one more template parameter inside a shared translation unit an added instantiation has cost the existing one its register
// allocation before (fast_wave_dev.h).
#pragma once
#include <algorithm>
#include "fast_wave_dev.h"

namespace aukit {

// interp_qr of fast_wave_dev.h without Audio:resample's clamp (the stream's interpolated sample is not clamped, Q2)
template <int INTERP>
AUKIT_DEV float interp_qr_raw(const FastParams &F, const float *tab, unsigned q, unsigned rem) {
    const float fx = (float)rem * F.inv_b;
    const float p1 = tab[q];
    if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
        return fmaf(tab[q + 1] - p1, fx, p1);
    } else {
        const float p0 = tab[(int)q - 1], p2 = tab[q + 1], p3 = tab[q + 2];
        const float c3 = fmaf(1.5f, p1 - p2, 0.5f * (p3 - p0));
        const float c2 = fmaf(-0.5f, p3, fmaf(2.0f, p2, fmaf(-2.5f, p1, p0)));
        const float c1 = 0.5f * (p2 - p0);
        return fmaf(fmaf(fmaf(c3, fx, c2), fx, c1), fx, p1);
    }
}

// the previous lane's value (lane 0: `carry`) and lane 63's value without going through the LDS crossbar: __shfl_up / __shfl
// compile to ds_bpermute_b32; a wave-wide DPP shift (invalid source lane keeps `old`) and a v_readlane do the same in the VALU
AUKIT_DEV float prev_lane(float s, float carry) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry), __float_as_int(s), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
AUKIT_DEV float last_lane(float s) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63)); }

template <int SRC, int INTERP, int NV>
__global__ __launch_bounds__(256) void k_fast_wave_stream(const ResampleParams P, const FastParams F) {
    extern __shared__ float smf[];
    constexpr int HL = (INTERP == AUKIT_INTERP_CUBIC ? 1 : 0) + 1, HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *const sm = smf + wave * (unsigned)F.cap;
    const unsigned nwaves = gridDim.x * 4u;
    const float alpha = F.alpha;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    uint4 pre[NV];
    WaveTile cur = describe<SRC, HL, HR>(P, F, t);
    issue_loads<NV>(P, cur, lane, pre);
    for (;;) {
        write_lds<SRC, NV>(P, F, cur, lane, pre, sm);
        const bool first = (P.tiles_per_seg ? t % P.tiles_per_seg : t - as_const(P.seg_tile0)[as_const(P.tile_seg)[t]]) == 0;  // first tile of its iterator call
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe<SRC, HL, HR>(P, F, tn);
            issue_loads<NV>(P, nxt, lane, pre);  // in flight while this tile is interpolated
        }
        const float *tab = sm + cur.head + HL;  // tab[q] = d[1 + kb + q]
        float *orow = cur.orow;
        float carry = 0.f;  // the raw sample before the tile's first output: position n = r0 - a, one table step back when that is negative
        if (!first) carry = cur.r0 >= F.a ? interp_row<INTERP, false>(F, tab, cur.r0 - F.a) : interp_row<INTERP, false>(F, tab - 1, cur.r0 + F.b - F.a);
        if ((cur.cnt & 63u) == 0) {  // whole rows (every tile of an iterator call of 48000 outputs: 46 of 1024 and one of 896): (q, rem) advanced by additions, as in k_fast_wave
            const unsigned n0 = cur.r0 + (unsigned)lane * F.a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
            const int rows = (int)(cur.cnt >> 6);   // wave-uniform (the early exit keeps this a loop: hipcc does not unroll it, and it measured no slower)
            for (int r = 0; r < WT / 64; r++) {
                if (r >= rows) break;
                const float s = interp_qr_raw<INTERP>(F, tab, q, rem);
                const float prev = prev_lane(s, carry);
                carry = last_lane(s);
                const float ns = fmaf(alpha, s - prev, prev);                                                   // :2401
                orow[r * 64 + lane] = __builtin_amdgcn_fmed3f(ns * (ns < 0.f ? 128.f : 127.f), -128.f, 127.f);  // :2402
                rem += F.dr64;
                q += F.dq64;
                const bool wrap = rem >= F.b;
                rem -= wrap ? F.b : 0u;
                q += wrap ? 1u : 0u;
            }
        } else
        for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
            const unsigned j = rb + lane;
            const float s = interp_row<INTERP, false>(F, tab, cur.r0 + (j < cur.cnt ? j : cur.cnt - 1) * F.a);
            const float prev = prev_lane(s, carry);
            carry = last_lane(s);
            const float ns = fmaf(alpha, s - prev, prev);                                                // :2401
            if (j < cur.cnt) orow[j] = fminf(fmaxf(ns * (ns < 0.f ? 128.f : 127.f), -128.f), 127.f);  // :2402
        }
        if (!more) break;
        cur = nxt;
        t = tn;
    }
}


template <int SRC, int INTERP>
static int launch_nv_stream(aukit_ctx *ctx, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    switch (nv) {
    case 1: hipLaunchKernelGGL((k_fast_wave_stream<SRC, INTERP, 1>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 2: hipLaunchKernelGGL((k_fast_wave_stream<SRC, INTERP, 2>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 4: hipLaunchKernelGGL((k_fast_wave_stream<SRC, INTERP, 4>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 8:
        if constexpr (SRC == SRC_AUDIO_F32) { hipLaunchKernelGGL((k_fast_wave_stream<SRC, INTERP, 8>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break; }  // equal rates: 1027 samples per tile
        return fail(AUKIT_E_ARG, "bad NV");
    default: return fail(AUKIT_E_ARG, "bad NV");
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

}  // namespace aukit
