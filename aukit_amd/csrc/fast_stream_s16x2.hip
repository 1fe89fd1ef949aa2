// fast_stream_s16x2.hip — aukit.stream.pcm on interleaved 16-bit stereo PCM (what aukit.stream.wav hands over for nearly every WAV
// file, and what austream plays), AUKIT_F32 output: the stereo staging of fast_s16x2.hip with the stream.pcm epilogue of
// fast_stream.hip (aukit.lua:2397-2403), per channel; or, with `mono`, one table of (l + r) / 2 — the reference averages the
// channels as the samples are read (:2368) — and one output row.
//   s = raw interpolated sample (not clamped); ns = ls + alpha * (s - ls) with ls the RAW previous sample of the same iterator
//   call (0 for its first output, Q2); output clamp(ns * (ns < 0 and 128 or 127), -128, 127).
// f32 tolerance path (<= 1e-6 RMS against the fp64 oracle, tested); F64 storage keeps the reference-order kernel.
// Own translation unit: see fast_wave_dev.h.
#include <algorithm>
#include "fast_wave_dev.h"

namespace aukit {

template <> struct SrcTraits<SRC_PCM_S16LE_STEREO> { static constexpr int BYTES = 4, SPV = 4; };  // bytes per frame, frames per 16-byte vector

AUKIT_DEV float s16n(short s, const FastParams &F) { return (float)s * (s < 0 ? F.scale_neg : F.scale_pos); }  // aukit.lua:2345-2347

template <int INTERP>
AUKIT_DEV float interp_raw2(const FastParams &F, const float *tab, unsigned q, unsigned rem) {  // no clamp (Q2)
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

AUKIT_DEV float stream_out(float s, float prev, float alpha) {
    const float ns = fmaf(alpha, s - prev, prev);                                          // :2401
    return __builtin_amdgcn_fmed3f(ns * (ns < 0.f ? 128.f : 127.f), -128.f, 127.f);     // :2402
}

// the previous lane's value (lane 0: `carry`) and lane 63's value without going through the LDS crossbar: __shfl_up / __shfl
// compile to ds_bpermute_b32; a wave-wide DPP shift (invalid source lane keeps `old`) and a v_readlane do the same in the VALU
AUKIT_DEV float prev_lane(float s, float carry) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry), __float_as_int(s), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
AUKIT_DEV float last_lane(float s) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63)); }

template <int INTERP, int NV, bool MONO>
__global__ __launch_bounds__(256) void k_fast_wave_stream_s16x2(const ResampleParams P, const FastParams F) {
    extern __shared__ float smf[];
    constexpr int SRC = SRC_PCM_S16LE_STEREO;
    constexpr int NT = MONO ? 1 : 2;  // tables per wave
    constexpr int HL = (INTERP == AUKIT_INTERP_CUBIC ? 1 : 0) + 1, HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;  // one more tap to the left: the tile's carry
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *const smL = smf + wave * (unsigned)NT * (unsigned)F.cap;
    float *const smR = smL + (MONO ? 0 : F.cap);
    const unsigned nwaves = gridDim.x * 4u;
    const float alpha = F.alpha;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    uint4 pre[NV];
    WaveTile cur = describe<SRC, HL, HR>(P, F, t);
    issue_loads<NV>(P, cur, lane, pre);
    for (;;) {
        // ---- window → LDS
        auto put = [&](int idx, float l, float r) {
            if constexpr (MONO) smL[idx] = (l + r) * 0.5f;  // ((0 + l) + r) / channels  :2368
            else { smL[idx] = l; smR[idx] = r; }
        };
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = lane + 64 * i;
            if (v >= cur.nvec) continue;
            const unsigned ww[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
            float l[4], r[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { l[e] = s16n((short)(ww[e] & 0xFFFF), F); r[e] = s16n((short)(ww[e] >> 16), F); }
            if constexpr (MONO) {
                *reinterpret_cast<float4 *>(smL + 4 * v) = make_float4((l[0] + r[0]) * 0.5f, (l[1] + r[1]) * 0.5f, (l[2] + r[2]) * 0.5f, (l[3] + r[3]) * 0.5f);
            } else {
                *reinterpret_cast<float4 *>(smL + 4 * v) = make_float4(l[0], l[1], l[2], l[3]);
                *reinterpret_cast<float4 *>(smR + 4 * v) = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
        {
            auto frame = [&](const unsigned char *q, int c) { return s16n((short)(q[2 * c] | q[2 * c + 1] << 8), F); };
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare: vectors that straddle the allocation were zero-filled
                for (int idx = lane; idx < cur.nvec * 4; idx += 64) {
                    const unsigned char *q = cur.al + 4 * (size_t)idx;
                    const unsigned char *vb = cur.al + 16 * (size_t)(idx / 4);
                    if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) {
                        const bool in = q >= P.safe_lo && q + 4 <= P.safe_hi;
                        put(idx, in ? frame(q, 0) : 0.f, in ? frame(q, 1) : 0.f);
                    }
                }
            }
            // nil fall-backs of interpolate.{linear,cubic} (aukit.lua:259, :264) = replicated edge samples
            const int k_hi = cur.k_lo + cur.n_stage - 1;
            if (cur.k_lo < cur.w_lo) {
                const unsigned char *q = cur.base + 4 * (long long)cur.w_lo;
                const float el = frame(q, 0), er = frame(q, 1);
                for (int idx = lane; idx < cur.w_lo - cur.k_lo; idx += 64) put(cur.head + idx, el, er);
            }
            if (k_hi > cur.w_hi) {
                const unsigned char *q = cur.base + 4 * (long long)cur.w_hi;
                const float el = frame(q, 0), er = frame(q, 1);
                const int first = cur.w_hi + 1 - cur.k_lo;
                for (int idx = lane; idx < k_hi - cur.w_hi; idx += 64) put(cur.head + first + idx, el, er);
            }
        }
        unsigned sidx, tin;
        if (P.tiles_per_seg) { sidx = t / P.tiles_per_seg; tin = t - sidx * P.tiles_per_seg; }
        else { sidx = as_const(P.tile_seg)[t]; tin = t - as_const(P.seg_tile0)[sidx]; }
        const unsigned ostride = load_seg(P.segs, sidx).out_stride;   // (scalar loads: cf. floor_wave.hip)
        const bool first = tin == 0;  // first tile of its iterator call: ls = 0
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe<SRC, HL, HR>(P, F, tn);
            issue_loads<NV>(P, nxt, lane, pre);  // in flight while this tile is interpolated
        }
        const float *tabL = smL + cur.head + HL, *tabR = smR + cur.head + HL;  // tab[q] = d[1 + kb + q]
        float *orowL = cur.orow, *orowR = cur.orow + ostride;
        // the raw sample before the tile's first output: position n = r0 - a, one table step back when that is negative
        float carryL = 0.f, carryR = 0.f;
        if (!first) {
            const bool back = cur.r0 < F.a;
            const unsigned nc = back ? cur.r0 + F.b - F.a : cur.r0 - F.a;
            const unsigned qc = __umulhi(nc, F.magic), rc = nc - qc * F.b;
            carryL = interp_raw2<INTERP>(F, tabL - (back ? 1 : 0), qc, rc);
            if constexpr (!MONO) carryR = interp_raw2<INTERP>(F, tabR - (back ? 1 : 0), qc, rc);
        }
        auto row = [&](unsigned q, unsigned rem, unsigned j, bool active) {
            const float sl = interp_raw2<INTERP>(F, tabL, q, rem);
            const float pl = prev_lane(sl, carryL);
            carryL = last_lane(sl);
            const float ol = stream_out(sl, pl, alpha);
            if (active) orowL[j] = ol;
            if constexpr (!MONO) {
                const float sr = interp_raw2<INTERP>(F, tabR, q, rem);
                const float pr = prev_lane(sr, carryR);
                carryR = last_lane(sr);
                const float orr = stream_out(sr, pr, alpha);
                if (active) orowR[j] = orr;
            }
        };
        if ((cur.cnt & 63u) == 0) {  // whole rows (an iterator call of 48000 outputs ends in a tile of 896)
            const unsigned n0 = cur.r0 + (unsigned)lane * F.a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
            const int rows = (int)(cur.cnt >> 6);   // wave-uniform (the early exit keeps this a loop: hipcc does not unroll it, and it measured no slower)
            for (int r = 0; r < WT / 64; r++) {
                if (r >= rows) break;
                row(q, rem, (unsigned)(r * 64 + lane), true);
                rem += F.dr64;
                q += F.dq64;
                const bool wrap = rem >= F.b;
                rem -= wrap ? F.b : 0u;
                q += wrap ? 1u : 0u;
            }
        } else {
            for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
                const unsigned j = rb + lane;
                const unsigned n = cur.r0 + (j < cur.cnt ? j : cur.cnt - 1) * F.a;
                const unsigned q = __umulhi(n, F.magic);
                row(q, n - q * F.b, j, j < cur.cnt);
            }
        }
        if (!more) break;
        cur = nxt;
        t = tn;
    }
}

template <int INTERP, bool MONO>
static int launch_ss_nv(aukit_ctx *ctx, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    switch (nv) {
    case 1: hipLaunchKernelGGL((k_fast_wave_stream_s16x2<INTERP, 1, MONO>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 2: hipLaunchKernelGGL((k_fast_wave_stream_s16x2<INTERP, 2, MONO>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 4: hipLaunchKernelGGL((k_fast_wave_stream_s16x2<INTERP, 4, MONO>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 8: hipLaunchKernelGGL((k_fast_wave_stream_s16x2<INTERP, 8, MONO>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;  // equal rates: a tile's window is 1027 frames
    default: return fail(AUKIT_E_ARG, "bad NV");
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

// F.epi: 1 = both channels, 2 = mono (channels averaged as they are read)
int launch_fast_wave_stream_s16x2(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, unsigned grid) {
    const bool mono = F.epi == 2;
    const size_t lds = (size_t)F.cap * (mono ? 1 : 2) * 4 * 4;
    if (interp == AUKIT_INTERP_LINEAR) return mono ? launch_ss_nv<AUKIT_INTERP_LINEAR, true>(ctx, nv, P, F, lds, grid) : launch_ss_nv<AUKIT_INTERP_LINEAR, false>(ctx, nv, P, F, lds, grid);
    return mono ? launch_ss_nv<AUKIT_INTERP_CUBIC, true>(ctx, nv, P, F, lds, grid) : launch_ss_nv<AUKIT_INTERP_CUBIC, false>(ctx, nv, P, F, lds, grid);
}

}  // namespace aukit
