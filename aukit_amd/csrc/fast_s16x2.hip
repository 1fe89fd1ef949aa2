// fast_s16x2.hip — the wave-private f32 resampler (fast2.hip) for interleaved 16-bit stereo PCM, the layout of nearly every WAV
// file: aukit.pcm(data, 16, "signed", 2, rate):resample(new_rate) with AUKIT_F32 storage.
// A source element is a 4-byte frame (L, R); the window is de-interleaved into two LDS tables while it is staged, a lane computes
// the position once and evaluates both channels with it.  Same arithmetic per channel as k_fast_wave.  Own translation unit: see
// fast_wave_dev.h.
#include <algorithm>
#include "fast_wave_dev.h"

namespace aukit {

template <> struct SrcTraits<SRC_PCM_S16LE_STEREO> { static constexpr int BYTES = 4, SPV = 4; };  // bytes per frame, frames per 16-byte vector

AUKIT_DEV float s16f(short s, const FastParams &F) { return (float)s * (s < 0 ? F.scale_neg : F.scale_pos); }  // aukit.lua:1081

template <int INTERP, int NV>
__global__ __launch_bounds__(256) void k_fast_wave_s16x2(const ResampleParams P, const FastParams F) {
    extern __shared__ float smf[];
    constexpr int SRC = SRC_PCM_S16LE_STEREO;
    constexpr int HL = INTERP == AUKIT_INTERP_CUBIC ? 1 : 0, HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *const smL = smf + wave * 2u * (unsigned)F.cap;
    float *const smR = smL + F.cap;
    const unsigned nwaves = gridDim.x * 4u;
    const unsigned lane_a = (unsigned)lane * F.a;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    uint4 pre[NV];
    WaveTile cur = describe<SRC, HL, HR>(P, F, t);
    issue_loads<NV>(P, cur, lane, pre);
    pre_landed<NV>(pre);   // waited for on every edge into the loop (fast_wave_dev.h)
    for (;;) {
        // ---- window → two LDS tables
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = lane + 64 * i;
            if (v >= cur.nvec) continue;
            const unsigned ww[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
            float l[4], r[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { l[e] = s16f((short)(ww[e] & 0xFFFF), F); r[e] = s16f((short)(ww[e] >> 16), F); }
            *reinterpret_cast<float4 *>(smL + 4 * v) = make_float4(l[0], l[1], l[2], l[3]);
            *reinterpret_cast<float4 *>(smR + 4 * v) = make_float4(r[0], r[1], r[2], r[3]);
        }
        {
            auto frame = [&](const unsigned char *q, int c) { return s16f((short)(q[2 * c] | q[2 * c + 1] << 8), F); };
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare: vectors that straddle the allocation were zero-filled
                for (int idx = lane; idx < cur.nvec * 4; idx += 64) {
                    const unsigned char *q = cur.al + 4 * (size_t)idx;
                    const unsigned char *vb = cur.al + 16 * (size_t)(idx / 4);
                    if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) {
                        const bool in = q >= P.safe_lo && q + 4 <= P.safe_hi;
                        smL[idx] = in ? frame(q, 0) : 0.f;
                        smR[idx] = in ? frame(q, 1) : 0.f;
                    }
                }
            }
            // nil fall-backs of interpolate.{linear,cubic} (aukit.lua:259, :264) = replicated edge samples
            const int k_hi = cur.k_lo + cur.n_stage - 1;
            if (cur.k_lo < cur.w_lo) {
                const unsigned char *q = cur.base + 4 * (long long)cur.w_lo;
                const float el = frame(q, 0), er = frame(q, 1);
                for (int idx = lane; idx < cur.w_lo - cur.k_lo; idx += 64) { smL[cur.head + idx] = el; smR[cur.head + idx] = er; }
            }
            if (k_hi > cur.w_hi) {
                const unsigned char *q = cur.base + 4 * (long long)cur.w_hi;
                const float el = frame(q, 0), er = frame(q, 1);
                const int first = cur.w_hi + 1 - cur.k_lo;
                for (int idx = lane; idx < k_hi - cur.w_hi; idx += 64) { smL[cur.head + first + idx] = el; smR[cur.head + first + idx] = er; }
            }
        }
        // the tile's segment: the right channel's row is out_stride elements after the left one
        unsigned sidx;
        if (P.tiles_per_seg) sidx = t / P.tiles_per_seg; else sidx = as_const(P.tile_seg)[t];
        const unsigned ostride = load_seg(P.segs, sidx).out_stride;   // (scalar loads: cf. floor_wave.hip)
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe<SRC, HL, HR>(P, F, tn);
            issue_loads<NV>(P, nxt, lane, pre);  // in flight while this tile is interpolated
        }
        const float *tabL = smL + cur.head + HL, *tabR = smR + cur.head + HL;  // tab[q] = d[1 + kb + q]
        float *orowL = cur.orow, *orowR = cur.orow + ostride;
        // the rows wait in registers until the next tile's loads have been waited for, then they are stored (fast_wave_dev.h, "Where the wave waits")
        const bool full = cur.cnt == (unsigned)WT;  // wave-uniform
        float res[2 * (WT / 64)];
        if (full) {
            const unsigned n0 = cur.r0 + lane_a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
#pragma unroll
            for (int r = 0; r < WT / 64; r++) {
                res[2 * r] = interp_qr<SRC_PCM_S16LE_MONO, INTERP>(F, tabL, q, rem);
                res[2 * r + 1] = interp_qr<SRC_PCM_S16LE_MONO, INTERP>(F, tabR, q, rem);
                rem += F.dr64;
                q += F.dq64;
                const bool wrap = rem >= F.b;
                rem -= wrap ? F.b : 0u;
                q += wrap ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2 * (WT / 64); r++) res[r] = 0.f;
        }
        pre_landed<NV>(pre);
        hold_results(res);
        unsigned full2 = __builtin_amdgcn_readfirstlane((unsigned)full);
        asm volatile("" : "+s"(full2));  // opaque: or jump threading fuses the two `if (full)` and the wait is back inside the branches
        if (full2) {
#pragma unroll
            for (int r = 0; r < WT / 64; r++) { orowL[r * 64 + lane] = res[2 * r]; orowR[r * 64 + lane] = res[2 * r + 1]; }
        } else {
            for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
                const unsigned j = rb + lane;
                const unsigned n = cur.r0 + (j < cur.cnt ? j : cur.cnt - 1) * F.a;
                const unsigned q = __umulhi(n, F.magic);
                const unsigned rem = n - q * F.b;
                const float vl = interp_qr<SRC_PCM_S16LE_MONO, INTERP>(F, tabL, q, rem), vr = interp_qr<SRC_PCM_S16LE_MONO, INTERP>(F, tabR, q, rem);
                if (j < cur.cnt) { orowL[j] = vl; orowR[j] = vr; }
            }
        }
        if (!more) break;
        cur = nxt;
        t = tn;
    }
}

template <int INTERP>
static int launch_s16x2_nv(aukit_ctx *ctx, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    switch (nv) {
    case 1: hipLaunchKernelGGL((k_fast_wave_s16x2<INTERP, 1>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 2: hipLaunchKernelGGL((k_fast_wave_s16x2<INTERP, 2>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 4: hipLaunchKernelGGL((k_fast_wave_s16x2<INTERP, 4>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 8: hipLaunchKernelGGL((k_fast_wave_s16x2<INTERP, 8>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    default: return fail(AUKIT_E_ARG, "bad NV");
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

int launch_fast_wave_s16x2(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, unsigned grid) {
    const size_t lds = (size_t)F.cap * 2 * 4 * 4;  // two tables per wave, four waves
    if (interp == AUKIT_INTERP_LINEAR) return launch_s16x2_nv<AUKIT_INTERP_LINEAR>(ctx, nv, P, F, lds, grid);
    return launch_s16x2_nv<AUKIT_INTERP_CUBIC>(ctx, nv, P, F, lds, grid);
}

}  // namespace aukit
