// fast_coef.hip — the wave-private f32 resampler for up-sampling by 2x and more (8 kHz G.711 → 48 kHz, 22.05 kHz → 48 kHz ...).
// Several outputs then fall between the same two source samples and share the polynomial: its coefficients (c3, c2, c1, p1 for
// cubic; slope, p1 for linear) are computed once per source sample of the wave's window into a second LDS table, and an output
// is one 16-byte (8-byte) LDS read plus the Horner form: 12 VALU instructions per 64 outputs instead of 21 (config 2a is
// write-dominated — 4.17 B per output — and was VALU-issue-bound).  Same arithmetic, same results as k_fast_wave.
// Its own translation unit: see fast_wave_dev.h.
#include <algorithm>
#include "fast_wave_dev.h"

namespace aukit {

template <int SRC, int INTERP, int NV>
__global__ __launch_bounds__(256) void k_fast_wave_coef(const ResampleParams P, const FastParams F, const unsigned ccap) {
    extern __shared__ float smf[];
    constexpr int HL = INTERP == AUKIT_INTERP_CUBIC ? 1 : 0, HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    constexpr int CW = INTERP == AUKIT_INTERP_CUBIC ? 4 : 2;  // floats per coefficient entry
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *const sm = smf + wave * ((unsigned)F.cap + ccap * CW);
    float *const cf = sm + F.cap;  // 16-byte aligned: F.cap is a multiple of 16
    const unsigned nwaves = gridDim.x * 4u;
    const unsigned lane_a = (unsigned)lane * F.a;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    uint4 pre[NV];
    WaveTile cur = describe<SRC, HL, HR>(P, F, t);
    issue_loads<NV>(P, cur, lane, pre);
    for (;;) {
        write_lds<SRC, NV>(P, F, cur, lane, pre, sm);
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe<SRC, HL, HR>(P, F, tn);
            issue_loads<NV>(P, nxt, lane, pre);  // in flight while this tile is interpolated
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float *tab = sm + cur.head + HL;  // tab[q] = d[1 + kb + q]
        const int ncoef = cur.n_stage - HL - HR;  // q = 0 .. klast
        for (int q = lane; q < ncoef; q += 64) {
            const float p1 = tab[q], p2 = tab[q + 1];
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                *reinterpret_cast<float2 *>(cf + 2 * q) = make_float2(p2 - p1, p1);
            } else {
                const float p0 = tab[q - 1], p3 = tab[q + 2];
                const float c3 = fmaf(1.5f, p1 - p2, 0.5f * (p3 - p0));
                const float c2 = fmaf(-0.5f, p3, fmaf(2.0f, p2, fmaf(-2.5f, p1, p0)));
                const float c1 = 0.5f * (p2 - p0);
                *reinterpret_cast<float4 *>(cf + 4 * q) = make_float4(c3, c2, c1, p1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float *orow = cur.orow;
        auto eval = [&](unsigned q, unsigned rem) -> float {
            const float fx = (float)rem * F.inv_b;
            float v, p1;
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                const float2 c = *reinterpret_cast<const float2 *>(cf + 2 * q);
                p1 = c.y;
                v = fmaf(c.x, fx, c.y);
            } else {
                const float4 c = *reinterpret_cast<const float4 *>(cf + 4 * q);
                p1 = c.w;
                v = fmaf(fmaf(fmaf(c.x, fx, c.y), fx, c.z), fx, c.w);
            }
            (void)p1;
            return __builtin_amdgcn_fmed3f(v, -1.0f, 1.0f);  // rem == 0 → fx == 0 → v == p1 exactly, and |p1| <= 1 for these sources (aukit.lua:666-668)
        };
        if (cur.cnt == (unsigned)WT) {
            const unsigned n0 = cur.r0 + lane_a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
#pragma unroll
            for (int r = 0; r < WT / 64; r++) {
                orow[r * 64 + lane] = eval(q, rem);
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
                const float v = eval(q, n - q * F.b);
                if (j < cur.cnt) orow[j] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next tile's staging overwrites both tables
        if (!more) break;
        cur = nxt;
        t = tn;
    }
}

template <int SRC, int INTERP>
static int launch_coef_nv(aukit_ctx *ctx, int nv, const ResampleParams &P, const FastParams &F, unsigned ccap, size_t lds, unsigned grid) {
    switch (nv) {
    case 1: hipLaunchKernelGGL((k_fast_wave_coef<SRC, INTERP, 1>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap); break;
    case 2: hipLaunchKernelGGL((k_fast_wave_coef<SRC, INTERP, 2>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap); break;
    default: return fail(AUKIT_E_ARG, "bad NV");
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

// s16le mono / G.711 mono, ratio >= 2.  `win` = staged samples per wave tile (upper bound); returns the LDS bytes per workgroup in *lds.
int launch_fast_wave_coef(aukit_ctx *ctx, int src_kind, int interp, int nv, int win, const ResampleParams &P, const FastParams &F, unsigned grid) {
    const unsigned ccap = (unsigned)((win + 3) & ~3);
    const size_t lds = ((size_t)F.cap + (size_t)ccap * (interp == AUKIT_INTERP_CUBIC ? 4 : 2)) * 4 * 4;
    if (src_kind == SRC_PCM_S16LE_MONO)
        return interp == AUKIT_INTERP_LINEAR ? launch_coef_nv<SRC_PCM_S16LE_MONO, AUKIT_INTERP_LINEAR>(ctx, nv, P, F, ccap, lds, grid)
                                             : launch_coef_nv<SRC_PCM_S16LE_MONO, AUKIT_INTERP_CUBIC>(ctx, nv, P, F, ccap, lds, grid);
    return interp == AUKIT_INTERP_LINEAR ? launch_coef_nv<SRC_G711_MONO, AUKIT_INTERP_LINEAR>(ctx, nv, P, F, ccap, lds, grid)
                                         : launch_coef_nv<SRC_G711_MONO, AUKIT_INTERP_CUBIC>(ctx, nv, P, F, ccap, lds, grid);
}

}  // namespace aukit
