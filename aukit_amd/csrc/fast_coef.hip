// fast_coef.hip — the wave-private f32 resampler for up-sampling by 2x and more (8 kHz G.711 → 48 kHz, 22.05 kHz → 48 kHz ...).
// Several outputs then fall between the same two source samples and share the polynomial: its coefficients (c3, c2, c1, p1 for
// cubic; slope, p1 for linear) are computed once per source sample of the wave's window into a second LDS table, and an output
// is one 16-byte (8-byte) LDS read plus the Horner form: 12 VALU instructions per 64 outputs instead of 21 (config 2a is
// write-dominated — 4.17 B per output — and was VALU-issue-bound).  Same arithmetic, same results as k_fast_wave.
// Its own translation unit: see fast_wave_dev.h.
#include <algorithm>
#include "fast_wave_dev.h"

namespace aukit {

// DW (G.711 with a window of at most 16 vectors, i.e. up-sampling by > ~4.6 like 8 kHz → 48 kHz): every lane stages one dword of
// the window instead of the first dozen lanes converting 16 bytes each, and the window table shrinks from 1024 to 256 floats per
// wave — staging was 40 % of the tile's instructions, and the smaller table lets 8 workgroups stay resident per CU.
template <int SRC, int INTERP, int NV, bool DW = false>
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
    unsigned pre1 = 0;
    auto load_window = [&](const WaveTile &w) {
        if constexpr (DW) {
            const unsigned char *vb = w.al + 16 * (size_t)(lane >> 2);  // same rule as issue_loads: a vector that straddles the allocation reads as zero
            pre1 = 0;
            if ((lane >> 2) < w.nvec && vb >= P.safe_lo && vb + 16 <= P.safe_hi) pre1 = *(const __attribute__((address_space(1))) unsigned *)(w.al + 4 * (size_t)lane);   // a global load (a generic dereference is a flat load)
        } else issue_loads<NV>(P, w, lane, pre);
    };
    WaveTile cur = describe<SRC, HL, HR>(P, F, t);
    load_window(cur);
    if constexpr (DW) asm volatile("" : "+v"(pre1)); else pre_landed<NV>(pre);   // waited for on every edge into the loop (fast_wave_dev.h)
    for (;;) {
        if constexpr (DW) {
            static_assert(!DW || SRC == SRC_G711_MONO, "dword staging is for one-byte samples");
            if ((lane >> 2) < cur.nvec) {
                const float sc = (float)P.g711_scale;
                *reinterpret_cast<float4 *>(sm + 4 * lane) = make_float4(g711_f32b(pre1 & 0xFF, P.ulaw, sc), g711_f32b((pre1 >> 8) & 0xFF, P.ulaw, sc),
                                                                         g711_f32b((pre1 >> 16) & 0xFF, P.ulaw, sc), g711_f32b(pre1 >> 24, P.ulaw, sc));
            }
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare: patch the zero-filled vectors byte by byte (as write_lds does)
                for (int idx = lane; idx < cur.nvec * 16; idx += 64) {
                    const unsigned char *q = cur.al + idx;
                    const unsigned char *vb = cur.al + 16 * (size_t)(idx / 16);
                    if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) sm[idx] = (q >= P.safe_lo && q < P.safe_hi) ? sample_at<SRC>(P, F, q) : 0.f;
                }
            }
            WaveTile edges = cur;
            edges.nvec = 0;  // write_lds: nothing to convert, only the replicated edge samples
            write_lds<SRC, NV>(P, F, edges, lane, pre, sm);
        } else
        write_lds<SRC, NV>(P, F, cur, lane, pre, sm);
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe<SRC, HL, HR>(P, F, tn);
            load_window(nxt);  // in flight while this tile is interpolated
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
        // the rows wait in registers until the next tile's loads have been waited for, then they are stored (fast_wave_dev.h, "Where the wave waits";
        // same-box A/B in DESIGN.md §3.7: k_wave_coef_f64, this kernel's fp64 sibling, gained 4 % from exactly this)
        const bool full = cur.cnt == (unsigned)WT;  // wave-uniform
        float res[WT / 64];
        if (full) {
            const unsigned n0 = cur.r0 + lane_a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
#pragma unroll
            for (int r = 0; r < WT / 64; r++) {
                res[r] = eval(q, rem);
                rem += F.dr64;
                q += F.dq64;
                const bool wrap = rem >= F.b;
                rem -= wrap ? F.b : 0u;
                q += wrap ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int r = 0; r < WT / 64; r++) res[r] = 0.f;
        }
        if constexpr (DW) asm volatile("" : "+v"(pre1)); else pre_landed<NV>(pre);
        hold_results(res);
        unsigned full2 = __builtin_amdgcn_readfirstlane((unsigned)full);
        asm volatile("" : "+s"(full2));  // opaque: or jump threading fuses the two `if (full)` and the wait is back inside the branches
        if (full2) {
#pragma unroll
            for (int r = 0; r < WT / 64; r++) orow[r * 64 + lane] = res[r];
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
    if (src_kind == SRC_G711_MONO && nv == 1 && win + 2 * 16 <= 16 * 16) {  // dword staging, 256-float window
        FastParams Fd = F;
        Fd.cap = 16 * 16;
        const size_t ldsd = ((size_t)Fd.cap + (size_t)ccap * (interp == AUKIT_INTERP_CUBIC ? 4 : 2)) * 4 * 4;
        if (interp == AUKIT_INTERP_LINEAR) hipLaunchKernelGGL((k_fast_wave_coef<SRC_G711_MONO, AUKIT_INTERP_LINEAR, 1, true>), dim3(grid), dim3(256), ldsd, ctx->stream, P, Fd, ccap);
        else hipLaunchKernelGGL((k_fast_wave_coef<SRC_G711_MONO, AUKIT_INTERP_CUBIC, 1, true>), dim3(grid), dim3(256), ldsd, ctx->stream, P, Fd, ccap);
        AUKIT_HIP_CHECK(hipGetLastError());
        return AUKIT_OK;
    }
    const size_t lds = ((size_t)F.cap + (size_t)ccap * (interp == AUKIT_INTERP_CUBIC ? 4 : 2)) * 4 * 4;
    if (src_kind == SRC_PCM_S16LE_MONO)
        return interp == AUKIT_INTERP_LINEAR ? launch_coef_nv<SRC_PCM_S16LE_MONO, AUKIT_INTERP_LINEAR>(ctx, nv, P, F, ccap, lds, grid)
                                             : launch_coef_nv<SRC_PCM_S16LE_MONO, AUKIT_INTERP_CUBIC>(ctx, nv, P, F, ccap, lds, grid);
    return interp == AUKIT_INTERP_LINEAR ? launch_coef_nv<SRC_G711_MONO, AUKIT_INTERP_LINEAR>(ctx, nv, P, F, ccap, lds, grid)
                                         : launch_coef_nv<SRC_G711_MONO, AUKIT_INTERP_CUBIC>(ctx, nv, P, F, ccap, lds, grid);
}

}  // namespace aukit
