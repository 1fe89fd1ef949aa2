// floor_wave.hip — aukit.stream.g711 (mono) on the wave-private tile engine of fast2.hip, bit-exact.
//
// The stream outputs are floor()ed and clamped (aukit.lua:2909-2910), so all that matters of the interpolated value is which
// integer interval it falls into.  The reference-order evaluation costs ≈41 fp64 instructions per output (exact division for the
// position, pow() emulation, the polynomial term by term).  Three tiers, each taken only when it is certain:
//   1. f32, straight-line for the whole wave.  G.711 samples are multiples of 1/64 up to 126 in magnitude (m / 0x40, :2891), so
//      they and the spline coefficients (multiples of 1/128 below 1400) are exact in f32; the three Horner FMAs round at
//      magnitudes below 2048 (half an ulp = 6.1e-5 each) and the position fraction is good to one ulp (4.5e-5 after the slope of
//      < 750): the f32 value is within 2.3e-4 of the exact one and is taken when it lies more than 5e-4 away from an integer
//      (TIER1_GUARD below; 1e-3 until late round 3).
//   2. (about one output in 500) the same polynomial in fp64 with exact rational positions and FMA Horner form, taken when more
//      than 1e-6 away from an integer.  Margin: the reference's x = (i-1)/ratio + 1 carries at most 48000 * 2^-53 = 5.3e-12 of
//      rounding error, the spline's slope is below 3 * 1004 (samples are < 512 in magnitude), so the two values differ by
//      < 2e-8; their own evaluation errors are ~1e-12.
//   3. otherwise — and at positions that are mathematically integers, where the reference's own `x % 1 == 0` test decides between
//      copy and interpolation (with an integer ratio that test is exact and the copy is taken in tier 1) — the reference-order
//      code (resample_dev.h) on the same window.
// Bit-exact whichever tier answers; the tests compare every output with the oracle and with the k_resample path.
//
// What the speed came from (same box, config 2b): fp64-only tier 2 with branches 488 G samples/s; + f32 tier, window staged one
// dword per lane into 256 slots (more resident waves), positions advanced by additions 519; tier 1 straight-line with ONE rare
// branch for tiers 2-3 734; + per-source-sample coefficient table for up-sampling by > 4.6 (six outputs share a polynomial) and
// no fp64 copy of the window (tiers 2-3 convert the exact f32 samples on read) 778.  Four outputs per lane with packed stores on
// top of that changed nothing (771): what is left is per-tile work (describe, staging, the coefficient pass, LDS round trips).
#include <algorithm>
#include <type_traits>
// Tiles of 960 outputs in this translation unit (15 rows): a multiple of lcm(64, b) for b = 2^i, 3 · 2^i and 5 · 2^i (i <= 6), so that a tile
// starts at phase 0 and a lane meets at most five phases (the PH > 0 instantiations below), and an iterator call's 48000 outputs are 50 whole tiles.
#define AUKIT_WT 960
#include "fast_wave_dev.h"
#include "resample_dev.h"

namespace aukit {

// How far from an integer tier 1's f32 value must lie to be taken.  Its error is below 2.3e-4 (header); 1e-3 was the guard of rounds 1-2 and early round 3.
// The turned-down outputs are what a quarter of this kernel's instructions were spent on after the phases went into registers (one output in
// 500 at 1e-3, and a tile of 960 outputs has one more often than not): 5e-4 halves them and keeps a factor of two over the bound.
#ifndef AUKIT_TIER1_GUARD
#define AUKIT_TIER1_GUARD 5e-4f
#endif
constexpr float TIER1_GUARD = AUKIT_TIER1_GUARD;

AUKIT_DEV void store_floor(signed char *p, float v) { *p = (signed char)(int)v; }
AUKIT_DEV void store_floor(double *p, float v) { *p = (double)v; }

// DW: the window is at most 16 vectors (up-sampling by > ~4.6, e.g. 8 kHz -> 48 kHz): every lane stages one dword of it instead
// of the first few lanes staging 16 bytes each, and the polynomial coefficients are tabulated per source sample (ccap entries).
// PH > 0 (round 3, with DW): the phases in registers.  Lane l of row r is output 64 r + l of its tile, at phase ((64 r + l) a) mod b; rows r and
// r + PH share it (PH = b / gcd(b, 64)) and every tile starts at phase 0, so fx, the "rem == 0" answer and the coefficient entry's offset
// are per-lane constants of the launch: a row is one table read, three FMAs and the guard — no position arithmetic, in a kernel bound by its VALU
// instructions.  Same f32 operations on the same values as the generic rows: the same tier decisions, bit for bit.
// AUDIT (round 4): the same kernel, and next to every tier-1 answer the fp64 value of tier 2 — the largest |tier 1 - tier 2| of
// the launch goes to audit[0] (bits of a non-negative float: they order like the value), the outputs compared to audit[1].  The error BOUND of the
// header is a derivation; this is what the device actually produced (AUKIT_OPT_COLLECT_STATS: aukit_ctx_get_counter, tests/test_gpu_guard_band.py).
template <int INTERP, bool DW, typename OUT_T, int PH, bool AUDIT = false>
__global__ __launch_bounds__(256) void k_floor_wave_g711(const ResampleParams P, const FastParams F, const unsigned ccap, const unsigned qstep, unsigned *audit = nullptr) {
    extern __shared__ float smf[];
    constexpr int SRC = SRC_G711_MONO;
    // one more tap to the left than the polynomial needs: at a mathematically integer position the reference's x may round to just
    // below the integer, and its floor(x) is then one table index lower
    constexpr int HL = (INTERP == AUKIT_INTERP_CUBIC ? 1 : 0) + 1, HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    constexpr int CW = INTERP == AUKIT_INTERP_CUBIC ? 4 : 2;  // floats per coefficient entry
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *const sm = smf + wave * ((unsigned)F.cap + ccap * CW);  // the window (exact in f32), then the coefficient table
    float *const cf = sm + F.cap;                                  // 16-byte aligned: F.cap is a multiple of 16
    const unsigned nwaves = gridDim.x * 4u;
    const double inv_b = 1.0 / (double)F.b;
    const float bf = (float)F.b, inv_bf = 1.0f / bf;
    const bool int_ratio = F.a == 1;  // ratio = b: (i-1)/ratio is an integer exactly when b divides i-1, in floating point too
    const float sc = (float)P.g711_scale;
    OUT_T *const out = reinterpret_cast<OUT_T *>(P.out);

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    [[maybe_unused]] float amax = 0.f;
    [[maybe_unused]] unsigned acnt = 0;
    constexpr int NPH = PH > 0 ? PH : 1;
    [[maybe_unused]] float fxp[NPH];
    [[maybe_unused]] unsigned qo[NPH];
    [[maybe_unused]] bool acc0[NPH];
    if constexpr (PH > 0) {
#pragma unroll
        for (int p = 0; p < PH; p++) {
            const unsigned n = (unsigned)(64 * p + lane) * F.a;
            const unsigned q = __umulhi(n, F.magic), rem = n - q * F.b;
            const float remf = (float)rem;
            float fx = remf * inv_bf;
            fx = __builtin_fmaf(__builtin_fmaf(-fx, bf, remf), inv_bf, fx);  // rem / b to one ulp, as in the generic rows
            fxp[p] = fx;
            qo[p] = q;
            acc0[p] = int_ratio && rem == 0;
        }
    }
    constexpr int NV = 1;
    uint4 pre[NV];
    unsigned pre1 = 0;
    auto load_window = [&](const WaveTile &w) {
        if constexpr (DW) {
            const unsigned char *vb = w.al + 16 * (size_t)(lane >> 2);  // same rule as issue_loads: a vector that straddles the allocation reads as zero
            pre1 = 0;
            if ((lane >> 2) < w.nvec && vb >= P.safe_lo && vb + 16 <= P.safe_hi) pre1 = *(const __attribute__((address_space(1))) unsigned *)(uintptr_t)(w.al + 4 * (size_t)lane);   // (global_load: through the generic pointer it was a FLAT load, counted on lgkmcnt too — every LDS wait of the tile waited for the prefetch)
        } else issue_loads<NV>(P, w, lane, pre);
    };
    WaveTile cur = describe<SRC, HL, HR>(P, F, t);
    load_window(cur);
    for (;;) {
        // ---- window → LDS (write_lds also patches vectors that straddle the allocation and replicates the table's ends into the
        // slots below / above it, which is what the nil fall-backs of interpolate.linear / cubic read (:259, :264))
        if constexpr (DW) {
            if ((lane >> 2) < cur.nvec)
                *reinterpret_cast<float4 *>(sm + 4 * lane) = make_float4(g711_f32b(pre1 & 0xFF, P.ulaw, sc), g711_f32b((pre1 >> 8) & 0xFF, P.ulaw, sc),
                                                                         g711_f32b((pre1 >> 16) & 0xFF, P.ulaw, sc), g711_f32b(pre1 >> 24, P.ulaw, sc));
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare
                for (int idx = lane; idx < cur.nvec * 16; idx += 64) {
                    const unsigned char *q = cur.al + idx;
                    const unsigned char *vb = cur.al + 16 * (size_t)(idx / 16);
                    if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) sm[idx] = (q >= P.safe_lo && q < P.safe_hi) ? sample_at<SRC>(P, F, q) : 0.f;
                }
            }
            WaveTile edges = cur;
            edges.nvec = 0;  // write_lds: nothing to convert, only the replicated edge samples
            write_lds<SRC, NV>(P, F, edges, lane, pre, sm);
        } else write_lds<SRC, NV>(P, F, cur, lane, pre, sm);
        // the tile's segment, for the reference-order fallback
        unsigned sidx, tin;
        if (P.tiles_per_seg) { sidx = t / P.tiles_per_seg; tin = t - sidx * P.tiles_per_seg; }
        else { sidx = as_const(P.tile_seg)[t]; tin = t - as_const(P.seg_tile0)[sidx]; }
        const Seg sg = load_seg(P.segs, sidx);   // (scalar loads: as vector loads they sat on vmcnt in front of the next tile's window request, and the wait for THEM — vmcnt(0), the request is in a fork — was the wait for the request: no prefetch)
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe<SRC, HL, HR>(P, F, tn);
            load_window(nxt);  // in flight while this tile is interpolated
        }
        const float *tab = sm + cur.head + HL;  // tab[q] = d[1 + kb + q]
        const float *tab_klo = sm + cur.head;   // slot of table index cur.k_lo
        if constexpr (DW) {  // the polynomial between source samples q and q + 1, once for the ~6 outputs that fall there
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int ncoef = cur.n_stage - HL - HR;
            for (int q = lane; q < ncoef; q += 64) {
                const float f1 = tab[q], f2 = tab[q + 1];
                if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                    *reinterpret_cast<float2 *>(cf + 2 * q) = make_float2(f2 - f1, f1);
                } else {
                    const float f0 = tab[q - 1], f3 = tab[q + 2];
                    const float c3 = __builtin_fmaf(1.5f, f1 - f2, 0.5f * (f3 - f0));
                    const float c2 = __builtin_fmaf(-2.5f, f1, f0) + __builtin_fmaf(2.0f, f2, -0.5f * f3);
                    const float c1 = 0.5f * (f2 - f0);
                    *reinterpret_cast<float4 *>(cf + 4 * q) = make_float4(c3, c2, c1, f1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        OUT_T *orow = out + sg.out_off + (size_t)tin * WT;
        // (q, rem) of the lane's first output by the verified reciprocal, then advanced by additions: 64 outputs further is
        // 64 a = dq64 b + dr64 input positions further
        unsigned q, rem;
        {
            const unsigned n0 = cur.r0 + (unsigned)lane * F.a;
            q = __umulhi(n0, F.magic);
            rem = n0 - q * F.b;
        }
        // tier 2's value: the polynomial in fp64 on the exact rational position
        auto exact_v = [&](unsigned q, unsigned rem) -> double {
            const double p1 = (double)tab[q];  // the same samples (exact in f32, converted on read)
            if (rem == 0) return p1;
            const double fxd = (double)rem * inv_b;
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                return __builtin_fma((double)tab[q + 1] - p1, fxd, p1);
            } else {
                const double p0 = (double)tab[(int)q - 1], p2 = (double)tab[q + 1], p3 = (double)tab[q + 2];
                const double c3 = __builtin_fma(1.5, p1 - p2, 0.5 * (p3 - p0));
                const double c2 = __builtin_fma(-2.5, p1, p0) + __builtin_fma(2.0, p2, -0.5 * p3);
                const double c1 = 0.5 * (p2 - p0);
                return __builtin_fma(__builtin_fma(__builtin_fma(c3, fxd, c2), fxd, c1), fxd, p1);
            }
        };
        // tiers 2 and 3 for one output (rare): returns the floored, clamped value
        auto slow = [&](unsigned q, unsigned rem, unsigned j) -> float {
            double v = exact_v(q, rem);
            bool ok = false;
            if (rem != 0) {
                const double frd = v - floor(v);
                ok = frd > 1e-6 && frd < 1 - 1e-6;
            }
            if (!ok) {  // tier 3
                bool isint;
                v = eval_at<INTERP>(P, sg, tab_klo, cur.k_lo, tin * (unsigned)WT + j, &isint);
            }
            return (float)lua_clamp(floor(v), -128, 127);
        };
        // (a full tile — all of a one-second chunk's but its last — runs a copy of the loop without the `active` predicate: three VALU instructions
        // per row in a kernel that is bound by them)
        auto rows = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;
        for (unsigned rb = 0; rb < (FULL ? (unsigned)WT : cur.cnt); rb += 64, q += F.dq64, rem += F.dr64) {
            const unsigned j = rb + lane;
            const bool active = FULL || j < cur.cnt;  // lanes past the end of a short tile compute on (their taps are still inside the tables)
            const bool wrap = rem >= F.b;
            rem -= wrap ? F.b : 0u;
            q += wrap ? 1u : 0u;
            // tier 1.  rem == 0 gives fx = 0 and w = p1 exactly: with an integer ratio that is the reference's copy branch (accepted
            // whatever the guard says); otherwise the reference's own `x % 1 == 0` decides and the lane goes to tier 3.
            const float remf = (float)rem;
            float fx = remf * inv_bf;
            fx = __builtin_fmaf(__builtin_fmaf(-fx, bf, remf), inv_bf, fx);  // rem / b to one ulp
            float w;
            if constexpr (DW) {
                if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                    const float2 c = *reinterpret_cast<const float2 *>(cf + 2 * q);
                    w = __builtin_fmaf(c.x, fx, c.y);
                } else {
                    const float4 c = *reinterpret_cast<const float4 *>(cf + 4 * q);
                    w = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c.x, fx, c.y), fx, c.z), fx, c.w);
                }
            } else {
                const float *tf = tab + q;
                const float f1 = tf[0];
                if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                    w = __builtin_fmaf(tf[1] - f1, fx, f1);
                } else {
                    const float f0 = tf[-1], f2 = tf[1], f3 = tf[2];
                    const float c3 = __builtin_fmaf(1.5f, f1 - f2, 0.5f * (f3 - f0));
                    const float c2 = __builtin_fmaf(-2.5f, f1, f0) + __builtin_fmaf(2.0f, f2, -0.5f * f3);
                    const float c1 = 0.5f * (f2 - f0);
                    w = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c3, fx, c2), fx, c1), fx, f1);
                }
            }
            if constexpr (AUDIT) { if (active) { amax = fmaxf(amax, (float)fabs((double)w - exact_v(q, rem))); acnt++; } }
            float fl = floorf(w);
            const float fr = w - fl;
            // (as plain boolean algebra — the masks stay in scalar registers; written `rem == 0 ? int_ratio : guard` hipcc built the choice out
            // of six VALU instructions per row, in a kernel that is bound by exactly those.  At rem == 0 without an integer ratio the guard
            // may answer as well: the reference returns p1 itself or, where its x rounds just below the integer, a value within 1e-9 of it.)
            const bool guard = fr > TIER1_GUARD && fr < 1 - TIER1_GUARD;
            const bool accept = guard || (int_ratio && rem == 0);
            if (active && !accept) fl = slow(q, rem, j);  // about one wave row in eight has such a lane
            if (active) store_floor(orow + j, __builtin_amdgcn_fmed3f(fl, -128.0f, 127.0f));  // :2909 (one v_med3; fl is never a nan: the window holds finite samples)
        }
        };
        [[maybe_unused]] auto rows_reg = [&]() {   // a full tile with the phases in registers
            // Every row stores its tier-1 answer and notes in a bit whether the guard turned it down; the (rare) turned-down outputs are redone
            // behind the rows by ONE copy of tiers 2-3 and stored again (same lane, same address: in order).  Inline, fifteen copies of the slow
            // path cost 137 VGPRs and measured slower than the generic rows.
            unsigned need = 0;
#pragma unroll
            for (int r = 0; r < WT / 64; r++) {
                const int p = r % NPH;
                const unsigned qq = qo[p] + (unsigned)(r / NPH) * qstep;
                const float fx = fxp[p];
                float w;
                if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                    const float2 c = *reinterpret_cast<const float2 *>(cf + 2 * qq);
                    w = __builtin_fmaf(c.x, fx, c.y);
                } else {
                    const float4 c = *reinterpret_cast<const float4 *>(cf + 4 * qq);
                    w = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c.x, fx, c.y), fx, c.z), fx, c.w);
                }
                const float fl = floorf(w);
                const float fr = w - fl;
                const bool guard = fr > TIER1_GUARD && fr < 1 - TIER1_GUARD;
                const bool accept = guard || acc0[p];
                need |= accept ? 0u : 1u << r;
                store_floor(orow + r * 64 + lane, __builtin_amdgcn_fmed3f(fl, -128.0f, 127.0f));
            }
            if (__any(need != 0)) {
                while (need) {
                    const int r = __builtin_ctz(need);
                    need &= need - 1;
                    const unsigned j = (unsigned)(r * 64 + lane), n = j * F.a;
                    const unsigned q2 = __umulhi(n, F.magic);
                    const float fl = slow(q2, n - q2 * F.b, j);
                    store_floor(orow + j, __builtin_amdgcn_fmed3f(fl, -128.0f, 127.0f));
                }
            }
        };
        if (cur.cnt == (unsigned)WT) {
            if constexpr (PH > 0 && DW) rows_reg(); else rows(std::true_type{});
        } else rows(std::false_type{});
        if (DW) __builtin_amdgcn_wave_barrier();  // the next tile's staging overwrites both tables
        if (!more) break;
        cur = nxt;
        t = tn;
    }
    if constexpr (AUDIT) {
        for (int o = 32; o; o >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, o)); acnt += __shfl_xor(acnt, o); }
        if (lane == 0 && audit) { atomicMax(&audit[0], __float_as_uint(amax)); atomicAdd(&audit[1], acnt); }
    }
}

bool fast_eligible(int src_kind, int interp, double old_rate, double new_rate, FastParams &F);

// returns true when this kernel took the launch (*rc = its status).  Needs: mono, linear / cubic, integer rates, a window of one
// vector per lane (up-sampling by >= ~1.06), and the reciprocal-based exact division verified for the fallback path.
bool floor_wave_g711_try(aukit_ctx *ctx, int interp, double old_rate, const std::vector<Seg> &segs, ResampleParams &P, int dtype, uint64_t algorithmic_bytes, int *rc) {
    if (ctx->exact_math) return false;
    FastParams F;
    if (!fast_eligible(SRC_G711_MONO, interp, old_rate, 48000, F)) return false;
    for (const Seg &g : segs)
        if (g.w_hi < g.w_lo && g.n_out) return false;
    const int hl = (interp == AUKIT_INTERP_CUBIC ? 1 : 0) + 1, hr = interp == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int win = (int)(((unsigned long long)(WT - 1) * F.a) / F.b) + 2 + hl + hr;
    if (win + 2 * 16 > 64 * 16) return false;  // one 16-byte vector per lane
    uint64_t max_tiles = 0, max_out = 0;
    for (const Seg &g : segs) { max_tiles = std::max<uint64_t>(max_tiles, (g.n_out + WT - 1) / WT); max_out = std::max<uint64_t>(max_out, g.n_out); }
    F.wc = (unsigned)(((unsigned long long)WT * F.a) / F.b);
    F.wd = (unsigned)(((unsigned long long)WT * F.a) % F.b);
    if ((double)max_tiles * (double)F.wd >= 4294967296.0 || ((double)max_tiles + 1) * (double)F.wc >= 2147483648.0) return false;
    if (((double)F.b + (double)WT * (double)F.a) * (double)F.b >= 4294967296.0) return false;
    const bool dw = win + 2 * 16 <= 16 * 16;
    F.cap = dw ? 16 * 16 : 64 * 16;  // window slots per wave
    const unsigned ccap = dw ? (unsigned)((win + 3) & ~3) : 0u;
    F.scale_pos = F.scale_neg = 0.f;
    F.dq64 = (unsigned)((64ull * F.a) / F.b);
    F.dr64 = (unsigned)((64ull * F.a) % F.b);
    P.ratio = 48000 / old_rate;
    P.rcp = 1.0 / P.ratio;
    P.exact_rcp = exact_div_verified(ctx, P.ratio, max_out + 2) ? 1 : 0;
    P.halo_l = hl; P.halo_r = hr; P.sinc_w = ctx->sinc_w;
    if ((*rc = plan_tiles_sized(ctx, segs, WT, P))) return true;
    if (P.n_tiles == 0) { *rc = AUKIT_OK; return true; }
    const size_t lds = ((size_t)F.cap + (size_t)ccap * (interp == AUKIT_INTERP_CUBIC ? 4 : 2)) * 4 * 4;  // per wave: window + coefficient table
    unsigned per_cu = 64;   // workgroups per CU in the grid (five are resident): 4 / 5 / 8 / 16 / 32 / 64 / 128 measured 1.65 / 1.55 / 1.52 / 1.43 / 1.38 / 1.35 / 1.36 ms on config 2b
    if (const char *e = getenv("AUKIT_FLOOR_PER_CU")) { const int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }   // tuning knob
    const unsigned grid = std::min<unsigned>((P.n_tiles + 3) / 4, (unsigned)ctx->num_cus * per_cu);
    // the phases in registers: tiles start at phase 0 and a lane meets PH = b / gcd(b, 64) <= 5 phases
    unsigned g64 = F.b, h64 = 64;
    while (h64) { const unsigned r = g64 % h64; g64 = h64; h64 = r; }
    unsigned ph = F.b / g64;
    const char *er = getenv("AUKIT_FLOOR_REGS");
    if (!dw || F.wd != 0 || (ph != 1 && ph != 3 && ph != 5) || (WT / 64) % (int)ph != 0 || (er && atoi(er) == 0)) ph = 0;
    const unsigned qstep = ph ? (unsigned)((64ull * ph * F.a) / F.b) : 0u;
    unsigned *audit = nullptr;
    if (ctx->collect_stats) {   // the audited instantiation (generic rows): every output's tier-1 value against tier 2's
        if ((*rc = ctx->fmt_flag.ensure(64))) return true;
        audit = reinterpret_cast<unsigned *>(ctx->fmt_flag.p) + 8;
        if (hipMemsetAsync(audit, 0, 8, ctx->stream) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "hipMemsetAsync failed"); return true; }
    }
    if ((*rc = ctx_begin_kernel(ctx))) return true;
#define AUKIT_FWA(I, T) do { if (dw) hipLaunchKernelGGL((k_floor_wave_g711<I, true, T, 0, true>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, qstep, audit); \
                             else hipLaunchKernelGGL((k_floor_wave_g711<I, false, T, 0, true>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, qstep, audit); } while (0)
#define AUKIT_FW(I, T) do { if (audit) AUKIT_FWA(I, T); else if (dw && ph == 1) hipLaunchKernelGGL((k_floor_wave_g711<I, true, T, 1>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, qstep); \
                            else if (dw && ph == 3) hipLaunchKernelGGL((k_floor_wave_g711<I, true, T, 3>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, qstep); \
                            else if (dw && ph == 5) hipLaunchKernelGGL((k_floor_wave_g711<I, true, T, 5>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, qstep); \
                            else if (dw) hipLaunchKernelGGL((k_floor_wave_g711<I, true, T, 0>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, qstep); \
                            else hipLaunchKernelGGL((k_floor_wave_g711<I, false, T, 0>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, qstep); } while (0)
    if (dtype == AUKIT_I8) { if (interp == AUKIT_INTERP_LINEAR) AUKIT_FW(AUKIT_INTERP_LINEAR, signed char); else AUKIT_FW(AUKIT_INTERP_CUBIC, signed char); }
    else { if (interp == AUKIT_INTERP_LINEAR) AUKIT_FW(AUKIT_INTERP_LINEAR, double); else AUKIT_FW(AUKIT_INTERP_CUBIC, double); }
#undef AUKIT_FW
#undef AUKIT_FWA
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_floor_wave_g711 launch failed"); return true; }
    if (audit) {
        unsigned h[2] = {0, 0};
        if (hipMemcpyAsync(h, audit, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "audit read-back failed"); return true; }
        float e; memcpy(&e, &h[0], 4);
        ctx->counters[AUKIT_COUNTER_TIER1_ERR_NANO] = (uint64_t)std::llround((double)e * 1e9);
        ctx->counters[AUKIT_COUNTER_TIER1_OUTPUTS] = h[1];
    }
    static thread_local char nm[96];
    snprintf(nm, sizeof nm, "k_floor_wave_g711<%s>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic");
    *rc = ctx_end_kernel(ctx, nm, algorithmic_bytes);
    return true;
}

}  // namespace aukit
