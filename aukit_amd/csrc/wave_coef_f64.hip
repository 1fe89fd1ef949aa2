// wave_coef_f64.hip — `aukit.g711(d, ulaw, 1, rate):resample(new_rate, interp)` computed in fp64 and stored as f32 (AUKIT_OPT_EXACT_MATH = 1) for
// up-sampling by more than ≈ 4.6 (8 kHz → 48 kHz: BASELINE config 2a): the fp64 sibling of k_fast_wave_coef (fast_coef.hip), in the same
// sense in which k_wave_f64 is the fp64 sibling of k_fast_wave — the reference's arithmetic TYPE (every sample, coefficient, product and sum
// is a double; aukit.lua:261-266, :1374-1379 compute in Lua doubles), not its operation order:
//   * a G.711 sample is an integer below 2^14 times 2^-13: exact in f32 and in f64, staged once per tile as in the f32 kernel (every lane one
//     dword of the window);
//   * six outputs fall between the same two source samples at 8 → 48 kHz and share the polynomial, so its coefficients
//     (c3, c2, c1, p1 — the regrouping of k_wave_f64's Horner variant) are computed once per SOURCE sample in fp64 into an LDS table, and an
//     output is two ds_read_b128 + three fp64 FMAs on fx = rem · RN(1 / b) — 32 bytes of LDS per output where the phase-table kernel reads 64
//     (its LDS reads are what binds that one);
//   * positions are the exact rationals (q, rem) advanced by additions; rem == 0 gives fx = 0 and the sample itself (:666); the clamp runs
//     after the rounding to f32 (monotone, representable bounds).
// Against the oracle: ≤ one f32 ulp (tests/test_gpu_wave_f64.py).  Write-dominated: 4.17 B per output.
#include <algorithm>
// tiles of 960 outputs in this translation unit (15 rows): a multiple of lcm(64, b) for b = 2^i, 3 · 2^i, 5 · 2^i (i <= 6), so that a tile starts at phase 0
// and a lane meets at most five phases (the PH > 0 instantiations: 8 → 48 kHz has b = 6, PH = 3)
#define AUKIT_WT 960
#include "fast_wave_dev.h"

namespace aukit {

bool fast_eligible(int src_kind, int interp, double old_rate, double new_rate, FastParams &F);

typedef double cdbl2 __attribute__((ext_vector_type(2)));
struct CoefRow {   // (c3, c2), (c1, p1) of one source sample, read from LDS by hand
    cdbl2 a, b;
    AUKIT_DEV void issue(unsigned addr) { asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(a), "=&v"(b) : "v"(addr)); }
    template <int K> AUKIT_DEV void wait() { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(K)); }
};

// PH > 0: the phases in registers (as k_wave_f64_reg, wave_f64.hip) — fx and the coefficient entry's offset of rows r, r + PH, … are per-lane constants
// of the launch; a row is its two table reads, three FMAs, the rounding and the clamp.  Same operations on the same values as the generic rows.
template <int INTERP, int PH>
__global__ __launch_bounds__(256) void k_wave_coef_f64(const ResampleParams P, const FastParams F, const unsigned ccap, const double inv_b, const unsigned qstep) {
    extern __shared__ double smd[];
    constexpr int SRC = SRC_G711_MONO;
    constexpr int HL = INTERP == AUKIT_INTERP_CUBIC ? 1 : 0, HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    constexpr int CW = INTERP == AUKIT_INTERP_CUBIC ? 4 : 2;  // doubles per coefficient entry
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *const base = smd + wave * ((unsigned)F.cap / 2 + ccap * CW);   // per wave: F.cap floats of window, then the coefficient table
    float *const sm = reinterpret_cast<float *>(base);
    double *const cf = base + F.cap / 2;
    const unsigned nwaves = gridDim.x * 4u;
    const unsigned lane_a = (unsigned)lane * F.a;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    constexpr int NPH = PH > 0 ? PH : 1;
    [[maybe_unused]] double fxp[NPH];
    [[maybe_unused]] unsigned qo[NPH];
    if constexpr (PH > 0) {
#pragma unroll
        for (int p = 0; p < PH; p++) {
            const unsigned n = (unsigned)(64 * p + lane) * F.a;
            const unsigned q = __umulhi(n, F.magic), rem = n - q * F.b;
            fxp[p] = (double)rem * inv_b;
            qo[p] = q;
        }
    }
    uint4 pre[1];
    unsigned pre1 = 0;
    auto load_window = [&](const WaveTile &w) {
        const unsigned char *vb = w.al + 16 * (size_t)(lane >> 2);  // a vector that straddles the allocation reads as zero (patched below)
        pre1 = 0;
        if ((lane >> 2) < w.nvec && vb >= P.safe_lo && vb + 16 <= P.safe_hi) pre1 = *(const __attribute__((address_space(1))) unsigned *)(w.al + 4 * (size_t)lane);   // global_load (a generic dereference is a flat_load, which also counts in lgkmcnt: the rows' counted waits would sit it out)
    };
    WaveTile cur = describe<SRC, HL, HR>(P, F, t);
    load_window(cur);
    asm volatile("" : "+v"(pre1));   // waited for on every edge into the loop (fast_wave_dev.h: or hipcc's vmcnt(0) lands at the top of the loop, behind the stores)
    for (;;) {
        if ((lane >> 2) < cur.nvec) {
            const float sc = (float)P.g711_scale;
            *reinterpret_cast<float4 *>(sm + 4 * lane) = make_float4(g711_f32b(pre1 & 0xFF, P.ulaw, sc), g711_f32b((pre1 >> 8) & 0xFF, P.ulaw, sc),
                                                                     g711_f32b((pre1 >> 16) & 0xFF, P.ulaw, sc), g711_f32b(pre1 >> 24, P.ulaw, sc));
        }
        {
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare
                for (int idx = lane; idx < cur.nvec * 16; idx += 64) {
                    const unsigned char *q = cur.al + idx;
                    const unsigned char *vb = cur.al + 16 * (size_t)(idx / 16);
                    if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) sm[idx] = (q >= P.safe_lo && q < P.safe_hi) ? sample_at<SRC>(P, F, q) : 0.f;
                }
            }
            WaveTile edges = cur;
            edges.nvec = 0;  // write_lds: nothing to convert, only the replicated edge samples (the nil fall-backs of :259, :264)
            write_lds<SRC, 1>(P, F, edges, lane, pre, sm);
        }
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
            const double p1 = (double)tab[q], p2 = (double)tab[q + 1];
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                *reinterpret_cast<double2 *>(cf + 2 * q) = make_double2(p2 - p1, p1);
            } else {
                const double p0 = (double)tab[q - 1], p3 = (double)tab[q + 2];
                const double c3 = __builtin_fma(1.5, p1 - p2, 0.5 * (p3 - p0));
                const double c2 = __builtin_fma(-0.5, p3, __builtin_fma(2.0, p2, __builtin_fma(-2.5, p1, p0)));
                const double c1 = 0.5 * (p2 - p0);
                double2 *e = reinterpret_cast<double2 *>(cf + 4 * q);
                e[0] = make_double2(c3, c2);
                e[1] = make_double2(c1, p1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float *orow = cur.orow;
        auto eval = [&](unsigned q, unsigned rem) -> float {
            const double fx = (double)rem * inv_b;
            double v;
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                const double2 c = *reinterpret_cast<const double2 *>(cf + 2 * q);
                v = __builtin_fma(c.x, fx, c.y);
            } else {
                const double2 a = reinterpret_cast<const double2 *>(cf + 4 * q)[0], b = reinterpret_cast<const double2 *>(cf + 4 * q)[1];
                v = __builtin_fma(__builtin_fma(__builtin_fma(a.x, fx, a.y), fx, b.x), fx, b.y);
            }
            return __builtin_amdgcn_fmed3f((float)v, -1.0f, 1.0f);  // :667-668
        };
        // A tile's rows wait in registers until the NEXT tile's window dword has been waited for, and only then are stored (the order of
        // k_fast_wave / k_wave_f64: loads and stores share vmcnt, so a wait that stands right behind sixteen fresh stores — where hipcc puts it
        // when the rows are stored as they are evaluated — sits out their write latency once per tile; here the only stores still counted
        // are the tile-before's).
        const bool full = cur.cnt == (unsigned)WT;  // wave-uniform
        float res[WT / 64];
        if (full && PH > 0) {
            if constexpr (INTERP == AUKIT_INTERP_CUBIC) {
                const unsigned cf0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)cf;
                CoefRow nx;
                nx.issue(cf0 + 32u * qo[0]);
#pragma unroll
                for (int r = 0; r < WT / 64; r++) {
                    CoefRow c = nx;
                    if (r + 1 < WT / 64) {
                        nx.issue(cf0 + 32u * (qo[(r + 1) % NPH] + (unsigned)((r + 1) / NPH) * qstep));
                        c.template wait<2>();
                    } else {
                        c.template wait<0>();
                    }
                    const double fx = fxp[r % NPH];
                    const double v = __builtin_fma(__builtin_fma(__builtin_fma(c.a.x, fx, c.a.y), fx, c.b.x), fx, c.b.y);
                    res[r] = __builtin_amdgcn_fmed3f((float)v, -1.0f, 1.0f);
                }
            } else {
#pragma unroll
                for (int r = 0; r < WT / 64; r++) {
                    const double2 c = *reinterpret_cast<const double2 *>(cf + 2 * (qo[r % NPH] + (unsigned)(r / NPH) * qstep));
                    res[r] = __builtin_amdgcn_fmed3f((float)__builtin_fma(c.x, fxp[r % NPH], c.y), -1.0f, 1.0f);
                }
            }
        } else if (full) {
            const unsigned n0 = cur.r0 + lane_a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
            if constexpr (INTERP == AUKIT_INTERP_CUBIC) {
                // a row's two table reads are issued one row ahead and waited for with a counted lgkmcnt (LDS returns in order), as in
                // wave_f64.hip's Row: hipcc on its own waits lgkmcnt(0) right behind every row's reads (tools/isa_check.py replays the counters)
                const unsigned cf0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)cf;
                CoefRow nx;
                nx.issue(cf0 + 32u * q);
#pragma unroll
                for (int r = 0; r < WT / 64; r++) {
                    const double fx = (double)rem * inv_b;
                    CoefRow c = nx;
                    if (r + 1 < WT / 64) {
                        rem += F.dr64;
                        q += F.dq64;
                        const bool wrap = rem >= F.b;
                        rem -= wrap ? F.b : 0u;
                        q += wrap ? 1u : 0u;
                        nx.issue(cf0 + 32u * q);
                        c.template wait<2>();
                    } else {
                        c.template wait<0>();
                    }
                    const double v = __builtin_fma(__builtin_fma(__builtin_fma(c.a.x, fx, c.a.y), fx, c.b.x), fx, c.b.y);
                    res[r] = __builtin_amdgcn_fmed3f((float)v, -1.0f, 1.0f);
                }
            } else {
#pragma unroll
                for (int r = 0; r < WT / 64; r++) {
                    res[r] = eval(q, rem);
                    rem += F.dr64;
                    q += F.dq64;
                    const bool wrap = rem >= F.b;
                    rem -= wrap ? F.b : 0u;
                    q += wrap ? 1u : 0u;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < WT / 64; r++) res[r] = 0.f;
        }
        // ---- the next tile's dword has landed (on the straight path, as in k_wave_f64: inside the branches hipcc finds a way around it)
        asm volatile("" : "+v"(pre1));
#pragma unroll
        for (int r = 0; r < WT / 64; r++) asm volatile("" : "+v"(res[r]));
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

// returns true when this kernel took the launch (*rc = its status): G.711 mono → linear / cubic → f32, integer rates, up-sampling by > ≈ 4.6
bool wave_coef_f64_try(aukit_ctx *ctx, int src_kind, int interp, double old_rate, double new_rate, const std::vector<Seg> &segs, ResampleParams &P,
                       uint64_t algorithmic_bytes, int *rc) {
    if (src_kind != SRC_G711_MONO || (interp != AUKIT_INTERP_LINEAR && interp != AUKIT_INTERP_CUBIC)) return false;
    FastParams F;
    if (!fast_eligible(SRC_G711_MONO, interp, old_rate, new_rate, F)) return false;
    for (const Seg &g : segs)
        if (g.w_hi < g.w_lo && g.n_out) return false;
    const int hl = interp == AUKIT_INTERP_CUBIC ? 1 : 0, hr = interp == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int win = (int)(((unsigned long long)(WT - 1) * F.a) / F.b) + 2 + hl + hr;  // staged samples per wave tile (upper bound)
    if (win + 2 * 16 > 16 * 16) return false;   // one dword per lane must cover the window
    uint64_t max_tiles = 0;
    for (const Seg &g : segs) max_tiles = std::max<uint64_t>(max_tiles, (g.n_out + WT - 1) / WT);
    F.wc = (unsigned)(((unsigned long long)WT * F.a) / F.b);
    F.wd = (unsigned)(((unsigned long long)WT * F.a) % F.b);
    if ((double)max_tiles * (double)F.wd >= 4294967296.0 || ((double)max_tiles + 1) * (double)F.wc >= 2147483648.0) return false;
    if (((double)F.b + (double)WT * (double)F.a) * (double)F.b >= 4294967296.0) return false;
    F.cap = 16 * 16;   // floats per wave window
    F.dq64 = (unsigned)((64ull * F.a) / F.b);
    F.dr64 = (unsigned)((64ull * F.a) % F.b);
    if ((*rc = plan_tiles_sized(ctx, segs, WT, P))) return true;
    if (P.n_tiles == 0) { *rc = AUKIT_OK; return true; }
    const unsigned ccap = (unsigned)((win + 3) & ~3);
    const int cw = interp == AUKIT_INTERP_CUBIC ? 4 : 2;
    const size_t lds = ((size_t)F.cap / 2 + (size_t)ccap * cw) * 8 * 4;
    // the phases in registers: a lane meets PH = b / gcd(b, 64) <= 5 phases and every tile starts at phase 0
    unsigned g64 = F.b, h64 = 64;
    while (h64) { const unsigned r = g64 % h64; g64 = h64; h64 = r; }
    unsigned ph = F.b / g64;
    const char *er = getenv("AUKIT_COEF_REGS");
    if (F.wd != 0 || (ph != 1 && ph != 3 && ph != 5) || (WT / 64) % (int)ph != 0 || (er && atoi(er) == 0)) ph = 0;
    const unsigned qstep = ph ? (unsigned)((64ull * ph * F.a) / F.b) : 0u;
    unsigned per_cu = 512;   // workgroups per CU in the grid (six are resident): 64 / 128 / 256 / 512 / 1024 / 2048 measured 1.573 / 1.550 / 1.541 / 1.473 / 1.581 / 1.771 ms on config 2a (a wave then takes 3-4 tiles: the hardware's dispatcher balances better than the static tile stride)
    if (const char *e = getenv("AUKIT_FAST_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }
    const unsigned grid = std::min<unsigned>((P.n_tiles + 3) / 4, (unsigned)ctx->num_cus * per_cu);
    if ((*rc = ctx_begin_kernel(ctx))) return true;
    const double inv_b = 1.0 / (double)F.b;
#define AUKIT_WC(I) do { if (ph == 1) hipLaunchKernelGGL((k_wave_coef_f64<I, 1>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, inv_b, qstep); \
                         else if (ph == 3) hipLaunchKernelGGL((k_wave_coef_f64<I, 3>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, inv_b, qstep); \
                         else if (ph == 5) hipLaunchKernelGGL((k_wave_coef_f64<I, 5>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, inv_b, qstep); \
                         else hipLaunchKernelGGL((k_wave_coef_f64<I, 0>), dim3(grid), dim3(256), lds, ctx->stream, P, F, ccap, inv_b, qstep); } while (0)
    if (interp == AUKIT_INTERP_LINEAR) AUKIT_WC(AUKIT_INTERP_LINEAR); else AUKIT_WC(AUKIT_INTERP_CUBIC);
#undef AUKIT_WC
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_wave_coef_f64 launch failed"); return true; }
    static thread_local char nm[96];
    snprintf(nm, sizeof nm, "k_wave_coef_f64<g711_mono,%s>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic");
    *rc = ctx_end_kernel(ctx, nm, algorithmic_bytes);
    return true;
}

}  // namespace aukit
