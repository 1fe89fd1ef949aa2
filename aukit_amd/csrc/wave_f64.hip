// wave_f64.hip — `aukit.pcm(d, 16, "signed", 1, rate):resample(new_rate, interp)` computed in fp64 and stored as f32: the graded
// configuration of SURVEY §8d ("config T: f32 store of fp64 math"), AUKIT_OPT_EXACT_MATH = 1 with AUKIT_F32 storage.
//
// Arithmetic.  The reference evaluates, per output i (aukit.lua:662-669, :261-266), x = (i-1)/ratio + 1 and the Catmull-Rom cubic
//   (-0.5 p0 + 1.5 p1 - 1.5 p2 + 0.5 p3) fx^3 + (p0 - 2.5 p1 + 2 p2 - 0.5 p3) fx^2 + (-0.5 p0 + 0.5 p2) fx + p1
// in doubles.  k_exact_wave (exact_wave.hip) reproduces that expression operation by operation (≈41 fp64 instructions per output:
// verified reciprocal division, a correctly rounded pow(fx, 3), every product and sum in the reference's order) and is what AUKIT_F64
// storage uses.  A result that is rounded to f32 on its way out cannot show the last bits of those doubles, so this kernel keeps the
// *arithmetic type* (every tap, weight, product and sum is an fp64 value) and drops the operation order:
//   * the position is the exact rational (i-1)·a/b (a/b = old_rate/new_rate in lowest terms) carried as (q, rem) by integer
//     additions, so fx = rem/b takes one of b values;
//   * the polynomial is regrouped by tap, out = w0(fx) p0 + w1(fx) p1 + w2(fx) p2 + w3(fx) p3, with the four weights of every one of
//     the b phases computed on the host in extended precision, rounded to fp64 once and kept in LDS (b ≤ 512: 44.1 → 48 kHz has
//     b = 160, 5 KiB) — one multiply and three FMAs per output instead of fifteen fp64 operations; for larger b the kernel
//     evaluates the same polynomial as an fp64 Horner form on fx = rem · RN(1/b);
//   * samples are staged once per tile as doubles, s · 2^-15 (s < 0) or s · RN(1/32767) (≤ 1 ulp from the reference's s / 32767);
//   * `x % 1 == 0` (:666) is rem == 0, whose weights are (0, 1, 0, 0): the copy falls out of the same expression, and samples
//     of a 16-bit source lie in [-1, 1], so clamping them changes nothing; the clamp runs after the rounding to f32 (monotone
//     rounding and representable bounds: the same value as clamping before it).
// The polynomial regrouping moves a result by ulps of fp64 (1e-16).  The exact position moves it more: the reference's x is a
// rounded double (relative error 1e-16 of x, up to 5e-11 in fx ten seconds into a 44.1 kHz stream), so its interpolated double
// and the one computed here differ by up to ~1e-11 — the kernel is the closer of the two to the real-number value.  In the f32
// store that shows as neighbouring floats in 1e-4 … 1e-3 of the outputs, never more than one f32 ulp (tests/test_gpu_wave_f64.py
// asserts both); the bar of SURVEY §8d is 1e-6 RMS.
//
// Memory pipeline.  A wave owns tiles of TILE outputs and a private LDS window; per tile it
//   1. converts the tile's raw 16-bit samples (already in LDS) to doubles in its window;
//   2. starts the NEXT tile's raw samples on their way, global → LDS directly (`global_load_lds_dwordx4`: no VGPR holds them, so
//      hipcc has no load result to wait for anywhere);
//   3. evaluates the tile's TILE/64 rows into registers — LDS reads hand-issued one row ahead with counted lgkmcnt waits;
//   4. waits vmcnt(0) — the DMA of step 2 has had the whole of step 3 to land, and the only stores still counted are the tile
//      before's, a tile older still;
//   5. issues the tile's row stores, which drain while steps 1-3 of the next tile run.
// The order 3-4-5 is the point.  On gfx9 loads and stores share vmcnt and return out of order with respect to each other, so a wave
// that must see its next tile's samples can only wait for vmcnt(0).  With the stores issued row by row (the first version of this
// kernel, and k_fast_wave in round 1) that wait sits right behind sixteen fresh stores and every wave sits out their write latency
// once per tile; a hand-counted vmcnt(16) there returns garbage (tried: the loads are overtaken).  Same-box A/B in DESIGN.md §3.
#include <algorithm>
#include "fast_wave_dev.h"

namespace aukit {

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define AUKIT_GLOBAL_AS __attribute__((address_space(1)))
#define AUKIT_LDS_AS __attribute__((address_space(3)))
AUKIT_DEV unsigned lds_addr(const void *p) { return (unsigned)(size_t)(const AUKIT_LDS_AS char *)p; }

// ---- one row's operands, read from LDS by hand.  hipcc merges adjacent 8-byte reads into ds_read2_b64 (8 LDS cycles where two
// ds_read_b64 cost 2 each, MI355X_MICROARCH.md §LDS), waits lgkmcnt(0) for every row before it issues the next row's reads, and —
// knowing that an LDS-DMA is in flight — would put a vmcnt(0) in front of its own ds_reads.  Here a row's reads are issued as
// written and waited for only after the NEXT row's are in the queue (LDS returns in order: lgkmcnt(N) = this row has landed, N = reads
// per row).  The values pass through the s_waitcnt statement as in/out operands so that no use can be scheduled above it.
template <int INTERP, bool TAB> struct Row;
template <> struct Row<AUKIT_INTERP_CUBIC, true> {  // 4 taps, 4 weights of the phase
    static constexpr int N = 6;
    double p0, p1, p2, p3; dbl2 w01, w23;
    AUKIT_DEV void issue(unsigned tap, unsigned wa, unsigned wb) {
        asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:8\n\tds_read_b64 %2, %6 offset:16\n\tds_read_b64 %3, %6 offset:24\n\t"
                     "ds_read_b128 %4, %7\n\tds_read_b128 %5, %8"
                     : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(w01), "=&v"(w23) : "v"(tap), "v"(wa), "v"(wb));
    }
    template <int K> AUKIT_DEV void wait() { asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(w01), "+v"(w23) : "n"(K)); }
    AUKIT_DEV double eval(unsigned, double) const { return __builtin_fma(w23.y, p3, __builtin_fma(w23.x, p2, __builtin_fma(w01.y, p1, w01.x * p0))); }
};
template <> struct Row<AUKIT_INTERP_LINEAR, true> {  // 2 taps, fx of the phase
    static constexpr int N = 3;
    double p1, p2, fx;
    AUKIT_DEV void issue(unsigned tap, unsigned wa, unsigned) {
        asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %4" : "=&v"(p1), "=&v"(p2), "=&v"(fx) : "v"(tap), "v"(wa));
    }
    template <int K> AUKIT_DEV void wait() { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(p1), "+v"(p2), "+v"(fx) : "n"(K)); }
    AUKIT_DEV double eval(unsigned, double) const { return __builtin_fma(p2 - p1, fx, p1); }
};
template <> struct Row<AUKIT_INTERP_CUBIC, false> {  // 4 taps, Horner on fx = rem * RN(1/b)
    static constexpr int N = 4;
    double p0, p1, p2, p3;
    AUKIT_DEV void issue(unsigned tap, unsigned, unsigned) {
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24"
                     : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(tap));
    }
    template <int K> AUKIT_DEV void wait() { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "n"(K)); }
    AUKIT_DEV double eval(unsigned rem, double inv_b) const {
        const double fx = (double)rem * inv_b;
        const double c3 = __builtin_fma(1.5, p1 - p2, 0.5 * (p3 - p0));
        const double c2 = __builtin_fma(-0.5, p3, __builtin_fma(2.0, p2, __builtin_fma(-2.5, p1, p0)));
        const double c1 = 0.5 * (p2 - p0);
        return __builtin_fma(__builtin_fma(__builtin_fma(c3, fx, c2), fx, c1), fx, p1);
    }
};
template <> struct Row<AUKIT_INTERP_LINEAR, false> {
    static constexpr int N = 2;
    double p1, p2;
    AUKIT_DEV void issue(unsigned tap, unsigned, unsigned) { asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=&v"(p1), "=&v"(p2) : "v"(tap)); }
    template <int K> AUKIT_DEV void wait() { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(p1), "+v"(p2) : "n"(K)); }
    AUKIT_DEV double eval(unsigned rem, double inv_b) const { return __builtin_fma(p2 - p1, (double)rem * inv_b, p1); }
};

// the compiler-scheduled evaluation of one output (partial tiles: the last tile of a stream)
template <int INTERP, bool TAB>
AUKIT_DEV double eval_plain(const double *tab, const double *wt, unsigned b, double inv_b, unsigned q, unsigned rem) {
    const double p1 = tab[q], p2 = tab[q + 1];
    if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
        const double fx = TAB ? wt[rem] : (double)rem * inv_b;
        return __builtin_fma(p2 - p1, fx, p1);
    } else {
        const double p0 = tab[(int)q - 1], p3 = tab[q + 2];
        if constexpr (TAB) {
            const double2 w01 = reinterpret_cast<const double2 *>(wt)[rem], w23 = reinterpret_cast<const double2 *>(wt)[b + rem];
            return __builtin_fma(w23.y, p3, __builtin_fma(w23.x, p2, __builtin_fma(w01.y, p1, w01.x * p0)));
        } else {
            const double fx = (double)rem * inv_b;
            const double c3 = __builtin_fma(1.5, p1 - p2, 0.5 * (p3 - p0));
            const double c2 = __builtin_fma(-0.5, p3, __builtin_fma(2.0, p2, __builtin_fma(-2.5, p1, p0)));
            const double c1 = 0.5 * (p2 - p0);
            return __builtin_fma(__builtin_fma(__builtin_fma(c3, fx, c2), fx, c1), fx, p1);
        }
    }
}

// describe() of fast_wave_dev.h for a tile size that is a template parameter (F.wc / F.wd are computed for the same TILE on the host)
template <int TILE, int HL, int HR>
AUKIT_DEV WaveTile describe_t(const ResampleParams &P, const FastParams &F, unsigned t) {
    unsigned sidx, tin;
    if (P.tiles_per_seg) { sidx = t / P.tiles_per_seg; tin = t - sidx * P.tiles_per_seg; }
    else { sidx = as_const(P.tile_seg)[t]; tin = t - as_const(P.seg_tile0)[sidx]; }
    const Seg sg = load_seg(P.segs, sidx);
    WaveTile w;
    const unsigned o0 = tin * (unsigned)TILE;
    w.cnt = o0 < sg.n_out ? min((unsigned)TILE, sg.n_out - o0) : 0u;
    const unsigned td = tin * F.wd;  // (o0 * a) = (tin * wc + td / b) * b + td % b
    const unsigned tq = td / F.b;
    const unsigned kb = tin * F.wc + tq;
    w.r0 = td - tq * F.b;
    const unsigned klast = w.cnt ? (w.r0 + (w.cnt - 1) * F.a) / F.b : 0u;
    w.k_lo = 1 + (int)kb - HL;
    w.n_stage = (int)klast + 1 + HL + HR;
    w.w_lo = sg.w_lo;
    w.w_hi = sg.w_hi;
    w.base = P.src + (size_t)as_const(P.src_off)[sg.stream] + 2ll * sg.src_base;
    const unsigned char *a0 = w.base + 2ll * w.k_lo;
    w.al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
    w.head = (int)(a0 - w.al) / 2;
    w.nvec = (w.head + w.n_stage + 7) / 8;
    w.orow = reinterpret_cast<float *>(P.out) + sg.out_off + o0;
    return w;
}

constexpr int VMCNT0 = 0x0F70;  // s_waitcnt vmcnt(0) expcnt(7) lgkmcnt(15) on gfx9: every VMEM operation of the wave has completed

// previous lane's value (lane 0: `carry`) / lane 63's value of a double, in the VALU (two 32-bit DPP moves / two v_readlane)
AUKIT_DEV double prev_lane_d(double s, double carry) {
    const unsigned long long sb = (unsigned long long)__double_as_longlong(s), cb = (unsigned long long)__double_as_longlong(carry);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)cb, (int)(unsigned)sb, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(cb >> 32), (int)(unsigned)(sb >> 32), 0x138, 0xF, 0xF, false);
    return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}
AUKIT_DEV double last_lane_d(double s) {
    const unsigned long long sb = (unsigned long long)__double_as_longlong(s);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)sb, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(sb >> 32), 63);
    return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}

// EPI 0: Audio:resample (clamp to [-1, 1], :667-668).  EPI 1: aukit.stream.pcm's chunk sample (:2397-2403, Q2): the interpolated sample is NOT
// clamped, `ns = ls + lp_alpha * (s - ls)` with ls the RAW sample before it (0 at the start of an iterator call's chunk = of a segment),
// `clamp(ns * (ns < 0 and 128 or 127), -128, 127)` — every operation in fp64, the store in f32.  The window reaches one tap further left so
// that a tile can re-evaluate the raw sample before its first output.
template <int INTERP, int TILE, int NV, bool TAB, int EPI>
__global__ __launch_bounds__(256) void k_wave_f64(const ResampleParams P, const FastParams F, const double *__restrict__ wg, const unsigned wt_doubles,
                                                  const double inv_b, const double alpha) {
    extern __shared__ double smd[];  // [phase weights][4 × window of F.cap doubles][4 × raw tile of NV KiB]: ONE array (a second one makes hipcc drain every DMA early)
    constexpr int HL = (INTERP == AUKIT_INTERP_CUBIC ? 1 : 0) + (EPI ? 1 : 0), HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    auto finish = [&](double s, double prev) -> float {   // one output from its raw interpolated sample (EPI 1: and the raw sample before it)
        if constexpr (EPI == 0) return __builtin_amdgcn_fmed3f((float)s, -1.0f, 1.0f);   // :667-668
        else {
            const double ns = __builtin_fma(alpha, s - prev, prev);                       // :2401 (fused: this kernel keeps the arithmetic type, not the operation order)
            // ns * (ns < 0 and 128 or 127): the factor is 127.5 - copysign(0.5, ns) (a bit operation and one add instead of a compare and two
            // selects; -0 gets 128 instead of 127 and is -0 either way); clamp(…, -128, 127) AFTER the rounding to f32 — the bounds are f32
            // numbers and rounding is monotone: the same value as clamping the double first, in one v_med3_f32 instead of v_max_f64 + v_min_f64  :2402
            const double v = ns * (127.5 - __builtin_copysign(0.5, ns));
            return __builtin_amdgcn_fmed3f((float)v, -128.0f, 127.0f);
        }
    };
    constexpr int ROWS = TILE / 64;
    using R = Row<INTERP, TAB>;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (TAB) {  // the weights of the b phases: LDS, once per workgroup
        for (unsigned i = threadIdx.x; i < wt_doubles / 2; i += 256) reinterpret_cast<double2 *>(smd)[i] = reinterpret_cast<const double2 *>(wg)[i];
        __syncthreads();
    }
    const double *const wt = smd;
    double *const sm = smd + (TAB ? wt_doubles : 0u) + wave * (unsigned)F.cap;
    unsigned char *const raw = reinterpret_cast<unsigned char *>(smd + (TAB ? wt_doubles : 0u) + 4u * (unsigned)F.cap) + wave * (unsigned)(NV * 1024);
    const unsigned nwaves = gridDim.x * 4u;
    const unsigned lane_a = (unsigned)lane * F.a;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    // raw samples of a tile, global → LDS: lane l of instruction i brings the 16-byte vector l + 64 i (clamped into the allocation: slots
    // of vectors the tile does not have, or that straddle the allocation, are never converted / are patched below)
    auto dma = [&](const WaveTile &w) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = lane + 64 * i;
            const unsigned char *p = w.al + 16 * (size_t)v;
            const bool ok = v < w.nvec && p >= P.safe_lo && p + 16 <= P.safe_hi;
            __builtin_amdgcn_global_load_lds((const AUKIT_GLOBAL_AS void *)(ok ? p : P.safe_lo), (AUKIT_LDS_AS void *)(raw + 1024 * i), 16, 0, 0);
        }
    };
    WaveTile cur = describe_t<TILE, HL, HR>(P, F, t);
    dma(cur);
    __builtin_amdgcn_s_waitcnt(VMCNT0);
    const double sc_pos = 1.0 / 32767.0, sc_neg = 1.0 / 32768.0;
    for (;;) {
        // ---- 1. raw → window as doubles.  Instruction j: lane l converts the dword 64 j + l (two samples) and writes window slots
        // 128 j + 2 l, + 1 — consecutive lanes read consecutive dwords and write consecutive 16-byte pairs: conflict-free both ways.
        // (The first version had every lane convert its own 16-byte vector into 64 contiguous bytes: ds_write_b128 at a 64-byte lane
        // stride is a 4-way bank conflict, and the PMC pass showed a third of the kernel's LDS cycles to be conflicts — profiles/.)
        unsigned rw[NV * 4];
#pragma unroll
        for (int j = 0; j < NV * 4; j++) rw[j] = reinterpret_cast<const unsigned *>(raw)[64 * j + lane];
#pragma unroll
        for (int j = 0; j < NV * 4; j++) {
            const unsigned w = rw[j];
            const int s0 = (int)(short)(w & 0xFFFF), s1 = (int)w >> 16;
            reinterpret_cast<double2 *>(sm)[64 * j + lane] = make_double2((double)s0 * (s0 < 0 ? sc_neg : sc_pos), (double)s1 * (s1 < 0 ? sc_neg : sc_pos));  // s / (s < 0 and 32768 or 32767)  :1081
        }
        {
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            auto sample = [&](const unsigned char *q) { const short s = (short)(q[0] | q[1] << 8); return (double)s * (s < 0 ? sc_neg : sc_pos); };
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare: vectors that straddle the allocation
                for (int idx = lane; idx < cur.nvec * 8; idx += 64) {
                    const unsigned char *q = cur.al + (size_t)idx * 2;
                    const unsigned char *vb = cur.al + 16 * (size_t)(idx / 8);
                    if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) sm[idx] = (q >= P.safe_lo && q + 2 <= P.safe_hi) ? sample(q) : 0.0;
                }
            }
            // nil fall-backs of interpolate.{linear,cubic} (aukit.lua:259, :264) = replicated edge samples
            const int k_hi = cur.k_lo + cur.n_stage - 1;
            if (cur.k_lo < cur.w_lo) {
                const double e_lo = sample(cur.base + 2ll * cur.w_lo);
                for (int idx = lane; idx < cur.w_lo - cur.k_lo; idx += 64) sm[cur.head + idx] = e_lo;
            }
            if (k_hi > cur.w_hi) {
                const double e_hi = sample(cur.base + 2ll * cur.w_hi);
                const int first = cur.w_hi + 1 - cur.k_lo;
                for (int idx = lane; idx < k_hi - cur.w_hi; idx += 64) sm[cur.head + first + idx] = e_hi;
            }
        }
        // ---- 2. the next tile's raw samples (the conversion above has consumed this tile's: its ds_reads were waited for)
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe_t<TILE, HL, HR>(P, F, tn);
            dma(nxt);
        }
        const double *tab = sm + cur.head + HL;  // tab[q] = d[1 + kb + q]
        float *orow = cur.orow;
        const bool full = cur.cnt == (unsigned)TILE;  // wave-uniform
        float res[ROWS];
        [[maybe_unused]] double carry = 0.0;   // EPI 1: the raw sample before the tile's first output (0 at the start of a chunk)
        if constexpr (EPI == 1) {
            const bool first = (P.tiles_per_seg ? t % P.tiles_per_seg : t - as_const(P.seg_tile0)[as_const(P.tile_seg)[t]]) == 0;
            if (!first) {   // position n = r0 - a, one table step back when that is negative
                const bool back = cur.r0 < F.a;
                const unsigned n = back ? cur.r0 + F.b - F.a : cur.r0 - F.a;
                const unsigned q = __umulhi(n, F.magic), rem = n - q * F.b;
                carry = eval_plain<INTERP, TAB>(back ? tab - 1 : tab, wt, F.b, inv_b, q, rem);
            }
        }
        if (full) {
            // ---- 3. the rows, into registers
            const unsigned n0 = cur.r0 + lane_a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
            asm volatile("" ::: "memory");  // the staging stores above stay above
            const unsigned tap0 = lds_addr(tab) - (INTERP == AUKIT_INTERP_CUBIC ? 8u : 0u), wa0 = lds_addr(wt), wb0 = lds_addr(wt) + 16u * F.b;
            constexpr unsigned WSH = INTERP == AUKIT_INTERP_CUBIC ? 4 : 3;  // bytes per phase in the first weight array: 16 (w0, w1) / 8 (fx)
            R nx;
            nx.issue(tap0 + 8u * q, wa0 + (rem << WSH), wb0 + (rem << WSH));
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                R c = nx;
                const unsigned rem_c = rem;
                if (r + 1 < ROWS) {
                    rem += F.dr64;  // the same lane, one row (64 outputs) further
                    q += F.dq64;
                    const bool wrap = rem >= F.b;
                    rem -= wrap ? F.b : 0u;
                    q += wrap ? 1u : 0u;
                    nx.issue(tap0 + 8u * q, wa0 + (rem << WSH), wb0 + (rem << WSH));
                    c.template wait<R::N>();
                } else {
                    c.template wait<0>();
                }
                const double sv = c.eval(rem_c, inv_b);
                if constexpr (EPI == 0) res[r] = finish(sv, 0.0);
                else { const double pv = prev_lane_d(sv, carry); carry = last_lane_d(sv); res[r] = finish(sv, pv); }
            }
            asm volatile("" ::: "memory");  // the next tile's staging stores stay below
        }
        // ---- 4. the next tile's samples have landed (and the stores of the tile before this one have long completed).  On the straight
        // path of the loop body on purpose: inside the two branches, the structurizer's flow blocks leave hipcc a static path around
        // the wait, and its waitcnt pass then drains the DMA (and every store) at the top of the loop after all.
        __builtin_amdgcn_s_waitcnt(VMCNT0);
        unsigned full2 = __builtin_amdgcn_readfirstlane((unsigned)full);
        asm volatile("" : "+s"(full2));  // opaque: or jump threading fuses the two `if (full)` and the wait is back inside the branches
        if (full2) {
            // ---- 5. this tile's stores: in flight while the next tile is converted and evaluated
#pragma unroll
            for (int r = 0; r < ROWS; r++) orow[r * 64 + lane] = res[r];
        } else {
            for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
                const unsigned j = rb + lane;
                const unsigned n = cur.r0 + (j < cur.cnt ? j : cur.cnt - 1) * F.a;
                const unsigned q = __umulhi(n, F.magic);
                const unsigned rem = n - q * F.b;
                const double v = eval_plain<INTERP, TAB>(tab, wt, F.b, inv_b, q, rem);
                double pv = 0.0;
                if constexpr (EPI == 1) { pv = prev_lane_d(v, carry); carry = last_lane_d(v); }
                if (j < cur.cnt) orow[j] = finish(v, pv);
            }
        }
        if (!more) break;
        cur = nxt;
        t = tn;
    }
}

bool fast_eligible(int src_kind, int interp, double old_rate, double new_rate, FastParams &F);

// ---- the phases in registers (round 3).  Lane l of row r evaluates output 64 r + l of its tile, at phase ((64 r + l) a) mod b.  When the
// tile is a multiple of lcm(64, b) outputs, every tile of a segment starts at phase 0 and a lane meets only PH = b / gcd(b, 64) phases —
// rows r and r + PH share one (44.1 → 48 kHz: b = 160, PH = 5, tiles of 640).  The lane then keeps the weights of its PH phases and the tap
// offsets of its first PH rows in registers for the whole launch: a row reads its four taps from LDS (32 bytes where the phase-table kernel
// reads 64 — its LDS reads are what bound that one) and needs no position arithmetic at all (one add for the tap address).  Same weights
// (the host's table, read once per lane from HBM), same order of operations: bit-identical to k_wave_f64<…, TAB = true>.
template <int INTERP> struct PhaseW;
template <> struct PhaseW<AUKIT_INTERP_CUBIC> {
    double w0, w1, w2, w3;
    AUKIT_DEV void load(const double *wg, unsigned b, unsigned rem) { w0 = wg[2 * rem]; w1 = wg[2 * rem + 1]; w2 = wg[2 * b + 2 * rem]; w3 = wg[2 * b + 2 * rem + 1]; }
    AUKIT_DEV double eval(const Row<AUKIT_INTERP_CUBIC, false> &t) const { return __builtin_fma(w3, t.p3, __builtin_fma(w2, t.p2, __builtin_fma(w1, t.p1, w0 * t.p0))); }
};
template <> struct PhaseW<AUKIT_INTERP_LINEAR> {
    double fx;
    AUKIT_DEV void load(const double *wg, unsigned, unsigned rem) { fx = wg[rem]; }
    AUKIT_DEV double eval(const Row<AUKIT_INTERP_LINEAR, false> &t) const { return __builtin_fma(t.p2 - t.p1, fx, t.p1); }
};

#ifndef AUKIT_F64_REG_NQ
#define AUKIT_F64_REG_NQ 6
#endif
constexpr int REG_NQ_MAX = AUKIT_F64_REG_NQ;  // conversion rounds of 128 samples a window can take
#ifndef AUKIT_F64_REG_K5
#define AUKIT_F64_REG_K5 2          // tiles of 64 · 5 · K5 outputs when a lane meets five phases (A/B)
#endif
#ifdef AUKIT_F64_REG_WAVES
#define AUKIT_REG_OCC __attribute__((amdgpu_waves_per_eu(AUKIT_F64_REG_WAVES, 8)))
#else
#define AUKIT_REG_OCC
#endif

template <int INTERP, int PH, int K, int EPI>
__global__ __launch_bounds__(256) AUKIT_REG_OCC void k_wave_f64_reg(const ResampleParams P, const FastParams F, const double *__restrict__ wg, const unsigned nq, const unsigned qstep8,
                                                      const double inv_b, const double alpha) {
    extern __shared__ double smd[];  // [4 × window of 128 nq doubles][4 × raw tile of 256 nq bytes]
    constexpr int TILE = 64 * PH * K, ROWS = PH * K;
    constexpr int HL = (INTERP == AUKIT_INTERP_CUBIC ? 1 : 0) + (EPI ? 1 : 0), HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    auto finish = [&](double s, double prev) -> float {   // as in k_wave_f64
        if constexpr (EPI == 0) return __builtin_amdgcn_fmed3f((float)s, -1.0f, 1.0f);
        else {
            const double ns = __builtin_fma(alpha, s - prev, prev);
            const double v = ns * (127.5 - __builtin_copysign(0.5, ns));
            return __builtin_amdgcn_fmed3f((float)v, -128.0f, 127.0f);
        }
    };
    using R = Row<INTERP, false>;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *const sm = smd + wave * (128u * nq);
    unsigned char *const raw = reinterpret_cast<unsigned char *>(smd + 4u * 128u * nq) + wave * (256u * nq);
    const unsigned nwaves = gridDim.x * 4u;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    // this lane's PH phases: weights and the byte offset of the first tap
    PhaseW<INTERP> pw[PH];
    unsigned qo[PH];
#pragma unroll
    for (int p = 0; p < PH; p++) {
        const unsigned n = (unsigned)(64 * p + lane) * F.a;
        const unsigned q = __umulhi(n, F.magic), rem = n - q * F.b;
        qo[p] = 8u * q;
        pw[p].load(wg, F.b, rem);
    }
    auto dma = [&](const WaveTile &w) {
#pragma unroll
        for (int i = 0; i < (REG_NQ_MAX + 3) / 4; i++) {
            const int v = lane + 64 * i;
            if (v < 16 * (int)nq) {   // the raw area is 16 nq vectors: lanes beyond it stay out of the LDS write
                const unsigned char *p = w.al + 16 * (size_t)v;
                const bool ok = v < w.nvec && p >= P.safe_lo && p + 16 <= P.safe_hi;
                __builtin_amdgcn_global_load_lds((const AUKIT_GLOBAL_AS void *)(ok ? p : P.safe_lo), (AUKIT_LDS_AS void *)(raw + 1024 * i), 16, 0, 0);
            }
        }
    };
    WaveTile cur = describe_t<TILE, HL, HR>(P, F, t);
    dma(cur);
    __builtin_amdgcn_s_waitcnt(VMCNT0);
    const double sc_pos = 1.0 / 32767.0, sc_neg = 1.0 / 32768.0;
    for (;;) {
        // ---- 1. raw → window as doubles (k_wave_f64's conflict-free mapping), nq rounds of 64 dwords
        unsigned rw[REG_NQ_MAX];
#pragma unroll
        for (int j = 0; j < REG_NQ_MAX; j++) if (j < (int)nq) rw[j] = reinterpret_cast<const unsigned *>(raw)[64 * j + lane];
#pragma unroll
        for (int j = 0; j < REG_NQ_MAX; j++) {
            if (j < (int)nq) {
                const unsigned w = rw[j];
                const int s0 = (int)(short)(w & 0xFFFF), s1 = (int)w >> 16;
                reinterpret_cast<double2 *>(sm)[64 * j + lane] = make_double2((double)s0 * (s0 < 0 ? sc_neg : sc_pos), (double)s1 * (s1 < 0 ? sc_neg : sc_pos));  // :1081
            }
        }
        {
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            auto sample = [&](const unsigned char *q) { const short s = (short)(q[0] | q[1] << 8); return (double)s * (s < 0 ? sc_neg : sc_pos); };
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare: vectors that straddle the allocation
                for (int idx = lane; idx < cur.nvec * 8; idx += 64) {
                    const unsigned char *q = cur.al + (size_t)idx * 2;
                    const unsigned char *vb = cur.al + 16 * (size_t)(idx / 8);
                    if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) sm[idx] = (q >= P.safe_lo && q + 2 <= P.safe_hi) ? sample(q) : 0.0;
                }
            }
            const int k_hi = cur.k_lo + cur.n_stage - 1;   // nil fall-backs of interpolate.{linear,cubic} (:259, :264) = replicated edge samples
            if (cur.k_lo < cur.w_lo) {
                const double e_lo = sample(cur.base + 2ll * cur.w_lo);
                for (int idx = lane; idx < cur.w_lo - cur.k_lo; idx += 64) sm[cur.head + idx] = e_lo;
            }
            if (k_hi > cur.w_hi) {
                const double e_hi = sample(cur.base + 2ll * cur.w_hi);
                const int first = cur.w_hi + 1 - cur.k_lo;
                for (int idx = lane; idx < k_hi - cur.w_hi; idx += 64) sm[cur.head + first + idx] = e_hi;
            }
        }
        // ---- 2. the next tile's raw samples
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe_t<TILE, HL, HR>(P, F, tn);
            dma(nxt);
        }
        const double *tab = sm + cur.head + HL;  // tab[q] = d[1 + kb + q]
        float *orow = cur.orow;
        const bool full = cur.cnt == (unsigned)TILE;  // wave-uniform
        float res[ROWS];
        [[maybe_unused]] double carry = 0.0;   // EPI 1: the raw sample before the tile's first output (0 at the start of a chunk)
        if constexpr (EPI == 1) {
            const bool first = (P.tiles_per_seg ? t % P.tiles_per_seg : t - as_const(P.seg_tile0)[as_const(P.tile_seg)[t]]) == 0;
            if (!first) carry = eval_plain<INTERP, true>(tab - 1, wg, F.b, inv_b, 0u, F.b - F.a);   // a tile starts at phase 0: one table step back, phase b - a (a < b)
        }
        if (full) {
            // ---- 3. the rows, into registers: row r = phase r mod PH, taps (r / PH) · qstep further
            asm volatile("" ::: "memory");  // the staging stores above stay above
            const unsigned tap0 = lds_addr(tab) - (INTERP == AUKIT_INTERP_CUBIC ? 8u : 0u);
            R nx;
            nx.issue(tap0 + qo[0], 0u, 0u);
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                R c = nx;
                if (r + 1 < ROWS) {
                    nx.issue(tap0 + qo[(r + 1) % PH] + (unsigned)((r + 1) / PH) * qstep8, 0u, 0u);
                    c.template wait<R::N>();
                } else {
                    c.template wait<0>();
                }
                const double sv = pw[r % PH].eval(c);
                if constexpr (EPI == 0) res[r] = finish(sv, 0.0);
                else { const double pv = prev_lane_d(sv, carry); carry = last_lane_d(sv); res[r] = finish(sv, pv); }
            }
            asm volatile("" ::: "memory");  // the next tile's staging stores stay below
        }
        // ---- 4. the next tile's samples have landed (k_wave_f64 explains why the wait is here, on the straight path)
        __builtin_amdgcn_s_waitcnt(VMCNT0);
        unsigned full2 = __builtin_amdgcn_readfirstlane((unsigned)full);
        asm volatile("" : "+s"(full2));
        if (full2) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) orow[r * 64 + lane] = res[r];
        } else {
            for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
                const unsigned j = rb + lane;
                const unsigned n = cur.r0 + (j < cur.cnt ? j : cur.cnt - 1) * F.a;
                const unsigned q = __umulhi(n, F.magic);
                const unsigned rem = n - q * F.b;
                const double v = eval_plain<INTERP, true>(tab, wg, F.b, inv_b, q, rem);
                double pv = 0.0;
                if constexpr (EPI == 1) { pv = prev_lane_d(v, carry); carry = last_lane_d(v); }
                if (j < cur.cnt) orow[j] = finish(v, pv);
            }
        }
        if (!more) break;
        cur = nxt;
        t = tn;
    }
}

template <int INTERP, int EPI>
static void launch_wf64_reg(int ph, const ResampleParams &P, const FastParams &F, const double *wg, unsigned nq, unsigned qstep8, double inv_b, double alpha, size_t lds,
                            unsigned grid, hipStream_t st) {
    if (ph == 1) hipLaunchKernelGGL((k_wave_f64_reg<INTERP, 1, 8, EPI>), dim3(grid), dim3(256), lds, st, P, F, wg, nq, qstep8, inv_b, alpha);
    else if (ph == 3) hipLaunchKernelGGL((k_wave_f64_reg<INTERP, 3, 3, EPI>), dim3(grid), dim3(256), lds, st, P, F, wg, nq, qstep8, inv_b, alpha);
    else hipLaunchKernelGGL((k_wave_f64_reg<INTERP, 5, AUKIT_F64_REG_K5, EPI>), dim3(grid), dim3(256), lds, st, P, F, wg, nq, qstep8, inv_b, alpha);
}

template <int INTERP, int TILE, bool TAB, int EPI>
static void launch_wf64(int nv, const ResampleParams &P, const FastParams &F, const double *wg, unsigned wtd, double inv_b, double alpha, size_t lds, unsigned grid, hipStream_t st) {
    switch (nv) {
    case 1: hipLaunchKernelGGL((k_wave_f64<INTERP, TILE, 1, TAB, EPI>), dim3(grid), dim3(256), lds, st, P, F, wg, wtd, inv_b, alpha); break;
    case 2: hipLaunchKernelGGL((k_wave_f64<INTERP, TILE, 2, TAB, EPI>), dim3(grid), dim3(256), lds, st, P, F, wg, wtd, inv_b, alpha); break;
    default: hipLaunchKernelGGL((k_wave_f64<INTERP, TILE, 4, TAB, EPI>), dim3(grid), dim3(256), lds, st, P, F, wg, wtd, inv_b, alpha); break;
    }
}
template <int TILE, int EPI>
static void launch_wf64_tile(int interp, bool tab, int nv, const ResampleParams &P, const FastParams &F, const double *wg, unsigned wtd, double inv_b, double alpha, size_t lds,
                             unsigned grid, hipStream_t st) {
    if (tab) { if (interp == AUKIT_INTERP_LINEAR) launch_wf64<AUKIT_INTERP_LINEAR, TILE, true, EPI>(nv, P, F, wg, wtd, inv_b, alpha, lds, grid, st); else launch_wf64<AUKIT_INTERP_CUBIC, TILE, true, EPI>(nv, P, F, wg, wtd, inv_b, alpha, lds, grid, st); }
    else { if (interp == AUKIT_INTERP_LINEAR) launch_wf64<AUKIT_INTERP_LINEAR, TILE, false, EPI>(nv, P, F, wg, wtd, inv_b, alpha, lds, grid, st); else launch_wf64<AUKIT_INTERP_CUBIC, TILE, false, EPI>(nv, P, F, wg, wtd, inv_b, alpha, lds, grid, st); }
}

// weights of the b phases, fx = rem / b, from the reference's polynomial (aukit.lua:265) regrouped by tap; computed in long double
// (64-bit mantissa on this host) and rounded once.  Layout: cubic [b] × (w0, w1) then [b] × (w2, w3); linear [b] × fx (padded to even).
static void phase_weights(unsigned b, int interp, std::vector<double> &w) {
    if (interp == AUKIT_INTERP_LINEAR) {
        w.assign((b + 1) & ~1u, 0.0);
        for (unsigned r = 0; r < b; r++) w[r] = (double)((long double)r / (long double)b);
        return;
    }
    w.assign((size_t)4 * b, 0.0);
    for (unsigned r = 0; r < b; r++) {
        const long double f = (long double)r / (long double)b, f2 = f * f, f3 = f2 * f;
        w[2 * r] = (double)(-0.5L * f3 + f2 - 0.5L * f);
        w[2 * r + 1] = (double)(1.5L * f3 - 2.5L * f2 + 1.0L);
        w[2 * b + 2 * r] = (double)(-1.5L * f3 + 2.0L * f2 + 0.5L * f);
        w[2 * b + 2 * r + 1] = (double)(0.5L * f3 - 0.5L * f2);
    }
}

// returns true when this kernel took the launch (*rc = its status): 16-bit signed little-endian mono → linear / cubic → f32, integer rates
// epi 1: aukit.stream.pcm's epilogue (alpha = lp_alpha, :2365) — tiles of 512 only
bool wave_f64_try(aukit_ctx *ctx, int src_kind, int interp, double old_rate, double new_rate, const std::vector<Seg> &segs, ResampleParams &P,
                  uint64_t algorithmic_bytes, int *rc, int epi, double alpha) {
    if (src_kind != SRC_PCM_S16LE_MONO) return false;
    FastParams F;
    if (!fast_eligible(SRC_PCM_S16LE_MONO, interp, old_rate, new_rate, F)) return false;
    for (const Seg &g : segs)
        if (g.w_hi < g.w_lo && g.n_out) return false;
    {   // the phases in registers, when a lane meets at most five of them (k_wave_f64_reg): up-sampling with b = 2^i, 3 · 2^i or 5 · 2^i, i <= 6
        const char *er = getenv("AUKIT_F64_REGS");
        unsigned g = F.b, h = 64;
        while (h) { const unsigned r = g % h; g = h; h = r; }
        const unsigned ph = F.b / g;
        if (!(er && atoi(er) == 0) && !getenv("AUKIT_F64_TILE") && !getenv("AUKIT_F64_HORNER") && F.a <= F.b && F.b <= 512 && (ph == 1 || ph == 3 || ph == 5)) {
            const int tile = ph == 1 ? 512 : (ph == 3 ? 576 : 320 * AUKIT_F64_REG_K5);
            const int hl = (interp == AUKIT_INTERP_CUBIC ? 1 : 0) + (epi ? 1 : 0), hr = interp == AUKIT_INTERP_CUBIC ? 2 : 1;
            const int win = (int)(((unsigned long long)(tile - 1) * F.a) / F.b) + 2 + hl + hr;
            const unsigned nq = (unsigned)(win + 16 + 127) / 128;   // + 16: the window starts at a 16-byte boundary (up to 7 samples early), as nv above
            uint64_t max_tiles = 0;
            for (const Seg &sg : segs) max_tiles = std::max<uint64_t>(max_tiles, (sg.n_out + tile - 1) / tile);
            F.wc = (unsigned)(((unsigned long long)tile * F.a) / F.b);
            F.wd = (unsigned)(((unsigned long long)tile * F.a) % F.b);
            const bool fits = nq <= (unsigned)REG_NQ_MAX && F.wd == 0 && ((double)max_tiles + 1) * (double)F.wc < 2147483648.0 &&
                              ((double)F.b + (double)tile * (double)F.a) * (double)F.b < 4294967296.0;
            if (fits) {
                F.cap = 128 * (int)nq;
                F.dq64 = (unsigned)((64ull * F.a) / F.b);
                F.dr64 = (unsigned)((64ull * F.a) % F.b);
                if (ctx->wt_b != F.b || ctx->wt_interp != interp) {
                    std::vector<double> w;
                    phase_weights(F.b, interp, w);
                    if ((*rc = upload_table(ctx, ctx->wt_buf, w.data(), w.size() * sizeof(double)))) return true;
                    ctx->wt_b = F.b; ctx->wt_interp = interp; ctx->wt_doubles = (unsigned)w.size();
                }
                const size_t lds = (size_t)nq * (4 * 128 * 8 + 4 * 256);
                if ((*rc = plan_tiles_sized(ctx, segs, tile, P))) return true;
                if (P.n_tiles == 0) { *rc = AUKIT_OK; return true; }
                unsigned per_cu = 128;   // 4 / 8 / 16 / 32 / 64 / 128 / 256 measured 2.15 / 2.22 / 2.30 / 2.26 / 2.06 / 2.04 / 2.05 ms on config T (four workgroups are resident)
                if (const char *e = getenv("AUKIT_FAST_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }
                const unsigned grid = std::min<unsigned>((P.n_tiles + 3) / 4, (unsigned)ctx->num_cus * per_cu);
                if ((*rc = ctx_begin_kernel(ctx))) return true;
                const double *wg = reinterpret_cast<const double *>(ctx->wt_buf.p);
                const unsigned qstep8 = 8u * (unsigned)((64ull * ph * F.a) / F.b);   // lcm(64, b) outputs further: a whole number of source samples
                const double inv_b = 1.0 / (double)F.b;
                if (interp == AUKIT_INTERP_LINEAR) {
                    if (epi) launch_wf64_reg<AUKIT_INTERP_LINEAR, 1>((int)ph, P, F, wg, nq, qstep8, inv_b, alpha, lds, grid, ctx->stream);
                    else launch_wf64_reg<AUKIT_INTERP_LINEAR, 0>((int)ph, P, F, wg, nq, qstep8, inv_b, 0.0, lds, grid, ctx->stream);
                } else {
                    if (epi) launch_wf64_reg<AUKIT_INTERP_CUBIC, 1>((int)ph, P, F, wg, nq, qstep8, inv_b, alpha, lds, grid, ctx->stream);
                    else launch_wf64_reg<AUKIT_INTERP_CUBIC, 0>((int)ph, P, F, wg, nq, qstep8, inv_b, 0.0, lds, grid, ctx->stream);
                }
                if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_wave_f64_reg launch failed"); return true; }
                static thread_local char nmr[96];
                snprintf(nmr, sizeof nmr, "k_wave_f64<pcm_s16le_mono,%s,tile%d,phase_regs%s>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", tile, epi ? ",stream_pcm" : "");
                *rc = ctx_end_kernel(ctx, nmr, algorithmic_bytes);
                return true;
            }
        }
    }
    int tile = 512;  // outputs per wave tile: 512 keeps six workgroups (24 waves) per CU next to their windows; 1024 keeps three
    if (const char *e = getenv("AUKIT_F64_TILE")) tile = (atoi(e) == 1024 && !epi) ? 1024 : 512;
    const int spv = 8;
    const int hl = (interp == AUKIT_INTERP_CUBIC ? 1 : 0) + (epi ? 1 : 0), hr = interp == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int win = (int)(((unsigned long long)(tile - 1) * F.a) / F.b) + 2 + hl + hr;
    int nv = (win + 2 * spv + 64 * spv - 1) / (64 * spv);
    nv = nv <= 1 ? 1 : (nv <= 2 ? 2 : (nv <= 4 ? 4 : 0));
    if (!nv) return false;
    uint64_t max_tiles = 0;
    for (const Seg &g : segs) max_tiles = std::max<uint64_t>(max_tiles, (g.n_out + tile - 1) / tile);
    F.wc = (unsigned)(((unsigned long long)tile * F.a) / F.b);
    F.wd = (unsigned)(((unsigned long long)tile * F.a) % F.b);
    if ((double)max_tiles * (double)F.wd >= 4294967296.0 || ((double)max_tiles + 1) * (double)F.wc >= 2147483648.0) return false;
    if (((double)F.b + (double)tile * (double)F.a) * (double)F.b >= 4294967296.0) return false;
    F.cap = nv * 64 * spv;  // doubles per wave window
    F.dq64 = (unsigned)((64ull * F.a) / F.b);
    F.dr64 = (unsigned)((64ull * F.a) % F.b);
    bool tab = F.b <= 512;
    if (const char *e = getenv("AUKIT_F64_HORNER")) tab = tab && atoi(e) == 0;  // A/B: the Horner form for every ratio
    unsigned wtd = 0;
    if (tab) {
        if (ctx->wt_b != F.b || ctx->wt_interp != interp) {
            std::vector<double> w;
            phase_weights(F.b, interp, w);
            if ((*rc = upload_table(ctx, ctx->wt_buf, w.data(), w.size() * sizeof(double)))) return true;
            ctx->wt_b = F.b; ctx->wt_interp = interp; ctx->wt_doubles = (unsigned)w.size();
        }
        wtd = ctx->wt_doubles;
    }
    const size_t lds = ((size_t)F.cap * 4 + wtd) * 8 + (size_t)4 * nv * 1024;
    if (lds > 64 * 1024) return false;
    if ((*rc = plan_tiles_sized(ctx, segs, tile, P))) return true;
    if (P.n_tiles == 0) { *rc = AUKIT_OK; return true; }
    unsigned per_cu = 64;  // workgroups per CU in the grid (6 are resident): 16 → 64 measured +4 % (finer hand-out of the tail across CUs / XCDs)
    if (const char *e = getenv("AUKIT_FAST_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }
    const unsigned grid = std::min<unsigned>((P.n_tiles + 3) / 4, (unsigned)ctx->num_cus * per_cu);
    if ((*rc = ctx_begin_kernel(ctx))) return true;
    const double *wg = reinterpret_cast<const double *>(ctx->wt_buf.p);
    const double inv_b = 1.0 / (double)F.b;
    if (epi) launch_wf64_tile<512, 1>(interp, tab, nv, P, F, wg, wtd, inv_b, alpha, lds, grid, ctx->stream);
    else if (tile == 1024) launch_wf64_tile<1024, 0>(interp, tab, nv, P, F, wg, wtd, inv_b, 0.0, lds, grid, ctx->stream);
    else launch_wf64_tile<512, 0>(interp, tab, nv, P, F, wg, wtd, inv_b, 0.0, lds, grid, ctx->stream);
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_wave_f64 launch failed"); return true; }
    static thread_local char nm[96];
    snprintf(nm, sizeof nm, "k_wave_f64<pcm_s16le_mono,%s,tile%d,nv%d,%s%s>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", tile, nv, tab ? "phase_table" : "horner", epi ? ",stream_pcm" : "");
    *rc = ctx_end_kernel(ctx, nm, algorithmic_bytes);
    return true;
}

}  // namespace aukit
