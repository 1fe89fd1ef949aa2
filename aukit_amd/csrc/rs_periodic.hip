// rs_periodic.hip — k_rsp: the tile chain of flac_tail.hip (`aukit.<loader>(d):resample(48000, "cubic")` then `aukit.effects.lowpass / highpass`
// in ONE pass over the decoder's int16 rows, aukit.lua:648-680, :3586-3618, and `Audio:mono` :682-687 behind it) for the ratios every benchmark
// configuration has: 44.1 kHz and 22.05 kHz -> 48 kHz, a / b = 147 / 160 and 147 / 320.  (k_rs_onepole at 26 - 33 VALU instructions
// per output, the VALU and the LDS pipe both two thirds busy.)
//
// What the general kernel pays per output and this one does not:
//   * (q, rem) of every output and the two LDS addresses they make (7 VALU): 320 outputs advance the source by a whole number of samples when
//     320 a = 0 (mod b), so a lane that takes outputs 5 lane .. 5 lane + 4 of every 320 meets the SAME five phases and the same tap offsets for the
//     whole row — computed once, in front of the tile loop;
//   * the weights' LDS read (a ds_read_b128 per output: a third of the kernel's LDS time): the lane's 5 x 4 weights live in registers;
//   * four tap reads per output: a lane's five outputs are consecutive, their taps overlap — 6 or 8 floats per lane feed all five.  Output i's first tap
//     sits DM[i] or DM[i] + 1 floats into them (which of the two is the lane's constant): instead of selecting taps, the weights are stored shifted —
//     five per output, the unused end 0 — and `0 * tap` joins the sum as its first or last term: w0 p0 + w1 p1 + w2 p2 + w3 p3 in k_rs_onepole's
//     order, bit for bit (an exact zero added to the front or the back of a chain of FMAs);
//   * the trip through LDS from "lane = output mod 64" (interpolation) to "lane = 5 consecutive outputs" (recurrence): both are the second here; the
//     previous output's sample (the high-pass needs x[n - 1]) comes from the lane before by one DPP move;
//   * the results leave through an LDS transpose whose lane stride (5 floats) is odd: no skewed addresses, no bank conflicts either way.
// The window of 640 outputs is staged at once (16 int16 = 32 bytes per lane: 38 of 64 lanes at 44.1 kHz — a window of 320 would use 19 — and the
// conversion costs what the wave costs, not what the busy lanes cost), then two sub-tiles of 320 run from it.
// Everything else is k_rs_onepole's: the window requested a tile ahead by an asm load the compiler does not see (tests/test_isa_schedule.py walks
// this file's code too), first / last tiles and `AUKIT_RS_NOVEC` element by element, frame-by-frame rows (FLAC's int16 finals), runs of tiles with
// a warm-up (segs / warm), the two-channel mean (NW = 2), the row maxima for effects.normalize.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include "rs_onepole_dev.h"

namespace aukit {

// (A/B builds: -DAUKIT_RSP_NOSTORE times the kernel without its global stores — results wrong —, -DAUKIT_RSP_NT stores non-temporally)
#if defined(AUKIT_RSP_NOSTORE)
#define RSP_STORE(p, v) do { if (P.n == 0xFFFFFFFFu) *(p) = (v); } while (0)
#elif defined(AUKIT_RSP_NT)
#define RSP_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define RSP_STORE(p, v) (*(p) = (v))
#endif

// the tap patterns built: DM[i] = floor(i a / b) for i < 5, NT = DM[4] + 5 floats per lane
template <int PAT> AUKIT_DEV constexpr int rsp_dm(int i) {
    if constexpr (PAT == 1) return i == 0 ? 0 : i - 1;            // 3/4 <= a / b < 1      (44.1 kHz: 147 / 160)
    else return i < 3 ? 0 : 1;                                     // 1/3 <= a / b < 1/2    (22.05 kHz: 147 / 320)
}
template <int PAT> constexpr int rsp_nt() { return PAT == 1 ? 8 : 6; }

// S: int16 rows (the loaders' audios), or int8 (JOBS: stream.qoa's decoded chunks).  JOBS: a workgroup's item is a job of stream.qoa's tail (stream_tail.h,
// aukit.lua:3312-3330: interpolate -> clamp -> low-pass seeded with the history sample -> chunk sample) instead of a row of an audio — its own table behind
// the history sample (table index 0), its own outputs, no row maxima: k_rs_onepole<..., JOBS>'s contract.
template <bool HP, bool R32, int NW, int PAT, typename S = short, bool JOBS = false>
__global__ __launch_bounds__(64 * NW, 4) void k_rsp(const RsOnepoleParams P) {
    using RT = std::conditional_t<R32, float, double>;
    extern __shared__ __attribute__((aligned(16))) float rsm_all[];
    constexpr int E = 5, T = 64 * E, NSUB = PAT == 1 ? 2 : 4, TT = NSUB * T, NT = rsp_nt<PAT>();   // NSUB sub-tiles from one window of about 590 source samples
    const unsigned wv = NW > 1 ? (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;   // the wave = the channel
    float *const rsm = rsm_all + (NW > 1 ? wv * (unsigned)P.wave_lds : 0u);
    float *const win = rsm + 16;                             // P.cap floats in all: 16 of slack in front, the window, 16 behind
    float *const ob = rsm_all + NW * P.wave_lds;             // [sub-tile parity][wave][T]: the results on their way out
    const int lane = (int)(threadIdx.x & 63u);
    const S *const rows_s = reinterpret_cast<const S *>(P.rows);
    constexpr int EPV = 16, VB = EPV * (int)sizeof(S);       // a lane's vector: 16 elements = 32 (int16) or 16 (int8) bytes, aligned to that
    struct VT { u32x4g lo, hi; };
    // ---- the recurrence's constants: as k_rs_onepole's (slope m, its powers in scalar registers, the scan's factors)
    const double m = HP ? P.coef : 1.0 - P.coef;
    double mpd[E + 1];
    mpd[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= E; i++) mpd[i] = mpd[i - 1] * m;
    double Mdd[6];       // M^1, M^2, M^4 ... M^32 with M = m^E
    Mdd[0] = mpd[E];
#pragma unroll
    for (int k = 1; k < 6; k++) Mdd[k] = Mdd[k - 1] * Mdd[k - 1];
    auto uni = [](double v) -> RT {
        if constexpr (R32) return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)v)));
        else {
            const long long b = __double_as_longlong(v);
            return __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b));
        }
    };
    RT mp[E + 1], Md[6];
    mp[0] = (RT)1;
#pragma unroll
    for (int i = 1; i <= E; i++) mp[i] = uni(mpd[i]);
#pragma unroll
    for (int k = 0; k < 6; k++) Md[k] = uni(Mdd[k]);
    const RT coef_r = (RT)P.coef;
    auto fmaR = [](RT a, RT b, RT c) -> RT { if constexpr (R32) return __builtin_fmaf(a, b, c); else return __builtin_fma(a, b, c); };
    const RT mA = (RT)((lane & 16) ? pow(mpd[E], (double)((lane & 15) + 1)) : 0.0);
    const RT mB = (RT)(lane >= 32 ? pow(mpd[E], (double)(lane - 31)) : 0.0);
    const RT mlane = (RT)pow(mpd[E], (double)(lane + 1));
    // ---- the lane's five phases: weights (shifted, see above) and the offset of its first tap in a sub-tile's window
    const unsigned half_adv = (unsigned)(((unsigned long long)T * P.fa) / P.fb);   // source samples per sub-tile (exact: the host checked 320 a = 0 mod b)
    unsigned q0;
    float W[E][5];
    {
        const unsigned nn = (unsigned)(E * lane) * P.fa;
        q0 = nn / P.fb;
        unsigned rem = nn - q0 * P.fb, q = q0;
#pragma unroll
        for (int i = 0; i < E; i++) {
            const bool c = (int)(q - q0) != rsp_dm<PAT>(i);     // the first tap one float further on
            const float4 w = *reinterpret_cast<const float4 *>(P.wg + 4 * (size_t)rem);
            W[i][0] = c ? 0.f : w.x; W[i][1] = c ? w.x : w.y; W[i][2] = c ? w.y : w.z; W[i][3] = c ? w.z : w.w; W[i][4] = c ? w.w : 0.f;
            rem += P.fa;
            if (rem >= P.fb) { rem -= P.fb; q++; }
        }
    }
    const int e0 = lane * E;
    // ---- this workgroup's row(s) and run of tiles
    const unsigned item = blockIdx.x;
    unsigned r = 0, seg = 0, s = 0, c = 0;
    unsigned long long nout = 0, obase = 0;
    int L = 0, hist = 0;
    const S *row = rows_s;
    if constexpr (JOBS) {
        const TailJob jb = P.jobs[item];
        c = wv;
        nout = (unsigned long long)jb.nout; obase = jb.out_off; L = jb.n;
        row = rows_s + jb.src_off + (unsigned long long)c * jb.src_cstride;
        if (jb.last_off != ~0ull) hist = (int)rows_s[jb.last_off + (unsigned long long)c * jb.last_cstride];
    } else {
        r = NW > 1 ? (item / (unsigned)P.segs) * (unsigned)NW + wv : item / (unsigned)P.segs;
        seg = item % (unsigned)P.segs; s = r / (unsigned)P.C; c = r - s * (unsigned)P.C;
        nout = P.a_meta[s]; obase = P.a_meta[P.n + s] + (NW > 1 ? 0ull : (unsigned long long)c * P.a_meta[2 * (size_t)P.n + s]);
        L = (int)P.row_len[r];
        row = rows_s + P.row_off[r];
    }
    float *const orow = P.out + obase;
    float mxf = 0.f;
    RT carry_y = 0;
    float carry_x = 0.f;
    if constexpr (JOBS) carry_y = (RT)(double)((float)hist * (hist < 0 ? P.scale_neg : P.scale));   // ls = last[2]  (:3316)
    // frame-by-frame rows: as k_rs_onepole (flac_tail.hip)
    const int bsn = P.frames ? P.bs0[s] : 0;
    const unsigned nfr_s = P.frames ? (unsigned)(P.fbase[s + 1] - P.fbase[s]) : 0u;
    const FrameRec *const fr_s = P.frames ? P.frames + P.fbase[s] : nullptr;
    long long base0 = 0, base1 = 0, base2 = 0;
    int bound = 0x7FFFFFFF;
    unsigned fcur = 0;
    auto rec_base = [&](unsigned f) -> long long {
        const __attribute__((address_space(4))) unsigned long long *q = (const __attribute__((address_space(4))) unsigned long long *)(fr_s + f);
        const unsigned long long scratch = q[1];
        const int bs = (int)(unsigned)q[2];
        return (long long)P.fr_mul * (long long)scratch + (long long)c * bs - (long long)f * bsn;
    };
    if (P.frames && L > 0 && nfr_s > 0) {
        base0 = rec_base(0);
        bound = bsn;
        base1 = nfr_s > 1 ? rec_base(1) : base0;
        base2 = nfr_s > 2 ? rec_base(2) : base1;
    }
    auto tile_nst = [&](int cn) { return (int)__umulhi(((unsigned)cn - 1u) * P.fa, P.fmagic) + 5; };   // table indices kb .. kb + nst - 1 (a tile starts on phase 0): the last output's four taps and one more — the float a lane reads for a weight of 0 must be a number
    struct VecDesc { int on, nvA, nvB, hA, dposB, loB; };
    auto fetch = [&](unsigned kk, int nst, VT &pva, VecDesc &vd) {
        if (P.frames && L > 0) {
            const unsigned k0 = kk < 1u ? 1u : (kk > (unsigned)L ? (unsigned)L : kk);
            while ((int)(k0 - 1u) >= bound && fcur + 1u < nfr_s) {
                fcur++;
                base0 = base1; base1 = base2; bound += bsn;
                if (fcur + 2u < nfr_s) base2 = rec_base(fcur + 2u);
            }
        }
        vd.on = 0; vd.nvA = vd.nvB = vd.hA = vd.dposB = vd.loB = 0;
        const int x0 = (int)kk - 1, x1 = x0 + nst;   // row elements [x0, x1)
        uintptr_t va = 0;
        if (L > 0 && x0 >= EPV && x1 + EPV <= L && !P.novec) {
            const int xeA = P.frames ? (x1 < bound ? x1 : bound) : x1;
            const S *const pA = P.frames ? rows_s + (base0 + (long long)x0) : row + x0;
            const uintptr_t aA = (uintptr_t)pA, alA = aA & ~(uintptr_t)(VB - 1);
            const int hA = (int)((aA - alA) / sizeof(S));
            const int nvA = (hA + (xeA - x0) + EPV - 1) / EPV;
            uintptr_t alB = alA;
            int nvB = 0, hB = 0;
            if (x1 > xeA) {
                const uintptr_t aB = (uintptr_t)(rows_s + (base1 + (long long)xeA));
                alB = aB & ~(uintptr_t)(VB - 1);
                hB = (int)((aB - alB) / sizeof(S));
                nvB = (hB + (x1 - xeA) + EPV - 1) / EPV;
            }
            if (nvA + nvB <= 64) {
                vd.on = 1; vd.nvA = nvA; vd.nvB = nvB; vd.hA = hA; vd.dposB = (xeA - x0) - hB; vd.loB = xeA - x0;
                const int sb = lane - nvA;
                const bool inA = lane < nvA, inB = !inA && sb < nvB;
                va = inB ? alB + (unsigned)VB * (unsigned)sb : alA + (inA ? (unsigned)VB * (unsigned)lane : 0u);
            }
        }
        {   // the request: in no arm of a fork, into the registers the value keeps (flac_tail.hip has the story)
            unsigned long long save;
            const unsigned on = (unsigned)vd.on;
            if constexpr (sizeof(S) == 2)
                asm volatile("v_cmp_ne_u32_e32 vcc, 0, %3\n\ts_and_saveexec_b64 %2, vcc\n\tglobal_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\ts_mov_b64 exec, %2"
                             : "+v"(pva.lo), "+v"(pva.hi), "=&s"(save) : "v"(on), "v"(va) : "vcc", "memory");
            else
                asm volatile("v_cmp_ne_u32_e32 vcc, 0, %2\n\ts_and_saveexec_b64 %1, vcc\n\tglobal_load_dwordx4 %0, %3, off\n\ts_mov_b64 exec, %1"
                             : "+v"(pva.lo), "=&s"(save) : "v"(on), "v"(va) : "vcc", "memory");
        }
    };
    VT pva{};
    auto landed = [](VT &v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v.lo), "+v"(v.hi) : : "memory"); };
    VecDesc vcur{0, 0, 0, 0, 0, 0}, vnxt{0, 0, 0, 0, 0, 0};
    const unsigned nsegs = JOBS ? 1u : (unsigned)P.segs;
    const unsigned long long ntiles = (nout + TT - 1) / TT;
    const unsigned long long t_lo = ntiles * seg / nsegs, t_hi = ntiles * (seg + 1u) / nsegs;
    const unsigned long long t_in = (!JOBS && t_lo > (unsigned long long)P.warm) ? t_lo - (unsigned long long)P.warm : 0ull;
    const unsigned long long o_lo = t_lo * TT, o_end = t_hi * TT < nout ? t_hi * TT : nout;
    unsigned kb = (unsigned)(t_in * ((unsigned long long)NSUB * half_adv));   // the tile's first table index (its first output sits on phase 0)
    {
        const unsigned long long left0 = o_end > t_in * TT ? o_end - t_in * TT : 0ull;
        if (left0) fetch(kb, tile_nst((int)(left0 < (unsigned long long)TT ? left0 : (unsigned long long)TT)), pva, vcur);
    }
    landed(pva);
    // results on their way out: ob[parity][wave][output of the sub-tile], written where the recurrence leaves them (lane stride 5), read a row of 64
    // consecutive outputs per store instruction — a sub-tile later, behind that one's tap reads (the stores then have a sub-tile's arithmetic before
    // the tile's end waits vmcnt(0)); the tile's last sub-tile leaves at the top of the next tile
    int pend_c0 = 0, pend_c1 = 0;
    unsigned long long pend_o0 = 0, pend_o1 = 0;
    auto drain = [&](const int p) {   // p: the buffer (wave-uniform)
        if constexpr (NW > 1) __syncthreads();   // (both channels' results of that sub-tile are in `ob`; the same turn of both waves)
        else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        const float *const b = ob + p * (NW * T);
        const int pc = p ? pend_c1 : pend_c0;
        float *const od = orow + (p ? pend_o1 : pend_o0) + (unsigned)lane;
        auto rows = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
            if constexpr (NW > 1) {   // the rows of 64 shared out: wave 0 takes rows 0, 2, 4, wave 1 rows 1, 3
                constexpr int KR = (E + 1) / 2;
                float val[KR];
#pragma unroll
                for (int k = 0; k < KR; k++) {
                    const int u = (int)wv + 2 * k;
                    val[k] = 0.f;
                    if (u < E) {
                        float acc = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; w++) acc = acc + b[w * T + lane + 64 * u];   // s = 0 + c1 + c2  :685
                        val[k] = acc / (float)NW;                                            // s / cn  :686
                    }
                }
#pragma unroll
                for (int k = 0; k < KR; k++) {
                    const int u = (int)wv + 2 * k;
                    if (u < E && (FULL || lane + 64 * u < pc)) RSP_STORE(od + 64 * u, val[k]);
                }
            } else {
                float val[E];
#pragma unroll
                for (int u = 0; u < E; u++) val[u] = b[lane + 64 * u];
#pragma unroll
                for (int u = 0; u < E; u++) if (FULL || lane + 64 * u < pc) RSP_STORE(od + 64 * u, val[u]);
            }
        };
        if (pc == T) rows(std::true_type{}); else rows(std::false_type{});   // (all reads, then all stores: a full sub-tile has no predicates in between)
        if (p) pend_c1 = 0; else pend_c0 = 0;
    };
    const bool sym = P.scale == P.scale_neg;   // (FLAC's int16 finals: no sign select)
    float *const ow = ob + (NW > 1 ? (int)wv * T : 0) + e0;   // where this lane's five results go (+ the buffer)
    for (unsigned long long o0 = t_in * TT; o0 < o_end; o0 += TT) {
        const bool emit = o0 >= o_lo;   // (a warm-up tile: state only)
        const int cnt = (int)((o_end - o0) < (unsigned long long)TT ? (o_end - o0) : (unsigned long long)TT);
        const int nst = tile_nst(cnt);
        __builtin_amdgcn_wave_barrier();
        float *const wb = win + vcur.hA;   // the window's first element (table index kb)
        // (v < 0 and 1 / 32768 or 1 / 32767: the larger of the two products is the right one either side of 0 — two multiplies and a max, no compare)
        // (v < 0 and 1 / 32768 or 1 / 32767: the larger of the two products is the right one either side of 0 — two packed multiplies for a pair and a max each, no compare)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto cvt_a = [&](int va, int vb, float &a, float &b) {
            const f32x2 f = {(float)va, (float)vb};
            const f32x2 n = f * (f32x2){P.scale_neg, P.scale_neg}, q = f * (f32x2){P.scale, P.scale};
            asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(n.x), "v"(q.x));   // (products of finite numbers: nothing to canonicalise first, which fmaxf would spend an instruction on)
            asm("v_max_f32 %0, %1, %2" : "=v"(b) : "v"(n.y), "v"(q.y));
        };
        auto cvt_s = [&](int va, int vb, float &a, float &b) { const f32x2 f = (f32x2){(float)va, (float)vb} * (f32x2){P.scale, P.scale}; a = f.x; b = f.y; };
        auto stage = [&](auto cvt) {
            if (vcur.on) {   // (wave-uniform)
                float4 f0, f1, f2, f3;
                if constexpr (sizeof(S) == 2) {
                    auto s16 = [&](unsigned w, float &a, float &b) { cvt((int)(short)(w & 0xFFFFu), (int)w >> 16, a, b); };
                    s16(pva.lo.x, f0.x, f0.y); s16(pva.lo.y, f0.z, f0.w); s16(pva.lo.z, f1.x, f1.y); s16(pva.lo.w, f1.z, f1.w);
                    s16(pva.hi.x, f2.x, f2.y); s16(pva.hi.y, f2.z, f2.w); s16(pva.hi.z, f3.x, f3.y); s16(pva.hi.w, f3.z, f3.w);
                } else {
                    auto s8 = [&](unsigned w, float4 &f) {
                        cvt((int)(signed char)(w & 0xFFu), (int)(signed char)((w >> 8) & 0xFFu), f.x, f.y);
                        cvt((int)(signed char)((w >> 16) & 0xFFu), (int)w >> 24, f.z, f.w);
                    };
                    s8(pva.lo.x, f0); s8(pva.lo.y, f1); s8(pva.lo.z, f2); s8(pva.lo.w, f3);
                }
                if (lane < vcur.nvA) {
                    float4 *const d = reinterpret_cast<float4 *>(win + EPV * lane);
                    d[0] = f0; d[1] = f1; d[2] = f2; d[3] = f3;
                }
                if (vcur.nvB) {   // (a window across two frames)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int sb = lane - vcur.nvA;
                    if (sb >= 0 && sb < vcur.nvB) {
                        const int di = vcur.dposB + EPV * sb;
                        const float fl[16] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w, f3.x, f3.y, f3.z, f3.w};
#pragma unroll
                        for (int k = 0; k < 16; k++) if (di + k >= vcur.loB) wb[di + k] = fl[k];
                    }
                }
            } else {
                for (int j = lane; j < nst; j += 64) {   // (base0 / base1 / bound still describe THIS tile: the fetch below moves them on)
                    const unsigned k = kb + (unsigned)j;
                    const int kc = k < 1u ? 1 : (k > (unsigned)L ? L : (int)k);   // the nil fall-backs of interpolate.cubic (:264): the edge samples repeated
                    const S *src = P.frames ? rows_s + ((kc - 1 < bound ? base0 : base1) + (long long)(kc - 1)) : row + (kc - 1);
                    float fa, fb2;
                    cvt(L > 0 ? ((JOBS && k == 0u) ? hist : (int)*src) : 0, 0, fa, fb2);   // (JOBS: table index 0 is the history sample, not the edge)
                    win[j] = fa;
                }
            }
        };
        auto cvt_u = [&](int va, int vb, float &a, float &b) { a = (float)va; b = (float)vb; };   // (stream.qoa's int8 rows: the samples as they are)
        if (P.scale == 1.0f && P.scale_neg == 1.0f) stage(cvt_u); else if (sym) stage(cvt_s); else stage(cvt_a);
        if (pend_c1) drain(1);   // (wave- and workgroup-uniform) the last sub-tile of the tile before
        const unsigned kb_n = kb + (unsigned)NSUB * half_adv;
        if (o0 + TT < o_end) {
            const unsigned long long left = o_end - o0 - TT;
            fetch(kb_n, tile_nst((int)(left < (unsigned long long)TT ? left : (unsigned long long)TT)), pva, vnxt);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto sub = [&](const int h, const int cn, auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;   // all 320 outputs exist: no predicates
            const int par = h & 1;
            const float *const tp = wb + (unsigned)h * half_adv + q0;
            float t[NT];
#pragma unroll
            for (int k = 0; k < NT; k++) t[k] = tp[k];
            if (h > 0 && (par ? pend_c0 : pend_c1)) drain(par ^ 1);   // the sub-tile before leaves behind this one's tap reads
            float v[E];
#pragma unroll
            for (int i = 0; i < E; i++) {
                const int d = rsp_dm<PAT>(i);
                float acc = W[i][0] * t[d];
                acc = fmaf(W[i][1], t[d + 1], acc);
                acc = fmaf(W[i][2], t[d + 2], acc);
                acc = fmaf(W[i][3], t[d + 3], acc);
                if (i > 0) acc = fmaf(W[i][4], t[d + 4], acc);   // (the lane's first output starts its taps: never shifted)
                v[i] = JOBS ? __builtin_amdgcn_fmed3f(acc, P.clo, P.chi) : __builtin_amdgcn_fmed3f(acc, -1.0f, 1.0f);   // :667-668; JOBS: :3323
            }
            // x[n - 1] of the lane's first output: the lane before's last, the sub-tile before's for lane 0
            float xprev = 0.f;
            if constexpr (HP) {
                xprev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[E - 1]), 0x138, 0xF, 0xF, true));
                if (lane == 0) xprev = carry_x;
            }
            RT z[E];
            {
                RT xp = (RT)xprev;
                const bool first = !JOBS && o0 == 0 && h == 0 && lane == 0;
                RT y = 0;
#pragma unroll
                for (int i = 0; i < E; i++) {
                    const RT xv = (FULL || e0 + i < cn) ? (RT)v[i] : xp;
                    if (i == 0) {
                        RT y1;
                        if constexpr (HP) y1 = coef_r * (xv - xp); else y1 = coef_r * xv;   // (from y = 0: :3614, :3594)
                        y = first ? xv : y1;                                                // y[1] = x[1]: the first sample passes  (:3592, :3612)
                    }
                    else if constexpr (HP) y = coef_r * (y + xv - xp);         // :3614
                    else y = fmaR(coef_r, xv - y, y);                          // :3594 (fused: the tolerance path)
                    xp = xv;
                    z[i] = y;
                }
            }
            RT Y = z[E - 1];
            Y = fmaR(Md[0], dpp_rt<0x111>(Y), Y);
            Y = fmaR(Md[1], dpp_rt<0x112>(Y), Y);
            Y = fmaR(Md[2], dpp_rt<0x114>(Y), Y);
            Y = fmaR(Md[3], dpp_rt<0x118>(Y), Y);
            Y = fmaR(mA, dpp_rt<0x142, 0xA>(Y), Y);
            Y = fmaR(mB, dpp_rt<0x143, 0xC>(Y), Y);
            Y = fmaR(mlane, carry_y, Y);
            RT yin = dpp_rt<0x138>(Y);
            if (lane == 0) yin = carry_y;
            float res[E];
            RT ylast = 0;
            float xlast = 0.f;
#pragma unroll
            for (int i = 0; i < E; i++) {
                const RT yv = fmaR(mp[i + 1], yin, z[i]);
                res[i] = (float)yv;
                if constexpr (!FULL) {
                    if (e0 + i < cn) mxf = fmaxf(mxf, fabsf(res[i]));
                    if (e0 + i == cn - 1) { ylast = yv; xlast = v[i]; }
                }
            }
            if constexpr (FULL) {
                // (a warm-up tile's maxima are dropped at its end; results of FMAs: nothing to canonicalise)
                asm("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|\n\tv_max_f32 %0, %0, |%5|" : "+v"(mxf) : "v"(res[0]), "v"(res[1]), "v"(res[2]), "v"(res[3]), "v"(res[4]));
                if constexpr (R32) carry_y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Y), 63));
                else carry_y = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(Y) >> 32), 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)__double_as_longlong(Y), 63));
                if constexpr (HP) carry_x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[E - 1]), 63));
            } else {
                carry_y = __shfl(ylast, (cn - 1) / E);
                if constexpr (HP) carry_x = __shfl(xlast, (cn - 1) / E);
            }
            if (emit) {
                float *const w = ow + par * (NW * T);
#pragma unroll
                for (int i = 0; i < E; i++) w[i] = res[i];
                if (par) { pend_c1 = cn; pend_o1 = o0 + (unsigned)(h * T); } else { pend_c0 = cn; pend_o0 = o0 + (unsigned)(h * T); }
            }
        };
        if (cnt == TT) {
            if constexpr (NSUB == 2) { sub(0, T, std::true_type{}); sub(1, T, std::true_type{}); }   // (two copies: the buffers' addresses and the drains' conditions fold)
            else {
#pragma nounroll
                for (int h = 0; h < NSUB; h++) sub(h, T, std::true_type{});
            }
        } else {
#pragma nounroll
            for (int h = 0; h * T < cnt; h++) sub(h, cnt - h * T < T ? cnt - h * T : T, std::false_type{});   // (a row's last tile)
        }
        if (!emit) mxf = 0.f;
        landed(pva);   // (requested before this tile's arithmetic)
        kb = kb_n; vcur = vnxt;
    }
    if (pend_c0) drain(0);
    if (pend_c1) drain(1);
    if constexpr (!JOBS) {
        for (int o = 32; o; o >>= 1) mxf = fmaxf(mxf, __shfl_xor(mxf, o));
        if (lane == 0) atomicMax(&P.rowmax[r], (unsigned long long)__double_as_longlong((double)mxf));
        if constexpr (NW > 1) { if (lane == 0) atomicMax(&P.rowmax2[s], (unsigned long long)__double_as_longlong((double)mxf)); }
    }
}

// the host's side: which ratios, how much LDS, how many runs per row
bool rsp_try(aukit_ctx *ctx, RsOnepoleParams &P, bool highpass, bool r32, int NW, size_t rows, uint64_t min_out_len, int min_frame, int *rc) {
    *rc = AUKIT_OK;
    constexpr int E = 5, T = 64 * E;
    if (getenv("AUKIT_RS_GENERIC") || !P.wg || P.fa >= P.fb || ((unsigned long long)T * P.fa) % P.fb) return false;
    int pat = 0;
    {
        int dm[E];
        for (int i = 0; i < E; i++) dm[i] = (int)(((unsigned long long)i * P.fa) / P.fb);
        if (dm[1] == 0 && dm[2] == 1 && dm[3] == 2 && dm[4] == 3) pat = 1;
        else if (dm[1] == 0 && dm[2] == 0 && dm[3] == 1 && dm[4] == 1) pat = 2;
    }
    if (!pat) return false;
    const int TT = (pat == 1 ? 2 : 4) * T;   // (k_rsp's NSUB)
    const unsigned adv = (unsigned)(((unsigned long long)TT * P.fa) / P.fb);
    if (P.frames && min_frame > 0 && (unsigned)min_frame < adv + 24u) return false;   // a tile's window lies in one frame or two consecutive ones
    if (((double)P.fb + (double)TT * (double)P.fa) * (double)P.fb >= 4294967296.0) return false;   // (tile_nst's magic division)
    RsOnepoleParams Q = P;
    Q.cap = (((int)adv + 24 + 3) & ~3) + 32;   // 16 in front, the window from up to 15 floats in, whole vectors of 16 to its end and one float more
    Q.wave_lds = Q.cap;
    const size_t lds = ((size_t)NW * Q.wave_lds + 2 * (size_t)NW * T) * 4;
    {   // runs of tiles per row: lazy_onepole_try's rule on this kernel's tile
        const double m = highpass ? P.coef : 1.0 - P.coef;
        int warm = 0, segs = 1;
        if (m > 0 && m < 1 && !getenv("AUKIT_RS_ONE_CHAIN")) {
            warm = (int)std::ceil(40.0 * M_LN2 / -std::log(m) / TT);
            const uint64_t min_tiles = (min_out_len + TT - 1) / TT;
            const uint64_t by_rows = std::max<uint64_t>(1, 4096 / std::max<size_t>(rows, 1)), by_len = min_tiles / (8ull * (uint64_t)std::max(warm, 1));
            segs = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min(by_rows, by_len), 16));
        }
        if (getenv("AUKIT_RS_SEGS")) segs = std::max(1, atoi(getenv("AUKIT_RS_SEGS")));
        Q.segs = segs; Q.warm = segs > 1 ? warm : 0;
    }
    const dim3 grid((unsigned)((rows / (size_t)NW) * (size_t)Q.segs));
#define AUKIT_RSP(H, R, N)                                                                                                      \
    do {                                                                                                                          \
        if (pat == 1) hipLaunchKernelGGL((k_rsp<H, R, N, 1>), grid, dim3(64 * N), lds, ctx->stream, Q);                           \
        else hipLaunchKernelGGL((k_rsp<H, R, N, 2>), grid, dim3(64 * N), lds, ctx->stream, Q);                                    \
    } while (0)
    if (NW == 2) { if (highpass) AUKIT_RSP(true, false, 2); else if (r32) AUKIT_RSP(false, true, 2); else AUKIT_RSP(false, false, 2); }
    else { if (highpass) AUKIT_RSP(true, false, 1); else if (r32) AUKIT_RSP(false, true, 1); else AUKIT_RSP(false, false, 1); }
#undef AUKIT_RSP
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_rsp launch failed"); return true; }
    return true;
}

// stream.qoa's tail on k_rsp: long jobs of int8 rows, a low-pass (rs_onepole_jobs_launch, flac_tail.hip, fills P and the grid).  false: not this shape
bool rsp_jobs_try(aukit_ctx *ctx, const RsOnepoleParams &P, bool r32, int NW, unsigned grid, int *rc) {
    *rc = AUKIT_OK;
    constexpr int E = 5, T = 64 * E;
    if (getenv("AUKIT_RS_GENERIC") || !P.wg || P.epi || P.fa >= P.fb || ((unsigned long long)T * P.fa) % P.fb) return false;
    int dm[E];
    for (int i = 0; i < E; i++) dm[i] = (int)(((unsigned long long)i * P.fa) / P.fb);
    if (!(dm[1] == 0 && dm[2] == 1 && dm[3] == 2 && dm[4] == 3)) return false;   // (44.1 kHz: the one pattern built for jobs)
    const int TT = 2 * T;
    const unsigned adv = (unsigned)(((unsigned long long)TT * P.fa) / P.fb);
    if (((double)P.fb + (double)TT * (double)P.fa) * (double)P.fb >= 4294967296.0) return false;
    RsOnepoleParams Q = P;
    Q.cap = (((int)adv + 24 + 3) & ~3) + 32;
    Q.wave_lds = Q.cap;
    Q.segs = 1; Q.warm = 0; Q.frames = nullptr;
    const size_t lds = ((size_t)NW * Q.wave_lds + 2 * (size_t)NW * T) * 4;
    if (NW == 2) { if (r32) hipLaunchKernelGGL((k_rsp<false, true, 2, 1, signed char, true>), dim3(grid), dim3(128), lds, ctx->stream, Q); else hipLaunchKernelGGL((k_rsp<false, false, 2, 1, signed char, true>), dim3(grid), dim3(128), lds, ctx->stream, Q); }
    else { if (r32) hipLaunchKernelGGL((k_rsp<false, true, 1, 1, signed char, true>), dim3(grid), dim3(64), lds, ctx->stream, Q); else hipLaunchKernelGGL((k_rsp<false, false, 1, 1, signed char, true>), dim3(grid), dim3(64), lds, ctx->stream, Q); }
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_rsp<jobs> launch failed"); return true; }
    return true;
}

}  // namespace aukit
