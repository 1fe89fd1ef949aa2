// flac_tail.hip — `aukit.flac(d):resample(48000, interp)` followed by `aukit.effects.highpass / lowpass` as ONE pass over the decoded rows
// (BASELINE config 5's tail).
//
// With F32 storage the resampled rows used to be written by k_fast_wave<i32> (7.9 GB for config 5) only to be read back and rewritten by
// k_onepole (7.9 + 8.1 GB).  Now `aukit_decode_resample` on FLAC leaves the audio LAZY: its int32 rows stay where the decoder put them (the
// buffer moves from the context's scratch into the audio, so no later call can overwrite it) and the resample is owed.  If the next call is
// effects.highpass / lowpass, k_rs_onepole below pays it inside the filter pass: one workgroup walks one output row tile by tile —
//   stage the tile's window of int32 samples as f32 (the `/ 2^depth` of :505 is an exact scaling) → interpolate exactly as
//   k_fast_wave<i32> does (exact rational positions; cubic through per-phase weights where the rate's denominator allows: within an f32 ulp or two of that kernel's Horner form, so `resample` then `highpass` may differ in the last bit from the unfused calls — tests: 4e-7) → run the one-pole recurrence over the tile in fp64,
//   thread ↔ 8 consecutive outputs with an affine carry scan across the workgroup and the carry of the tile before (the high-pass at 20 Hz
//   remembers thousands of samples: no truncation here, cf. stream_tail.hip) → store f32, record the row's |max| for effects.normalize.
// Anything else that reads the samples first (download, another effect, Audio:mono ...) materialises the resample with the ordinary kernel
// (audio_flush → lazy_materialize).  AUKIT_NO_TAIL_FUSION=1 switches the laziness off (A/B, tests).
#include <algorithm>
#include <type_traits>
#include "resample.h"
#include "flac_dev.h"
#include "stream_tail.h"
#include "rs_onepole_dev.h"

namespace aukit {

int audio_from_int_rows(aukit_ctx *ctx, int src_kind, const void *rows_dev, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len,
                        uint32_t n, int channels, double rate, double new_rate, int interp, bool do_resample, int dtype, double norm_pos,
                        double norm_neg, aukit_audio **out);
bool fast_eligible(int src_kind, int interp, double old_rate, double new_rate, FastParams &F);

// One WAVE per output row (4096 rows of config 5 = four waves per SIMD, all resident at once: no tail, no block barrier anywhere), tiles of 512
// outputs, the tile's carry in registers.  (The first version — a 256-thread workgroup per row, tiles of 2048, five block barriers per tile —
// ran at 7.3 ms on config 5 against 6.7 ms for the two kernels it replaces: a row is a serial chain of tiles, and what a tile costs is its
// latency, not its work.)
// TAB: the cubic as w0 p0 + w1 p1 + w2 p2 + w3 p3 with the weights of the output's phase from an LDS table (fb <= 512 phases; one multiply and
// three FMAs per output instead of the twelve operations of the coefficient + Horner form: the kernel is bound by its instruction count)
#ifndef AUKIT_RS_WAVES
#define AUKIT_RS_WAVES 4
#endif
// R32 (round 4, last): the recurrence and its scan in f32 — for slopes m <= 1/2 only (a low-pass well above the band: stream.qoa's and stream.flac's,
// effects.lowpass at a quarter of the rate), where a step's rounding (2^-24 of the state) is worth at most twice itself in the end: ~1e-7 of the
// scale, inside the f32 stages' 1e-6.  Saves the two conversions per output and half of the scan's moves in a kernel bound by its instruction count.
template <int INTERP, bool HP, bool TAB, typename S, int NW = 1, bool JOBS = false, bool LOOPJ = false, bool R32 = false>
__global__ __launch_bounds__(64 * NW, AUKIT_RS_WAVES) void k_rs_onepole(const RsOnepoleParams P) {
    using RT = std::conditional_t<R32, float, double>;   // (four waves per SIMD asked for: 128 VGPRs — hipcc takes 130 - 133 for some instantiations otherwise, three waves per SIMD, and the launches are sized for four: config 3b ran in 1.33 rounds, 2.8 -> 3.8 ms)
    extern __shared__ __attribute__((aligned(16))) float rsm_all[];
    constexpr int E = 8, T = 64 * E;
    const unsigned wv = NW > 1 ? (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;   // the wave = the channel
    float *const rsm = rsm_all + (NW > 1 ? wv * (unsigned)P.wave_lds : 0u);
    float *const win = rsm + 16;                             // P.cap floats in all: 16 of slack in front, the window, 16 behind (whole vectors spill up to EPV - 1 elements either way, and the window starts hA <= 7 floats in)
    float *const xb = rsm + P.cap;                           // T + T / E + 8
    // a lane's vector = EIGHT elements (8, 16 or 32 bytes of the row, aligned to that), one per lane and tile: their floats leave as two aligned
    // ds_write_b128 at a lane stride of 32 bytes.  (Until late round 4 a vector was 16 BYTES: 8 or 16 elements written float by float to wherever the
    // window's first element put them, at a lane stride of 8 or 16 dwords — an 8- or 16-way bank conflict on every one of those writes.  PMC on
    // stream.qoa's tail: SQ_LDS_BANK_CONFLICT 57 % of SQ_LDS_IDX_ACTIVE, waves waiting on the LDS 27 % of their time.)
    constexpr int EPV = 8, VB = EPV * (int)sizeof(S), DPV = VB / 4, VPL = 1;
    struct V32 { u32x4g lo, hi; };
    using VT = std::conditional_t<DPV == 8, V32, std::conditional_t<DPV == 4, u32x4g, u32x2g>>;
    [[maybe_unused]] float *const wt = NW > 1 ? rsm_all + NW * P.wave_lds : xb + (T + T / E + 8);   // TAB: 4 fb floats (one table per workgroup)
    constexpr int MIXN = T + T / E + 8;                                      // NW > 1: a wave's tile of results, skewed like xb
    [[maybe_unused]] float *const mix = rsm_all + NW * P.wave_lds + (TAB ? (4 * (int)P.fb + 3 & ~3) : 0);   // [wave][MIXN]
    if constexpr (TAB) {
        for (unsigned i = threadIdx.x; i < 4 * P.fb; i += 64 * NW) wt[i] = P.wg[i];
        if constexpr (NW > 1) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }
    const int lane = (int)(threadIdx.x & 63u);
    const S *const rows_s = reinterpret_cast<const S *>(P.rows);
    // the slope of the recurrence's affine map y -> m y + ...: a (high-pass, :3614), 1 - alpha (low-pass, :3594)
    const double m = HP ? P.coef : 1.0 - P.coef;
    double mpd[E + 1];   // m^1 .. m^E (mpd[0] = 1)
    mpd[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= E; i++) mpd[i] = mpd[i - 1] * m;
    double Mdd[6];       // M^1, M^2, M^4 ... M^32 with M = m^E: the steps of the wave scan
    Mdd[0] = mpd[E];
#pragma unroll
    for (int k = 1; k < 6; k++) Mdd[k] = Mdd[k - 1] * Mdd[k - 1];
    // (the same in every lane, but made by vector multiplies: said so, these 14 doubles live in scalar registers — 28 VGPRs of a kernel that is held to
    // 128 and spilled into its tile loop without them; an FMA takes one of them as its scalar operand)
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
    // the scan runs on DPP moves: four steps inside rows of 16 lanes (a lane without a source receives 0), then lane 15 / lane 31 of the row(s)
    // before — with what the receiving lane's distance makes of them
    const RT mA = (RT)((lane & 16) ? pow(mpd[E], (double)((lane & 15) + 1)) : 0.0);
    const RT mB = (RT)(lane >= 32 ? pow(mpd[E], (double)(lane - 31)) : 0.0);
    const RT mlane = (RT)pow(mpd[E], (double)(lane + 1));   // M^(lane + 1): what the tile's carry is worth after this lane's outputs
    auto skew = [](int i) { return i + i / E; };
    const unsigned wc = (unsigned)(((unsigned long long)T * P.fa) / P.fb), wd = (unsigned)(((unsigned long long)T * P.fa) % P.fb);
    // JOBS: a workgroup takes job after job (short jobs — a FLAC frame is nine tiles — would otherwise pay the set-up above once each)
    auto run_item = [&](const unsigned item) {
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
    float *orow = P.out + obase;
    float mxf = 0.f;
    RT carry_y = 0, carry_x = 0;
    if constexpr (JOBS) {   // ls = last[2] (stream.qoa :3316), or last[2] / (last[2] < 0 and 128 or 127) (stream.flac :3172)
        const double z0 = (double)((float)hist * (hist < 0 ? P.scale_neg : P.scale));
        carry_y = (RT)(P.epi ? z0 / (z0 < 0 ? 128.0 : 127.0) : z0);
    }
    // x - 1 = o fa / fb exactly.  The tile's first output: (kb, r0) advanced by additions from tile to tile (T fa = wc fb + wd); the outputs inside it
    // from there (n < fb + T fa: the magic division is exact, as in the wave kernels)
    unsigned kb = 0, r0 = 0;   // (row lengths stay below 2^31: checked by the host)
    // the window of a tile: table indices kb .. kb + nst - 1, clamped into 1 .. #data — the nil fall-backs of interpolate.{linear,cubic} (:259, :264)
    // are the edge samples repeated.  Its first 512 entries travel through registers, loaded one tile AHEAD (a row is a serial chain of tiles:
    // a load waited for where it is issued costs its whole latency, eight times per tile in the first cut of this loop)
    auto tile_nst = [&](unsigned rr, int cn) { const unsigned nn = rr + ((unsigned)cn - 1) * P.fa; return (int)__umulhi(nn, P.fmagic) + 4; };
    // frame-by-frame rows: sample i of the row lies at base0 + i (the frame that holds the window's first sample) or base1 + i (the frame after
    // it).  The records are walked as the tiles advance, the one after next already loaded (base2): no look-up sits in front of a tile's loads
    const int bsn = P.frames ? P.bs0[s] : 0;
    const unsigned nfr_s = P.frames ? (unsigned)(P.fbase[s + 1] - P.fbase[s]) : 0u;
    const FrameRec *const fr_s = P.frames ? P.frames + P.fbase[s] : nullptr;
    long long base0 = 0, base1 = 0, base2 = 0;
    int bound = 0x7FFFFFFF;
    unsigned fcur = 0;
    // (the record by SCALAR loads — the frame index is the wave's: as vector loads they waited vmcnt(0), behind the tile's stores)
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
    // Round 4, late: the window as ALIGNED VECTORS of eight elements where it lies inside the row (every tile but a row's first and last).  One
    // element per lane and load — a clamp, a frame select and a 64-bit address each — was 13 of the kernel's 45 VALU instructions per output.  A
    // window is one run of elements (part A) or two (frame-by-frame rows: the rest in the frame after, part B); a part is read from the vector that
    // holds its first element on, vector s of the tile by lane s (at most 64 of them: else the element-by-element way).  Part A's vectors are written
    // to LDS as they stand, where their alignment puts them (`win + 8 s`: the window's first element is `win[hA]`) — what they hold beyond the part
    // falls into the slack around the window or onto part B's place — part B's behind them, element by element, its first vector guarded.
    struct VecDesc { int on, nvA, nvB, hA, dposB, loB; };   // hA: the window's first element sits hA floats behind `win` (part A's vectors are written where their alignment puts them)
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
        const int e0 = (int)kk - 1, e1 = e0 + nst;   // row elements [e0, e1)
        uintptr_t va = 0;   // this lane's vector
        if (L > 0 && e0 >= EPV && e1 + EPV <= L && !P.novec) {
            const int xeA = P.frames ? (e1 < bound ? e1 : bound) : e1;
            const S *const pA = P.frames ? rows_s + (base0 + (long long)e0) : row + e0;
            const uintptr_t aA = (uintptr_t)pA, alA = aA & ~(uintptr_t)(VB - 1);
            const int hA = (int)((aA - alA) / sizeof(S));
            const int nvA = (hA + (xeA - e0) + EPV - 1) / EPV;
            uintptr_t alB = alA;
            int nvB = 0, hB = 0;
            if (e1 > xeA) {
                const uintptr_t aB = (uintptr_t)(rows_s + (base1 + (long long)xeA));
                alB = aB & ~(uintptr_t)(VB - 1);
                hB = (int)((aB - alB) / sizeof(S));
                nvB = (hB + (e1 - xeA) + EPV - 1) / EPV;
            }
            if (nvA + nvB <= 64 * VPL) {
                vd.on = 1; vd.nvA = nvA; vd.nvB = nvB; vd.hA = hA; vd.dposB = (xeA - e0) - hB; vd.loB = xeA - e0;
                const int sb = lane - nvA;
                const bool inA = lane < nvA, inB = !inA && sb < nvB;
                va = inB ? alB + (unsigned)VB * (unsigned)sb : alA + (inA ? (unsigned)VB * (unsigned)lane : 0u);   // (a lane without a vector reads the tile's first one)
            }
        }
        // (a window that cannot come as vectors — a row's first and last tiles — is loaded element by element where it is staged, at the top of its own
        // tile: C++ loads in here put a `vmcnt(0)` at this function's join, on every tile's path, behind the stores of the tile before)
        // A load hipcc does not see, in NO arm of a fork.  As a C++ load it sat in the vector arm above, its value met the other arm's at the join, and
        // hipcc waited `vmcnt(0)` for it right there — six instructions behind the load: the "tile ahead" was never ahead, every tile stood a whole HBM
        // latency (PMC: waves at an s_waitcnt 60 % of their time).  Now the request is inline asm into the register(s) the value keeps (a read-write
        // operand: no copy anywhere), issued for every tile with the lanes masked off when the window goes element by element, the tile's END waits
        // vmcnt(0) by hand (`landed` below), and tools/isa_check.py --cfg walks the built code's control-flow graph to prove that nothing touches those
        // registers in between (tests/test_isa_schedule.py).
        {
            unsigned long long save;
            const unsigned on = (unsigned)vd.on;
            if constexpr (DPV == 8)
                asm volatile("v_cmp_ne_u32_e32 vcc, 0, %3\n\ts_and_saveexec_b64 %2, vcc\n\tglobal_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\ts_mov_b64 exec, %2"
                             : "+v"(pva.lo), "+v"(pva.hi), "=&s"(save) : "v"(on), "v"(va) : "vcc", "memory");
            else if constexpr (DPV == 4)
                asm volatile("v_cmp_ne_u32_e32 vcc, 0, %2\n\ts_and_saveexec_b64 %1, vcc\n\tglobal_load_dwordx4 %0, %3, off\n\ts_mov_b64 exec, %1" : "+v"(pva), "=&s"(save) : "v"(on), "v"(va) : "vcc", "memory");
            else
                asm volatile("v_cmp_ne_u32_e32 vcc, 0, %2\n\ts_and_saveexec_b64 %1, vcc\n\tglobal_load_dwordx2 %0, %3, off\n\ts_mov_b64 exec, %1" : "+v"(pva), "=&s"(save) : "v"(on), "v"(va) : "vcc", "memory");
        }
    };
    // (the vector is a VALUE, not elements of an array: with an int pre[8] serving both ways of fetch(), hipcc merged the last store of either path
    // into ONE store at a variable index — and an array indexed by a variable lives in scratch memory: loads and stores behind vmcnt(0) in the middle
    // of the tile loop)
    VT pva{};   // this lane's vector of the window
    auto landed = [](VT &v) {   // the hand-issued request (fetch) has arrived — and with it every store of the tile
        if constexpr (DPV == 8) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v.lo), "+v"(v.hi) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");
    };
    VecDesc vcur{0, 0, 0, 0, 0, 0}, vnxt{0, 0, 0, 0, 0, 0};
    // this wave's run of tiles [t_lo, t_hi) of the row's ntiles, entered `warm` tiles early
    const unsigned long long ntiles = (nout + T - 1) / T;
    const unsigned long long t_lo = ntiles * seg / (unsigned)P.segs, t_hi = ntiles * (seg + 1u) / (unsigned)P.segs;
    const unsigned long long t_in = t_lo > (unsigned long long)P.warm ? t_lo - (unsigned long long)P.warm : 0ull;
    const unsigned long long o_lo = t_lo * T, o_end = t_hi * T < nout ? t_hi * T : nout;
    {
        const unsigned long long nn = t_in * (unsigned long long)T * P.fa;   // x - 1 of the run's first output = nn / fb exactly
        kb = (unsigned)(nn / P.fb); r0 = (unsigned)(nn % P.fb);
        const unsigned long long left0 = o_end > t_in * T ? o_end - t_in * T : 0ull;
        const int cnt0 = (int)(left0 < (unsigned long long)T ? left0 : (unsigned long long)T);
        if (left0) fetch(kb, tile_nst(r0, cnt0), pva, vcur);
    }
    landed(pva);
    // a full tile's results wait in registers (held[]) and are stored at the top of the NEXT turn, behind the wait for that tile's window: loads and
    // stores share vmcnt and hipcc waits vmcnt(0) across this loop's branches — stores issued at the end of a turn were waited for at the top of
    // the next one, a store latency per tile in a chain of 938 tiles (the late-store schedule of wave_f64.hip)
    float held[E];
    float *held_at = nullptr;
#pragma unroll
    for (int u = 0; u < E; u++) held[u] = 0.f;
    // NW > 1: a tile's results wait in LDS (mix[parity][wave]) for the other channel's; behind the workgroup barrier at the top of the next turn
    // every wave reads both, averages and stores ITS share of the tile's outputs (T / NW consecutive ones)
    [[maybe_unused]] int mix_cnt = 0, mix_par = 0, mix_par_next = 0;
    [[maybe_unused]] unsigned long long mix_o0 = 0;
    [[maybe_unused]] auto mix_out = [&]() {
        __syncthreads();   // (both channels' results of the tile before are in `mix`)
        const float *const mp0 = mix;
        constexpr int SH = T / NW;   // outputs per wave
#pragma unroll
        for (int u = 0; u < SH / 64; u++) {
            const int idx = (int)wv * SH + lane + 64 * u;
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) acc = acc + mp0[w * MIXN + skew(idx)];       // s = 0 + c1 + c2  :685
            if (idx < mix_cnt) orow[mix_o0 + (unsigned)idx] = acc / (float)NW;       // s / cn  :686
        }
        mix_cnt = 0;
    };
    for (unsigned long long o0 = t_in * T; o0 < o_end; o0 += T) {
        const bool emit = o0 >= o_lo;   // (a warm-up tile: state only)
        const int cnt = (int)((o_end - o0) < (unsigned long long)T ? (o_end - o0) : (unsigned long long)T);
        auto qr = [&](unsigned j, unsigned &q, unsigned &rem) { const unsigned nn = r0 + j * P.fa; q = __umulhi(nn, P.fmagic); rem = nn - q * P.fb; };   // q relative to kb
        const int nst = tile_nst(r0, cnt);   // table indices kb .. kb + ql + 3 (floor(x) = q + 1; taps q .. q + 3)
        __builtin_amdgcn_wave_barrier();
        // (FLAC rows — every int32 row, and int16 ones whose two scales agree — need no sign select: a wave-uniform fork, two copies of the loops below)
        const bool sym = sizeof(S) == 4 || P.scale == P.scale_neg;
        auto cvt_a = [&](int v) -> float { return (float)v * (v < 0 ? P.scale_neg : P.scale); };
        auto cvt_s = [&](int v) -> float { return (float)v * P.scale; };
        float *const wb = win + vcur.hA;   // the window's first element (table index kb)
        auto stage = [&](auto cvt) {
        if (vcur.on) {   // (wave-uniform)
            auto cv8 = [&](const VT &x, float4 &f0, float4 &f1) {
                auto s16 = [&](unsigned w, float &a, float &b) { a = cvt((int)(short)(w & 0xFFFFu)); b = cvt((int)w >> 16); };
                auto s8 = [&](unsigned w, float4 &f) { f = make_float4(cvt((int)(signed char)(w & 0xFFu)), cvt((int)(signed char)((w >> 8) & 0xFFu)), cvt((int)(signed char)((w >> 16) & 0xFFu)), cvt((int)w >> 24)); };
                if constexpr (DPV == 8) {
                    f0 = make_float4(cvt((int)x.lo.x), cvt((int)x.lo.y), cvt((int)x.lo.z), cvt((int)x.lo.w));
                    f1 = make_float4(cvt((int)x.hi.x), cvt((int)x.hi.y), cvt((int)x.hi.z), cvt((int)x.hi.w));
                } else if constexpr (DPV == 4) { s16(x.x, f0.x, f0.y); s16(x.y, f0.z, f0.w); s16(x.z, f1.x, f1.y); s16(x.w, f1.z, f1.w); }
                else { s8(x.x, f0); s8(x.y, f1); }
            };
            float4 f0, f1;
            cv8(pva, f0, f1);
            if (lane < vcur.nvA) {   // (aligned: two ds_write_b128)
                *reinterpret_cast<float4 *>(win + EPV * lane) = f0;
                *reinterpret_cast<float4 *>(win + EPV * lane + 4) = f1;
            }
            if (vcur.nvB) {   // (a window across two frames: one tile in eight or nine of config 5)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int sb = lane - vcur.nvA;
                if (sb >= 0 && sb < vcur.nvB) {
                    const int di = vcur.dposB + EPV * sb;
                    if (di >= vcur.loB) wb[di] = f0.x;
                    if (di + 1 >= vcur.loB) wb[di + 1] = f0.y;
                    if (di + 2 >= vcur.loB) wb[di + 2] = f0.z;
                    if (di + 3 >= vcur.loB) wb[di + 3] = f0.w;
                    if (di + 4 >= vcur.loB) wb[di + 4] = f1.x;
                    if (di + 5 >= vcur.loB) wb[di + 5] = f1.y;
                    if (di + 6 >= vcur.loB) wb[di + 6] = f1.z;
                    if (di + 7 >= vcur.loB) wb[di + 7] = f1.w;
                }
            }
        } else {
#pragma unroll
        for (int u = 0; u < 8; u++) {   // (cap >= 512; base0 / base1 / bound still describe THIS tile: the fetch below moves them on)
            const int j = lane + 64 * u;
            const unsigned k = kb + (unsigned)j;
            const int kc = k < 1u ? 1 : (k > (unsigned)L ? L : (int)k);
            const S *src = P.frames ? rows_s + ((kc - 1 < bound ? base0 : base1) + (long long)(kc - 1)) : row + (kc - 1);
            win[j] = cvt((j < nst && L > 0) ? ((JOBS && k == 0u) ? hist : (int)*src) : 0);   // (JOBS: table index 0 is the history sample, not the edge)
        }
        }
        };
        if (sym) stage(cvt_s); else stage(cvt_a);
        [[maybe_unused]] auto cvt = cvt_a;   // (the element-by-element rest of wide windows)
        if (held_at) {   // (wave-uniform) the tile before this one
#pragma unroll
            for (int u = 0; u < E; u++) held_at[64 * u] = held[u];
            held_at = nullptr;
        }
        if constexpr (NW > 1) { if (mix_cnt) mix_out(); }   // (the same turn of both waves: the condition is the stream's, not the channel's)
        if (!vcur.on)
        for (int j = lane + 512; j < nst; j += 64) {   // ratios above one source sample per output (base0 / base1 / bound still describe THIS tile: the fetch below moves them on)
            const unsigned k = kb + (unsigned)j;
            const int kc = k < 1u ? 1 : (k > (unsigned)L ? L : (int)k);
            const S *src = P.frames ? rows_s + ((kc - 1 < bound ? base0 : base1) + (long long)(kc - 1)) : row + (kc - 1);
            const int sv = L > 0 ? (int)*src : 0;
            win[j] = (float)sv * (sv < 0 ? P.scale_neg : P.scale);
        }
        unsigned kb_n = kb + wc, r0_n = r0 + wd;
        if (r0_n >= P.fb) { r0_n -= P.fb; kb_n++; }
        if (o0 + T < o_end) {
            const unsigned long long left = o_end - o0 - T;
            fetch(kb_n, tile_nst(r0_n, (int)(left < (unsigned long long)T ? left : (unsigned long long)T)), pva, vnxt);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto compute = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;   // all 512 outputs exist: no predicates
        {
            unsigned q, rem;
            qr((unsigned)lane, q, rem);
            // (the eight values first, their LDS writes behind them: with a write to xb between two outputs' reads of the window and the weight table hipcc
            // kept the order — it cannot tell the arrays apart — and waited for each output's reads before it asked for the next one's: eight LDS round
            // trips in a row per tile)
            float vo[E];
#pragma unroll
            for (int u = 0; u < E; u++) vo[u] = 0.f;
#pragma unroll
            for (int u = 0; u < E; u++, q += P.dq256, rem += P.dr256) {   // (dq256 / dr256 hold the step of 64 outputs here)
                const int idx = lane + 64 * u;
                if (rem >= P.fb) { rem -= P.fb; q++; }
                if (FULL || idx < cnt) {
                    const float *tp = wb + q;   // tp[0] = p0 (index kb + q), tp[1] = p1 = data[floor(x)]
                    const float fx = (float)rem * P.inv_b;
                    float v;
                    if constexpr (INTERP == AUKIT_INTERP_LINEAR) v = fmaf(tp[2] - tp[1], fx, tp[1]);
                    else if constexpr (TAB) {
                        const float4 w = *reinterpret_cast<const float4 *>(wt + 4 * rem);
                        v = fmaf(w.w, tp[3], fmaf(w.z, tp[2], fmaf(w.y, tp[1], w.x * tp[0])));
                    } else {
                        const float p0 = tp[0], p1 = tp[1], p2 = tp[2], p3 = tp[3];
                        const float c3 = fmaf(1.5f, p1 - p2, 0.5f * (p3 - p0));
                        const float c2 = fmaf(-0.5f, p3, fmaf(2.0f, p2, fmaf(-2.5f, p1, p0)));
                        const float c1 = 0.5f * (p2 - p0);
                        v = fmaf(fmaf(fmaf(c3, fx, c2), fx, c1), fx, p1);
                    }
                    vo[u] = JOBS ? __builtin_amdgcn_fmed3f(v, P.clo, P.chi) : __builtin_amdgcn_fmed3f(v, -1.0f, 1.0f);   // :667-668 (rem == 0: v is p1 itself); JOBS: :3323
                }
            }
#pragma unroll
            for (int u = 0; u < E; u++)
                if (FULL || lane + 64 * u < cnt) xb[skew(lane + 64 * u)] = vo[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // lane ↔ E consecutive outputs: the recurrence from a zero state, then the state it really started from
        const int e0 = lane * E;
        RT z[E];
        {
            RT xp = (e0 == 0) ? carry_x : ((FULL || e0 <= cnt) ? (RT)xb[skew(e0 - 1)] : (RT)0);
            const bool first = !JOBS && o0 == 0 && lane == 0;
            RT y = 0;
#pragma unroll
            for (int i = 0; i < E; i++) {
                const RT xv = (FULL || e0 + i < cnt) ? (RT)xb[skew(e0 + i)] : xp;
                if (i == 0 && first) y = xv;                               // y[1] = x[1]: the first sample passes  (:3592, :3612)
                else if constexpr (HP) y = coef_r * (y + xv - xp);              // :3614
                else y = fmaR(coef_r, xv - y, y);                      // :3594 (fused: the tolerance path)
                xp = xv;
                z[i] = y;
            }
        }
        // Y_t = z_t[E - 1] + M Y_(t-1): inclusive scan over the wave, Y_(-1) = the tile's carry
        RT Y = z[E - 1];
        Y = fmaR(Md[0], dpp_rt<0x111>(Y), Y);
        Y = fmaR(Md[1], dpp_rt<0x112>(Y), Y);
        Y = fmaR(Md[2], dpp_rt<0x114>(Y), Y);
        Y = fmaR(Md[3], dpp_rt<0x118>(Y), Y);
        Y = fmaR(mA, dpp_rt<0x142, 0xA>(Y), Y);
        Y = fmaR(mB, dpp_rt<0x143, 0xC>(Y), Y);
        Y = fmaR(mlane, carry_y, Y);          // true state after this lane's last output
        RT yin = dpp_rt<0x138>(Y);                     // the lane before (lane 0 receives 0)
        if (lane == 0) yin = carry_y;
        float res[E];
        RT ylast = 0;
#pragma unroll
        for (int i = 0; i < E; i++) {
            const RT yv = fmaR(mp[i + 1], yin, z[i]);
            res[i] = (float)yv;
            if (emit && (FULL || e0 + i < cnt)) mxf = fmaxf(mxf, fabsf(res[i]));
            if (!FULL && e0 + i == cnt - 1) ylast = yv;
        }
        if constexpr (JOBS) {
            if (P.epi) {   // stream.flac: dest[d] = clamp(s * (s < 0 and 128 or 127), -128, 127)  (:3181; the recurrence itself carries s)
#pragma unroll
                for (int i = 0; i < E; i++) res[i] = __builtin_amdgcn_fmed3f(res[i] * (res[i] < 0.f ? 128.f : 127.f), -128.f, 127.f);
            }
        }
        // the lane that holds the tile's last output hands its state to the next tile
        if constexpr (FULL) {
            if constexpr (R32) carry_y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Y), 63));
            else carry_y = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(Y) >> 32), 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)__double_as_longlong(Y), 63));
        }
        else carry_y = __shfl(ylast, (cnt - 1) / E);
        carry_x = (RT)xb[skew(cnt - 1)];
        if constexpr (NW > 1) {
            if (emit) {
                __syncthreads();   // (every wave has read the tile before out of `mix`: one buffer, two barriers a tile, and the workgroup's LDS lets every wave of config 5 be resident at once — with two buffers 14 of the 16 waves per CU were)
                float *const mw = mix + (size_t)wv * MIXN;
#pragma unroll
                for (int i = 0; i < E; i++) mw[skew(e0 + i)] = res[i];
            }
        } else
        if (!emit) {}
        else if constexpr (FULL) {
            // a full tile leaves through LDS once more: a lane's eight consecutive results as they stand are two 16-byte stores at a stride of 32 bytes —
            // each store instruction touches every line of the tile; transposed back (the skewed layout of the way in, read the other way round) a
            // store instruction writes 256 contiguous bytes
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < E; i++) xb[skew(e0 + i)] = res[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < E; u++) held[u] = xb[skew(lane + 64 * u)];
            held_at = orow + o0 + (unsigned)lane;
        } else if (e0 < cnt) {
            float *op = orow + o0 + (unsigned)e0;
            if (e0 + E <= cnt) {
                reinterpret_cast<float4 *>(op)[0] = make_float4(res[0], res[1], res[2], res[3]);
                reinterpret_cast<float4 *>(op)[1] = make_float4(res[4], res[5], res[6], res[7]);
            } else {
                for (int i = 0; i < E && e0 + i < cnt; i++) op[i] = res[i];
            }
        }
        };
        if (cnt == T) compute(std::true_type{}); else compute(std::false_type{});
        if constexpr (NW > 1) { if (emit) { mix_cnt = cnt; mix_o0 = o0; mix_par = mix_par_next; mix_par_next ^= 1; } }
        landed(pva);   // (requested before this tile's arithmetic)
        kb = kb_n; r0 = r0_n; vcur = vnxt;
    }
    if (held_at) {
#pragma unroll
        for (int u = 0; u < E; u++) held_at[64 * u] = held[u];
    }
    if constexpr (NW > 1) { if (mix_cnt) mix_out(); }
    for (int o = 32; o; o >>= 1) mxf = fmaxf(mxf, __shfl_xor(mxf, o));
    if constexpr (!JOBS) {
    if (lane == 0) atomicMax(&P.rowmax[r], (unsigned long long)__double_as_longlong((double)mxf));
    if constexpr (NW > 1) { if (lane == 0) atomicMax(&P.rowmax2[s], (unsigned long long)__double_as_longlong((double)mxf)); }
    }   // the largest |stored value|, as k_onepole reports it (bit patterns of non-negative doubles order like the values; zeroed by the host)
    };   // run_item
    if constexpr (LOOPJ) { for (unsigned item = blockIdx.x; item < P.njobs; item += gridDim.x) run_item(item); }   // (short jobs: several per workgroup)
    else run_item(blockIdx.x);   // (not a loop that runs once: as one hipcc spilled 15 - 18 registers of the audio kernels and 7 of the long-job ones)
}

// ---------------------------------------------------------------- the lazy state
// gives the rows' buffer back to the context (when its own scratch is empty or smaller) or frees it; clears the state
void lazy_drop(aukit_ctx *ctx, aukit_audio *a) {
    a->lazy_rs = false;
    a->lazy_fx = 0;
    if (a->lazy_rows.p) {
        DevBuf *home = ctx ? (a->lazy_indirect ? &ctx->tmp_buf3 : &ctx->tmp_buf) : nullptr;   // where the buffer came from
        if (home && home->cap < a->lazy_rows.cap) { home->release(); *home = a->lazy_rows; a->lazy_rows = DevBuf{}; }
        else a->lazy_rows.release();
    }
    a->lazy_row_off.clear(); a->lazy_row_len.clear();
    a->lazy_indirect = false;   // (lazy_tab keeps its allocation for the next call: freeing it here would wait for the kernel that reads it)
    a->lazy_scratch16 = false;
}

// the owed resample with the ordinary kernel, into the audio's own rows
int lazy_materialize(aukit_ctx *ctx, aukit_audio *a) {
    if (!a->lazy_rs) return AUKIT_OK;
    if (!ctx) ctx = ctx_is_live(a->lazy_ctx, a->lazy_ctx_id) ? a->lazy_ctx : nullptr;
    if (!ctx) return fail(AUKIT_E_ARG, "audio has a deferred resample and no context to run it with");
    { int orc = owner_ready(ctx, a->lazy_ctx, a->lazy_ctx_id); if (orc) return orc; }   // the decoder wrote the rows on lazy_ctx's stream
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    if (a->lazy_indirect) {   // frame by frame in the fused decoder's scratch: contiguous rows first (k_flac_gather)
        int grc;
        if ((grc = ctx->tmp_buf.ensure((size_t)a->lazy_tot * 4 + 256))) return grc;
        const char *T = reinterpret_cast<const char *>(a->lazy_tab.p);
        if ((grc = ctx_begin_kernel(ctx))) return grc;
        if ((grc = flac_gather_launch(ctx, reinterpret_cast<const FrameRec *>(T), a->lazy_nfr, a->channels, reinterpret_cast<const u64 *>(T + a->lazy_o_rowoff),
                                      reinterpret_cast<const int *>(a->lazy_rows.p), reinterpret_cast<int *>(ctx->tmp_buf.p), a->lazy_scratch16))) return grc;
        if (a->lazy_scratch16) { a->lazy_scratch16 = false; a->lazy_src = SRC_I32; a->lazy_norm_pos = a->lazy_norm_neg = a->lazy_full; }   // (gathered to int32 rows: the ordinary kernels take those)
        if ((grc = ctx_end_kernel(ctx, "k_flac_gather", 2 * a->lazy_tot * 4))) return grc;
        if (ctx->tmp_buf3.cap < a->lazy_rows.cap) {   // the scratch goes home
            ctx->tmp_buf3.release(); ctx->tmp_buf3 = a->lazy_rows; a->lazy_rows = DevBuf{};
            if (ctx->scratch_ev && hipEventRecord(ctx->scratch_ev, ctx->stream) == hipSuccess) ctx->scratch_ev_set = true;   // (behind the gather above)
        }
        else a->lazy_rows.release();
        a->lazy_indirect = false;
    } else {
        // the rows move back into the context's scratch: the wave kernels of audio_from_int_rows take them from there
        ctx->tmp_buf.release();
        ctx->tmp_buf = a->lazy_rows;
        a->lazy_rows = DevBuf{};
    }
    a->lazy_rs = false;
    const std::vector<uint64_t> ro = a->lazy_row_off, rl = a->lazy_row_len;
    aukit_audio *self = a;
    ctx->lazy_suppress = true;
    const int mrc = audio_from_int_rows(ctx, a->lazy_src, ctx->tmp_buf.p, ro, rl, a->n, a->channels, a->lazy_rate, a->rate, a->lazy_interp, true, AUKIT_F32, a->lazy_norm_pos, a->lazy_norm_neg, &self);
    ctx->lazy_suppress = false;
    return mrc;
}

// after the decoder has left int32 rows in ctx->tmp_buf: shape *out as the resampled audio and leave the resample owed.  false: not this shape
bool lazy_resample_try(aukit_ctx *ctx, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len, uint32_t n, int C, double rate, double new_rate, int interp,
                       double full, aukit_audio **out, int *rc, const LazyFrames *LF, int src_kind, double norm_pos, double norm_neg) {
    *rc = AUKIT_OK;
    if (getenv("AUKIT_NO_TAIL_FUSION") || ctx->exact_math || ctx->lazy_suppress || (interp != AUKIT_INTERP_LINEAR && interp != AUKIT_INTERP_CUBIC) || n == 0) return false;
    FastParams F;
    const int fsrc = src_kind == SRC_I16 ? SRC_PCM_S16LE_MONO : (src_kind == SRC_I8 ? SRC_PCM8_MONO : SRC_I32);   // the wave kernel that would run otherwise
    if (!fast_eligible(fsrc, interp, rate, new_rate, F)) return false;
    if (src_kind == SRC_I32) {
        int e = 0;
        if (std::frexp(full, &e) != 0.5 || full > 16777216.0) return false;   // v / full must be an exact f32 operation (as fast_try asks)
        norm_pos = norm_neg = full;
    } else if (!((src_kind == SRC_I16 && norm_pos == 32767 && norm_neg == 32768) || (src_kind == SRC_I8 && norm_pos == 127 && norm_neg == 128))) return false;
    const double ratio = new_rate / rate;
    std::vector<uint64_t> lens(n);
    for (uint32_t s = 0; s < n; s++) {
        const uint64_t Ls = row_len[(size_t)s * C];
        for (int c = 1; c < C; c++) if (row_len[(size_t)s * C + c] != Ls) return false;
        const double nl = (double)Ls * ratio;
        lens[s] = nl >= 1 ? (uint64_t)std::floor(nl) : 0;                   // newlen uses #data[1]  :659
        if (Ls == 0 || lens[s] == 0) return false;                          // (the ordinary path has the reference's answers for empty rows)
        if (std::floor(((double)(lens[s] - 1)) / ratio + 1) > (double)Ls) return false;
    }
    if (LF) {
        // frame-by-frame rows are read in place only where a tile's window (k_rs_onepole: 512 outputs) spans at most two frames of its stream
        if (!LF->uniform || !LF->nfr) return false;
        const uint64_t capw = std::max<uint64_t>(512, (512ull * F.a) / F.b + 19);
        for (uint32_t s = 0; s < n; s++) if ((*LF->nframes)[s] > 1 && (uint64_t)(*LF->bs0)[s] < capw) return false;   // (a stream of one frame is one run of samples)
    }
    aukit_audio *a = *out;
    if ((*rc = audio_prepare(ctx, &a, n, C, new_rate, AUKIT_F32, lens.data()))) return true;
    *out = a;
    a->lazy_rows.release();
    a->lazy_indirect = false;
    if (LF) {
        // the scratch leaves the context with the audio, and so do the records that say where its frames lie
        Carve cv;
        const size_t o_fr = cv.take(LF->nfr * sizeof(FrameRec)), o_fb = cv.take(((size_t)n + 1) * 8), o_bs = cv.take((size_t)n * 4), o_ro = cv.take((size_t)n * C * 8);
        (void)o_fr;
        if ((*rc = a->lazy_tab.ensure(cv.at))) return true;
        char *T = reinterpret_cast<char *>(a->lazy_tab.p);
        std::vector<uint64_t> fb(n + 1, 0);
        if (hipMemcpyAsync(T, LF->d_frames, LF->nfr * sizeof(FrameRec), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(T + o_fb, LF->d_fbase, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(T + o_ro, LF->d_rowoff, (size_t)n * C * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "copy of the frame records failed"); return true; }
        const uint64_t nfr = LF->nfr;
        if ((*rc = h2d_table(ctx, T + o_fb + (size_t)n * 8, &nfr, 8))) return true;          // fbase[n] = the number of records
        if ((*rc = h2d_table(ctx, T + o_bs, LF->bs0->data(), (size_t)n * 4))) return true;
        a->lazy_nfr = LF->nfr; a->lazy_tot = LF->tot_elems;
        a->lazy_min_bs = 0x7FFFFFFF;
        for (uint32_t s = 0; s < n; s++) if ((*LF->nframes)[s] > 1) a->lazy_min_bs = std::min(a->lazy_min_bs, (*LF->bs0)[s]);   // (k_rsp's window is wider than k_rs_onepole's)
        a->lazy_o_fbase = o_fb; a->lazy_o_bs0 = o_bs; a->lazy_o_rowoff = o_ro;
        a->lazy_indirect = true;
        a->lazy_rows = ctx->tmp_buf3;
        ctx->tmp_buf3 = DevBuf{};
        ctx->scratch_dirty = false;   // (its readers from here on record scratch_ev: lazy_onepole_try, lazy_materialize)
    } else {
        a->lazy_rows = ctx->tmp_buf;     // the rows leave the context's scratch with the audio: nothing can overwrite them
        ctx->tmp_buf = DevBuf{};
    }
    a->lazy_row_off = row_off; a->lazy_row_len = row_len;
    a->lazy_rate = rate; a->lazy_full = full; a->lazy_interp = interp; a->lazy_ctx = ctx; a->lazy_ctx_id = ctx->id;
    a->lazy_src = src_kind; a->lazy_norm_pos = norm_pos; a->lazy_norm_neg = norm_neg;
    a->lazy_scratch16 = LF && LF->scratch16;
    if (a->lazy_scratch16) a->lazy_src = SRC_I16;   // (k_rs_onepole<..., short> on the frames; v / full as for int32 rows)
    a->lazy_rs = true;
    ctx->last_kernel = "(resample deferred)";
    return true;
}

// effects.highpass / lowpass on an audio whose resample is owed: both in one pass.  false: not taken (the caller materialises and filters)
// mono_out (round 4, late): the two channels' mean goes to that (prepared, F32, one channel) audio instead of `a`'s own rows, `a` stays as it is —
// rows still owed — and mono_out's row maxima receive the larger channel maximum of every stream (k_rs_onepole<..., 2>)
bool lazy_onepole_try(aukit_ctx *ctx, aukit_audio *a, double coef, bool highpass, int *rc, aukit_audio *mono_out) {
    *rc = AUKIT_OK;
    if (!a->lazy_rs || a->dtype != AUKIT_F32) return false;
    if (mono_out && (a->channels != 2 || mono_out->dtype != AUKIT_F32 || mono_out->channels != 1 || mono_out->n != a->n)) return false;
    if ((*rc = owner_ready(ctx, a->lazy_ctx, a->lazy_ctx_id))) return true;
    FastParams F;
    if (!fast_eligible(a->lazy_src == SRC_I16 ? SRC_PCM_S16LE_MONO : (a->lazy_src == SRC_I8 ? SRC_PCM8_MONO : SRC_I32), a->lazy_interp, a->lazy_rate, a->rate, F)) return false;
    constexpr int T = 512;
    if (((double)F.b + (double)T * (double)F.a) * (double)F.b >= 4294967296.0) return false;   // exact (q, rem) inside a tile
    const int cap = std::max(512, ((int)(((unsigned long long)T * F.a) / F.b) + 16 + 3) & ~3) + 32;   // (+ 32: the slack around the window that whole vectors may spill into)
    const bool tabw = a->lazy_interp == AUKIT_INTERP_CUBIC && F.b <= 512 && !getenv("AUKIT_RS_HORNER");
    const int NWh = mono_out ? 2 : 1;
    const size_t wave_lds = (size_t)cap + T + T / 8 + 8;
    const size_t lds = ((size_t)NWh * wave_lds + (tabw ? ((4 * (size_t)F.b + 3) & ~(size_t)3) : 0) + (mono_out ? (size_t)NWh * (T + T / 8 + 8) : 0)) * 4;
    if (lds > 60 * 1024) return false;
    for (uint64_t l : a->lazy_row_len) if (l > 0x7FFFFFF0ull) return false;
    if ((*rc = audio_rowmax_ensure(a))) return true;
    if (mono_out && (*rc = audio_rowmax_ensure(mono_out))) return true;
    if (hipSetDevice(ctx->device) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "hipSetDevice failed"); return true; }
    const size_t rows = (size_t)a->n * a->channels;
    std::vector<uint64_t> tab(a->lazy_row_off);
    tab.insert(tab.end(), a->lazy_row_len.begin(), a->lazy_row_len.end());
    if ((*rc = upload_table(ctx, ctx->misc_buf, tab.data(), tab.size() * 8))) return true;
    RsOnepoleParams P{};
    P.rows = a->lazy_rows.p;
    P.row_off = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
    P.row_len = P.row_off + rows;
    P.a_meta = reinterpret_cast<const unsigned long long *>(mono_out ? mono_out->d_meta : a->d_meta);
    P.out = reinterpret_cast<float *>(mono_out ? mono_out->dev : a->dev);
    P.rowmax = reinterpret_cast<unsigned long long *>(a->d_rowmax);
    P.rowmax2 = mono_out ? reinterpret_cast<unsigned long long *>(mono_out->d_rowmax) : nullptr;
    P.wave_lds = (int)wave_lds;
    P.n = a->n; P.C = a->channels; P.cap = cap;
    P.fa = F.a; P.fb = F.b; P.fmagic = F.magic; P.inv_b = F.inv_b;
    P.dq256 = (unsigned)((64ull * F.a) / F.b); P.dr256 = (unsigned)((64ull * F.a) % F.b);   // the step of 64 outputs (one row of lanes)
    // (1 / 32767 etc. rounded to f32 once: the constants of fast.hip / fast_wave_dev.h)
    P.scale = a->lazy_scratch16 ? (float)(1.0 / a->lazy_full) : (a->lazy_src == SRC_I16 ? 1.0f / 32767.0f : (a->lazy_src == SRC_I8 ? 1.0f / 127.0f : (float)(1.0 / a->lazy_full)));
    P.scale_neg = a->lazy_scratch16 ? P.scale : (a->lazy_src == SRC_I16 ? 1.0f / 32768.0f : (a->lazy_src == SRC_I8 ? 1.0f / 128.0f : P.scale));
    P.fr_mul = a->lazy_scratch16 ? 2 : 1;
    P.coef = coef;
    if (a->lazy_indirect) {
        const char *T = reinterpret_cast<const char *>(a->lazy_tab.p);
        P.frames = reinterpret_cast<const FrameRec *>(T);
        P.fbase = reinterpret_cast<const unsigned long long *>(T + a->lazy_o_fbase);
        P.bs0 = reinterpret_cast<const int *>(T + a->lazy_o_bs0);
    }
    if (tabw) {
        std::vector<float> w(4 * (size_t)F.b);
        for (unsigned r = 0; r < F.b; r++) {
            const long double f = (long double)r / (long double)F.b, f2 = f * f, f3 = f2 * f;
            w[4 * r] = (float)(-0.5L * f3 + f2 - 0.5L * f); w[4 * r + 1] = (float)(1.5L * f3 - 2.5L * f2 + 1.0L);
            w[4 * r + 2] = (float)(-1.5L * f3 + 2.0L * f2 + 0.5L * f); w[4 * r + 3] = (float)(0.5L * f3 - 0.5L * f2);
        }
        if ((*rc = upload_table(ctx, ctx->tile_buf, w.data(), w.size() * 4))) return true;
        P.wg = reinterpret_cast<const float *>(ctx->tile_buf.p);
    }
    {   // rows cut into runs of tiles (RsOnepoleParams::segs): as many as bring a SMALL launch to the ≈ 4 K waves the chip holds at once (measured: beyond that nothing is gained — 2.94 against 3.06 ms on config 3b, 3.63 against 3.46 on config 5), each at least 8 x its warm-up long
        const double m = highpass ? coef : 1.0 - coef;
        uint64_t min_tiles = ~0ull;
        for (uint64_t l : a->len) min_tiles = std::min<uint64_t>(min_tiles, (l + T - 1) / T);
        int warm = 0, segs = 1;
        if (m > 0 && m < 1 && !getenv("AUKIT_RS_ONE_CHAIN")) {
            warm = (int)std::ceil(40.0 * M_LN2 / -std::log(m) / T);
            const uint64_t by_rows = std::max<uint64_t>(1, 4096 / std::max<size_t>(rows, 1))   /* (what the chip holds at once: more runs than that only add warm-up) */, by_len = min_tiles / (8ull * (uint64_t)std::max(warm, 1));
            segs = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min(by_rows, by_len), 16));
        }
        if (getenv("AUKIT_RS_SEGS")) segs = std::max(1, atoi(getenv("AUKIT_RS_SEGS")));
        P.segs = segs; P.warm = segs > 1 ? warm : 0;
        P.novec = getenv("AUKIT_RS_NOVEC") ? 1 : 0;
        if (hipMemsetAsync(a->d_rowmax, 0, rows * 8, ctx->stream) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "hipMemsetAsync failed"); return true; }
        if (mono_out && hipMemsetAsync(mono_out->d_rowmax, 0, (size_t)a->n * 8, ctx->stream) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "hipMemsetAsync failed"); return true; }
    }
    if ((*rc = ctx_begin_kernel(ctx))) return true;
    const size_t ldsb = lds;
 const dim3 grid((unsigned)((rows / (size_t)NWh) * (size_t)P.segs));
    const bool r32 = !highpass && (1.0 - coef) <= 0.5 && (1.0 - coef) >= 0 && !getenv("AUKIT_RS_F64");   // the recurrence in f32 (k_rs_onepole<..., R32>)
    ctx->counters[AUKIT_COUNTER_RECURRENCE_F32] = r32 ? 1 : 0;
#define AUKIT_RSO1(I, H, Tb, S)                                                                                                                            \
    do {                                                                                                                                                     \
        if (!H && r32) {                                                                                                                                     \
            if (mono_out) hipLaunchKernelGGL((k_rs_onepole<I, false, Tb, S, 2, false, false, true>), grid, dim3(128), ldsb, ctx->stream, P);              \
            else hipLaunchKernelGGL((k_rs_onepole<I, false, Tb, S, 1, false, false, true>), grid, dim3(64), ldsb, ctx->stream, P);                         \
        } else if (mono_out) hipLaunchKernelGGL((k_rs_onepole<I, H, Tb, S, 2>), grid, dim3(128), ldsb, ctx->stream, P);                                   \
        else hipLaunchKernelGGL((k_rs_onepole<I, H, Tb, S, 1>), grid, dim3(64), ldsb, ctx->stream, P);                                                     \
    } while (0)
#define AUKIT_RSO(S)                                                                                                                                         \
    do {                                                                                                                                                     \
        if (a->lazy_interp == AUKIT_INTERP_LINEAR) { if (highpass) AUKIT_RSO1(AUKIT_INTERP_LINEAR, true, false, S); else AUKIT_RSO1(AUKIT_INTERP_LINEAR, false, false, S); } \
        else if (tabw) { if (highpass) AUKIT_RSO1(AUKIT_INTERP_CUBIC, true, true, S); else AUKIT_RSO1(AUKIT_INTERP_CUBIC, false, true, S); }              \
        else { if (highpass) AUKIT_RSO1(AUKIT_INTERP_CUBIC, true, false, S); else AUKIT_RSO1(AUKIT_INTERP_CUBIC, false, false, S); }                      \
    } while (0)
    // int16 rows at a ratio whose phases repeat every 320 outputs (44.1 / 22.05 kHz -> 48 kHz): k_rsp (rs_periodic.hip), weights and tap offsets in registers
    uint64_t min_out = ~0ull;
    for (uint64_t l : a->len) min_out = std::min<uint64_t>(min_out, l);
    bool periodic = false;
    if (a->lazy_src == SRC_I16 && tabw) { periodic = rsp_try(ctx, P, highpass, r32, NWh, rows, min_out, a->lazy_indirect ? a->lazy_min_bs : 0, rc); if (periodic && *rc) return true; }
    if (periodic) {}
    else if (a->lazy_src == SRC_I16) AUKIT_RSO(short); else if (a->lazy_src == SRC_I8) AUKIT_RSO(signed char); else AUKIT_RSO(int);
#undef AUKIT_RSO1
#undef AUKIT_RSO
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_rs_onepole launch failed"); return true; }
    // (a frame scratch was read: the FLAC decoder of the NEXT call, on the look-ahead stream, writes such a buffer behind this point of ctx->stream — common.h)
    if (a->lazy_indirect && ctx->scratch_ev && hipEventRecord(ctx->scratch_ev, ctx->stream) == hipSuccess) ctx->scratch_ev_set = true;
    uint64_t in_elems = 0, out_elems = 0;
    for (uint64_t l : a->lazy_row_len) in_elems += l;
    for (uint64_t l : a->len) out_elems += l * (uint64_t)a->channels;
    const uint64_t in_bytes = in_elems * (a->lazy_src == SRC_I16 ? 2 : (a->lazy_src == SRC_I8 ? 1 : 4));
    if (mono_out) {   // `a` keeps everything it owes (its own rows were not written); the mean's maxima are not known (rowmax2 holds the channels')
        mono_out->rowmax_valid = false;
        *rc = ctx_end_kernel(ctx, periodic ? (highpass ? "k_rsp<highpass,mono>" : "k_rsp<lowpass,mono>") : (highpass ? "k_rs_onepole<highpass,mono>" : "k_rs_onepole<lowpass,mono>"), in_bytes + out_elems * 4 / (uint64_t)a->channels);
        return true;
    }
    a->rowmax_valid = true;
    a->lazy_fx = 0;
    lazy_drop(ctx, a);   // the resample is paid; the rows' buffer goes back to the context
    *rc = ctx_end_kernel(ctx, periodic ? (highpass ? "k_rsp<highpass>" : "k_rsp<lowpass>") : (highpass ? "k_rs_onepole<highpass>" : "k_rs_onepole<lowpass>"), in_bytes + out_elems * 4);
    return true;
}

// stream.qoa's / stream.flac's tail (aukit.lua:3312-3330, :3166-3183) on k_rs_onepole<..., JOBS>: interpolate [-> clamp] -> low-pass seeded with the
// history sample -> chunk sample (or the channels' mean), from the decoder's integer rows.  The kernel k_iir_tail_fast (stream_tail.hip) does
// the same with self-contained tiles that warm up (≈ 44 instructions per output, 2.4 ps per output on 4096 QOA streams); this one carries the
// state from tile to tile (1.7 ps).  false: not this shape (the caller keeps k_iir_tail)
static bool rs_onepole_jobs_launch(aukit_ctx *ctx, const void *rows, bool rows_i32, double full, int epi, const TailJob *d_jobs, size_t njobs, bool long_jobs, int mix_channels, double rate,
                                   int interp, double lp_alpha, float *out, uint64_t algorithmic_bytes, const char *name, int *rc) {
    *rc = AUKIT_OK;
    if (getenv("AUKIT_NO_RS_JOBS") || ctx->exact_math || !njobs) return false;
    if (mix_channels != 1 && mix_channels != 2) return false;
    if (interp != AUKIT_INTERP_LINEAR && interp != AUKIT_INTERP_CUBIC) return false;
    FastParams F;
    if (!fast_eligible(rows_i32 ? SRC_I32 : SRC_PCM8_MONO, interp, rate, 48000, F)) return false;
    if (rows_i32) { int e = 0; if (std::frexp(full, &e) != 0.5 || full > 16777216.0) return false; }   // v / full must be an exact f32 operation
    constexpr int T = 512;
    if (((double)F.b + (double)T * (double)F.a) * (double)F.b >= 4294967296.0) return false;
    if (!(lp_alpha > 0 && lp_alpha < 1)) return false;
    const int cap = std::max(512, ((int)(((unsigned long long)T * F.a) / F.b) + 16 + 3) & ~3) + 32;
    const bool tabw = interp == AUKIT_INTERP_CUBIC && F.b <= 512 && !getenv("AUKIT_RS_HORNER");
    const int NWh = mix_channels;
    const size_t wave_lds = (size_t)cap + T + T / 8 + 8;
    const size_t lds = ((size_t)NWh * wave_lds + (tabw ? ((4 * (size_t)F.b + 3) & ~(size_t)3) : 0) + (NWh > 1 ? (size_t)NWh * (T + T / 8 + 8) : 0)) * 4;
    if (lds > 60 * 1024) return false;
    if (hipSetDevice(ctx->device) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "hipSetDevice failed"); return true; }
    RsOnepoleParams P{};
    P.rows = rows;
    P.jobs = d_jobs;
    P.njobs = (unsigned)njobs;
    P.out = out;
    P.n = 0; P.C = 1; P.cap = cap;
    P.fa = F.a; P.fb = F.b; P.fmagic = F.magic; P.inv_b = F.inv_b;
    P.dq256 = (unsigned)((64ull * F.a) / F.b); P.dr256 = (unsigned)((64ull * F.a) % F.b);
    P.scale = P.scale_neg = rows_i32 ? (float)(1.0 / full) : 1.0f;
    P.coef = lp_alpha;
    P.segs = 1; P.warm = 0; P.novec = getenv("AUKIT_RS_NOVEC") ? 1 : 0; P.fr_mul = 1;
    P.wave_lds = (int)wave_lds;
    P.epi = epi;
    P.clo = epi ? -3.0e38f : -128.0f; P.chi = epi ? 3.0e38f : 127.0f;   // (stream.flac does not clamp the interpolated sample, :3177-3178)
    if (tabw) {
        std::vector<float> w(4 * (size_t)F.b);
        for (unsigned r = 0; r < F.b; r++) {
            const long double f = (long double)r / (long double)F.b, f2 = f * f, f3 = f2 * f;
            w[4 * r] = (float)(-0.5L * f3 + f2 - 0.5L * f); w[4 * r + 1] = (float)(1.5L * f3 - 2.5L * f2 + 1.0L);
            w[4 * r + 2] = (float)(-1.5L * f3 + 2.0L * f2 + 0.5L * f); w[4 * r + 3] = (float)(0.5L * f3 - 0.5L * f2);
        }
        if ((*rc = upload_table(ctx, ctx->tile_buf, w.data(), w.size() * 4))) return true;
        P.wg = reinterpret_cast<const float *>(ctx->tile_buf.p);
    }
    if ((*rc = ctx_begin_kernel(ctx))) return true;
    // long jobs (an iterator call of stream.qoa: ~ 100 tiles): a workgroup each — measured 6.8 ms against 7.6 for workgroups that take five in turn;
    // short ones (a FLAC frame: nine tiles): a workgroup takes several in turn and pays the set-up in front of its tile loop once
    const dim3 grid((unsigned)((long_jobs || NWh > 1) ? njobs : std::min<size_t>(njobs, (size_t)ctx->num_cus * 64)));
    const bool r32 = (1.0 - lp_alpha) <= 0.5 && (1.0 - lp_alpha) >= 0 && !getenv("AUKIT_RS_F64");   // the recurrence in f32 (k_rs_onepole<..., R32>)
    ctx->counters[AUKIT_COUNTER_RECURRENCE_F32] = r32 ? 1 : 0;
#define AUKIT_RSJ(I, Tb, S)                                                                                                                           \
    do {                                                                                                                                              \
        if (r32) {                                                                                                                                    \
            if (NWh == 2) hipLaunchKernelGGL((k_rs_onepole<I, false, Tb, S, 2, true, false, true>), grid, dim3(128), lds, ctx->stream, P);            \
            else if (long_jobs) hipLaunchKernelGGL((k_rs_onepole<I, false, Tb, S, 1, true, false, true>), grid, dim3(64), lds, ctx->stream, P);       \
            else hipLaunchKernelGGL((k_rs_onepole<I, false, Tb, S, 1, true, true, true>), grid, dim3(64), lds, ctx->stream, P);                       \
        } else if (NWh == 2) hipLaunchKernelGGL((k_rs_onepole<I, false, Tb, S, 2, true, false>), grid, dim3(128), lds, ctx->stream, P);               \
        else if (long_jobs) hipLaunchKernelGGL((k_rs_onepole<I, false, Tb, S, 1, true, false>), grid, dim3(64), lds, ctx->stream, P);                 \
        else hipLaunchKernelGGL((k_rs_onepole<I, false, Tb, S, 1, true, true>), grid, dim3(64), lds, ctx->stream, P);                                 \
    } while (0)
#define AUKIT_RSJS(S)                                                                                                                                 \
    do {                                                                                                                                              \
        if (interp == AUKIT_INTERP_LINEAR) AUKIT_RSJ(AUKIT_INTERP_LINEAR, false, S);                                                                  \
        else if (tabw) AUKIT_RSJ(AUKIT_INTERP_CUBIC, true, S);                                                                                        \
        else AUKIT_RSJ(AUKIT_INTERP_CUBIC, false, S);                                                                                                 \
    } while (0)
    bool periodic = false;
    if (!rows_i32 && tabw && (long_jobs || NWh > 1)) { periodic = rsp_jobs_try(ctx, P, r32, NWh, grid.x, rc); if (periodic && *rc) return true; }   // (rs_periodic.hip)
    if (periodic) {}
    else if (rows_i32) AUKIT_RSJS(int); else AUKIT_RSJS(signed char);
#undef AUKIT_RSJS
#undef AUKIT_RSJ
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_rs_onepole<jobs> launch failed"); return true; }
    *rc = ctx_end_kernel(ctx, periodic ? (std::string(name) == "k_rs_onepole<qoa>" ? "k_rsp<qoa>" : "k_rsp<jobs>") : name, algorithmic_bytes);
    return true;
}

bool rs_onepole_jobs_try(aukit_ctx *ctx, const void *rows_i8, const std::vector<TailJob> &jobs, int mix_channels, double rate, int interp, double lp_alpha, float *out,
                         uint64_t algorithmic_bytes, const char *name, int *rc) {
    *rc = AUKIT_OK;
    if (getenv("AUKIT_NO_RS_JOBS") || ctx->exact_math || jobs.empty()) return false;
    for (const TailJob &j : jobs) if (j.n <= 0 || j.nout < 0) return false;
    if ((*rc = upload_table(ctx, ctx->misc_buf, jobs.data(), jobs.size() * sizeof(TailJob)))) return true;
    uint64_t sum_out = 0;
    for (const TailJob &j : jobs) sum_out += (uint64_t)j.nout;
    return rs_onepole_jobs_launch(ctx, rows_i8, false, 1.0, 0, reinterpret_cast<const TailJob *>(ctx->misc_buf.p), jobs.size(), sum_out / jobs.size() >= 32 * 512, mix_channels, rate, interp, lp_alpha, out, algorithmic_bytes, name, rc);
}
// stream.flac: int32 rows (v / full), the jobs already on the device (k_flac_tail_jobs)
bool rs_onepole_jobs_try_dev(aukit_ctx *ctx, const void *rows_i32, double full, const TailJob *d_jobs, size_t njobs, double rate, int interp, double lp_alpha, float *out,
                             uint64_t algorithmic_bytes, const char *name, int *rc) {
    return rs_onepole_jobs_launch(ctx, rows_i32, true, full, 1, d_jobs, njobs, false, 1, rate, interp, lp_alpha, out, algorithmic_bytes, name, rc);
}

}  // namespace aukit
