// flac_stream.hip — FLAC frames decoded to FINAL integers in the registers of the lane that reads their bits (round 6).
//
// k_flac_decode (flac_fused.hip, round 4) works in rounds of 32 values: parse them into the lane's LDS row, predict over the row, flush the row
// through a cooperative store phase.  Its census (tools/r06_flac_stats.sh, profiles/r06_flac_phase.txt) says the hot loop is ideal — 28.2 M wave
// turns for 1.806 G values, 64 lanes each — and that two thirds of its 99 VALU instructions per value are NOT in that loop: the cooperative
// window slide (three passes of shuffles and 64-bit line arithmetic: ≈ 400 per round), the row's way through LDS (write, read, write, read), the
// flush's per-row metadata, the prefetch registers of both (80 VGPRs: two waves per SIMD).
// Here the same lane-per-frame walk of decodeFrame (aukit.lua:510-567) keeps every value in registers from the bit stream to the store:
//   * groups of FOUR values: four Rice codes / fixed-width fields (:370-376, :405, :423, :457) are parsed speculatively (nothing is committed
//     if one of them is longer than 32 bits), the prediction (:411-419) runs over them with the history in registers, the wasted-bits shift
//     (:467-469), the stereo decorrelation (:482-497) and the wrap (:501-507) follow, and the four finals go — packed, final bytes — into the lane's
//     128-byte output row in LDS with one ds_write per channel.  (The first version stored them from the lane itself, 8 bytes per lane and
//     group: 38 ms for config 5's 8 — 64 partial-line requests per store instruction; 15.8 ms with cached stores, 10.2 with none.)  The rows
//     leave at the top of the next round as 16-byte stores, four to eight adjacent lanes per row: 64 contiguous bytes per channel, bytes only —
//     nothing is computed in the flush;
//   * the first subframe of a decorrelating stereo frame is parked raw (int32) as before; the second subframe's groups read it back one group
//     ahead of its use (an unconditional load: lanes that do not decorrelate read the scratch's first vector);
//   * the bit-stream window is a ring of 32 dwords per lane in LDS that the LANE ITSELF fills: four 16-byte granules requested a round ahead
//     into registers (unconditional straight-line loads, a lane with nothing to fetch reads the batch's first vector), byte-swapped and
//     written at the top of the next round.  No shuffles, no line bookkeeping across lanes;
//   * everything that is not such a group — frame / subframe / partition headers, warm-up samples, the one to three values that align a run to
//     a multiple of four, codes longer than 32 bits, the last bytes of a stream, CONSTANT subframes — is generic single-value code (64-bit
//     sums, every check), entered a few times per subframe.
// What the kernel does not serve it DECLINES (FE_DECLINE), exactly as k_flac_decode does: predictor orders above 12, values beyond the ranges
// checked below, the :400 quirk, sample depths of a subframe outside 1..31.  Outputs (CandInfo, the scratch layout) are k_flac_decode's: nothing
// downstream knows which of the two ran (AUKIT_FLAC_DECODER=fused brings the round-4 kernel back for the A/B).
#include "flac_stream_dev.h"

namespace aukit {

// One 64-lane workgroup = one wave; a lane owns one candidate frame at a time (k_flac_find's list, a ticket counter, a persistent grid).
#ifndef AUKIT_FS_LB
#define AUKIT_FS_LB 2
#endif
template <bool O16>
__global__ __launch_bounds__(64, AUKIT_FS_LB) void k_flac_stream(const FusedArgs A) {
    constexpr int RV = O16 ? 32 : 16;   // values of a round: what a 64-byte half of the output row holds (int16 / int32 finals)
    __shared__ __attribute__((aligned(16))) unsigned s_win[64 * SWS];
    __shared__ __attribute__((aligned(16))) unsigned s_out[64 * SOS];   // per lane: half A [0, 64), half B [64, 128) — two destinations, or one of 128 bytes
    __shared__ u64 s_dst[64 * 2];        // byte offsets (from A.scratch) of the two halves' destinations
    __shared__ unsigned s_rng[64];       // which pieces of the row are whole (see publish)
    __shared__ unsigned s_end[64];       // values of the round | mode << 8 (the owner's ragged pass)
    const int lane = threadIdx.x;
    const int C = A.C, depth = A.depth;
    const int wrap_half = 1 << (depth - 1), wrap_full = 1 << depth;   // 1 <= depth <= 24 (the host sends nothing else here)
    auto wrap = [&](int v) -> int { return v >= wrap_half ? v - wrap_full : v; };   // :504
    unsigned *const lw = s_win + lane * SWS;

    SRd b;
    b.lw = lw;
    b.wlo = 0; b.whi = 0; b.pos = 0; b.end = 0; b.eof = 0; b.oow = 0;
    u64 limit = ~0ull;
    bool have = false, fresh = false;
    unsigned idx = 0, nolimit = 0;
    int st = S_DONE, status = FE_OK;
    int bs = 0, chan_asgn = 0, nsub = 0, ch = 0;
    int order = 0, wasted = 0, sdepth = 0, lshift = 0, after = S_SUBEND;
    int nparts = 0, psize = 0, pi = 0, param_bits = 4, remaining = 0, jpos = 0, rk = 0, cval = 0, hb = 1 << 23;
    bool fixed = false, wide = false, store_ok = false, lpc = false;
    u64 cand_scratch = 0, end_byte = 0;
    unsigned badacc = 0, bad16 = 0;
    int coef[SMAXO], hist[SMAXO];
#pragma unroll
    for (int q = 0; q < SMAXO; q++) { coef[q] = 0; hist[q] = 0; }
    // the granules requested a round ago
    v4u pf[SPF];
    u64 pf_g0 = 0;
    int pf_n = 0;
    unsigned pf_inr = 0;
#pragma unroll
    for (int i = 0; i < SPF; i++) pf[i] = v4u{0, 0, 0, 0};

    auto start = [&](unsigned rel) {
        have = true;
        idx = A.first + rel;
        const Cand c = A.cands[idx];
        b.end = A.G.base_bit + 8 * A.G.off[c.stream + 1];
        b.pos = A.G.base_bit + 8 * c.byte;
        b.eof = 0; b.oow = 0;
        b.wlo = 0; b.whi = 0;
        pf_n = 0;
        limit = ~0ull;
        nolimit = c.nolimit;
        st = S_FRAME; status = FE_OK; fresh = true;
        bs = 0; chan_asgn = 0; nsub = 0; ch = 0; jpos = 0; remaining = 0;
        store_ok = false;
        cand_scratch = 0; end_byte = 0;
        badacc = 0;
    };
    auto finish = [&]() {
        CandInfo f;
        f.end_byte = end_byte; f.scratch = cand_scratch; f.sample_off = 0;
        f.blocksize = bs; f.chan_asgn = chan_asgn; f.status = status; f.nsub = nsub;
        f.seq = 0; f.used = 0;
        A.ci[idx] = f;
        have = false;
    };
    auto take = [&]() {   // every lane without a frame takes a ticket: one atomic per wave
        const u64 m = __ballot(!have);
        if (!m) return;
        unsigned base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(A.ticket, (unsigned)__builtin_popcountll(m));
        base = __shfl(base, __builtin_ctzll(m));
        const unsigned rel = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1));
        if (!have && rel < A.count) start(rel);
    };
    take();

    // ---- where a value of the running subframe goes.  mode 0: wrap and store; 1: park the first subframe of a decorrelating frame; 2: decorrelate
    // with the parked values and store both channels.  O16: the finals are int16 in the first half of the frame's region (channel c at
    // 2 * scratch + c * bs, in int16 units), the parked values int32 in its second half; else int32 finals, parked in the second channel's place.
    // A round's values are those of ONE subframe with indices in [R0, R0 + RV), R0 a multiple of RV: the lane's output row holds them as the bytes
    // that go to memory — O16: int16 finals of the (first) channel in half A, of the second channel (mode 2) in half B, parked int32 values over
    // both halves; else int32 finals or parked values in half A, the second channel's finals in half B.
    int mode = 0;
    bool dswap = false;
    int dsh = 0, dmask = 0;
    u64 o_fin = 0, o_park = 0;   // element offsets of the subframe's value 0: finals (int16 / int32 units), parked values (int32 units)
    int sv_lo = -1, sv_hi = 0;   // the subframe indices staged in this round: [sv_lo, sv_hi), sv_lo < 0: none
    unsigned *const orow = s_out + lane * SOS;
    auto set_outputs = [&]() {
        const bool decor = C == 2 && chan_asgn >= 8 && chan_asgn <= 10;
        mode = !decor ? 0 : (ch == 0 ? 1 : 2);
        dswap = chan_asgn == 9; dsh = chan_asgn == 10 ? 1 : 0; dmask = chan_asgn == 9 ? 0 : -1;
        if constexpr (O16) { o_fin = 2 * cand_scratch + (mode == 0 ? (u64)ch * (u64)bs : 0ull); o_park = cand_scratch + (u64)bs; }
        else { o_fin = cand_scratch + (mode == 0 ? (u64)ch * (u64)bs : 0ull); o_park = cand_scratch + (u64)bs; }
    };
    // the finals' range (O16): smallest and largest value of the round, looked at when the round is published (a parked value is no final)
    int fmin = 0, fmax = 0;
    auto staged = [&](int n) { if (sv_lo < 0) sv_lo = jpos; sv_hi = jpos + n; };
    // one value (the generic path)
    auto emit1 = [&](int o) {
        const int jr = jpos & (RV - 1);
        if constexpr (O16) {
            short *hrow = reinterpret_cast<short *>(orow);
            // (fmin / fmax: at 16 bits the values BEFORE the wrap, bounds -32768 and 98303 as in the group loop; else the wrapped ones)
            if (mode == 0) { const int w = wrap(o), t = depth == 16 ? o : w; fmin = min(fmin, t); fmax = max(fmax, t); hrow[jr] = (short)w; }
            else if (mode == 1) orow[jr] = (unsigned)o;
            else {
                int l, r;
                sdecor(dswap, dsh, dmask, store_ok ? A.scratch[o_park + (u64)jpos] : 0, o, l, r);
                const int wl = wrap(l), wr = wrap(r), tl = depth == 16 ? l : wl, tr = depth == 16 ? r : wr;
                fmin = min(fmin, min(tl, tr)); fmax = max(fmax, max(tl, tr));
                hrow[jr] = (short)wl; hrow[32 + jr] = (short)wr;
            }
        } else {
            if (mode == 0) orow[jr] = (unsigned)wrap(o);
            else if (mode == 1) orow[jr] = (unsigned)o;
            else { int l, r; sdecor(dswap, dsh, dmask, store_ok ? A.scratch[o_fin + (u64)bs + (u64)jpos] : 0, o, l, r); orow[jr] = (unsigned)wrap(l); orow[16 + jr] = (unsigned)wrap(r); }
        }
        staged(1);
    };
    // the address of the parked values of the group at `j` (mode 2), else the scratch's first vector
    auto park_addr = [&](int j, bool on) -> const int * {
        const u64 e = O16 ? o_park + (u64)j : o_fin + (u64)bs + (u64)j;
        return A.scratch + ((on && mode == 2 && store_ok && (bs & 3) == 0) ? e : 0ull);
    };
    // what the round staged: where its bytes go, and which aligned pieces of them are whole (at the end of the round; the rows leave at the top
    // of the next one, behind its wait).  s_rng: bits 0-7 / 8-15 the 8-byte pieces of half A / B that are whole, 16 / 17 the half's destination
    // lies at a multiple of 16, 18 something is left for the owner (single values at a subframe's ragged ends, a last frame's odd block size)
    auto publish = [&]() {
        unsigned m = 0;
        u64 dA = 0, dB = 0;
        if (sv_lo >= 0 && store_ok && !(A.dbg & 2)) {
            const int R0 = sv_lo & ~(RV - 1);
            const int lo = sv_lo - R0, hi = sv_hi - R0;   // 0 <= lo < hi <= RV
            int la, ha, lb, hb2;                          // byte ranges of the two halves
            if (O16 && mode == 1) {   // int32 over both halves
                dA = 4 * (o_park + (u64)R0); dB = dA + 64;
                la = min(4 * lo, 64); ha = min(4 * hi, 64); lb = max(4 * lo, 64) - 64; hb2 = max(4 * hi, 64) - 64;
            } else {
                const int es = O16 ? 2 : 4;
                const u64 eA = (!O16 && mode == 1) ? o_fin + (u64)bs : o_fin;
                dA = (u64)es * (eA + (u64)R0);
                dB = (u64)es * (o_fin + (u64)bs + (u64)R0);
                la = es * lo; ha = es * hi;
                lb = mode == 2 ? la : 0; hb2 = mode == 2 ? ha : 0;
            }
            auto pieces = [](int l, int h, u64 d) -> unsigned {   // the whole 8-byte pieces of [l, h) when d lies at a multiple of 8
                if (h <= l || (d & 7)) return 0u;
                return ((1u << (h >> 3)) - 1u) & ~((1u << ((l + 7) >> 3)) - 1u);
            };
            const unsigned pa = pieces(la, ha, dA), pb = pieces(lb, hb2, dB);
            const bool rag = (ha > la && (((la | ha) & 7) || (dA & 7))) || (hb2 > lb && (((lb | hb2) & 7) || (dB & 7)));
            m = pa | (pb << 8) | ((dA & 15) == 0 ? 1u << 16 : 0u) | ((dB & 15) == 0 ? 1u << 17 : 0u) | (rag ? 1u << 18 : 0u)
                | ((unsigned)la << 19) | ((unsigned)lb << 26);   // (where a ragged half begins: la, lb <= 64 in seven bits each — only the owner reads them)
            if constexpr (O16) { if (mode != 1 && (fmin < -32768 || fmax > (depth == 16 ? 98303 : 32767))) bad16 = 1; }   // (at 16 bits the group loop leaves the wrap to the truncation)
        }
        fmin = 0; fmax = 0;
        s_dst[2 * lane] = dA; s_dst[2 * lane + 1] = dB;
        s_rng[lane] = m;
        s_end[lane] = (unsigned)(sv_lo >= 0 ? (sv_hi - (sv_lo & ~(RV - 1))) : 0) | ((unsigned)mode << 8);
        sv_lo = -1;
    };
    // the rows of the round before: 16 bytes per lane and instruction, eight lanes per row — four per 64-byte half
    bool have_flush = false;
    auto flush = [&]() {
        __builtin_amdgcn_wave_barrier();
        const unsigned own = s_rng[lane];
        // the usual round: every row's halves are whole (eight pieces at a multiple of 16) or empty — a lane stores its 16 bytes or nothing
        const unsigned pa_ = own & 0xFFu, pb_ = (own >> 8) & 0xFFu;
        const bool usual = !((own >> 18) & 1u) && (pa_ == 0u || (pa_ == 0xFFu && ((own >> 16) & 1u))) && (pb_ == 0u || (pb_ == 0xFFu && ((own >> 17) & 1u)));
        if (__all(usual)) {
            if (__any(own != 0u)) {
                const int part = lane & 7, half = part >> 2, c = part & 3;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int r = 8 * i + (lane >> 3);
                    if (((s_rng[r] >> (8 * half)) & 0xFFu) != 0u) {
                        const u64 dst = s_dst[2 * r + half];
                        sstore(*reinterpret_cast<const v4u *>(s_out + r * SOS + 16 * half + 4 * c), reinterpret_cast<v4u *>(reinterpret_cast<char *>(A.scratch) + dst + (u64)(16 * c)));
                    }
                }
            }
        } else if (__any(own != 0u)) {
            const int part = lane & 7, half = part >> 2, c = part & 3;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int r = 8 * i + (lane >> 3);
                const unsigned m = s_rng[r];
                const unsigned bits = (m >> (8 * half + 2 * c)) & 3u;
                if (__any(bits != 0u)) {
                    const u64 dst = s_dst[2 * r + half];
                    const unsigned *src = s_out + r * SOS + 16 * half + 4 * c;
                    char *g = reinterpret_cast<char *>(A.scratch) + dst + (u64)(16 * c);
                    const bool a16 = (m >> (16 + half)) & 1u;
                    if (bits == 3u && a16) sstore(*reinterpret_cast<const v4u *>(src), reinterpret_cast<v4u *>(g));
                    else {   // a round cut short by its window ends on a group (8 bytes of int16 finals); a destination at 8 modulo 16
                        if (bits & 1u) sstore(*reinterpret_cast<const v2u *>(src), reinterpret_cast<v2u *>(g));
                        if (bits & 2u) sstore(*reinterpret_cast<const v2u *>(src + 2), reinterpret_cast<v2u *>(g + 8));
                    }
                }
            }
            if (__any((own >> 18) & 1u)) {   // what no aligned 8 bytes cover, by the owner, two bytes at a time (serves both element sizes)
                if ((own >> 18) & 1u) {
                    const u64 dA = s_dst[2 * lane], dB = s_dst[2 * lane + 1];
                    const unsigned e = s_end[lane];
                    const int n = (int)(e & 0xFFu), md = (int)(e >> 8);
                    for (int h = 0; h < 2; h++) {
                        const u64 dst = h ? dB : dA;
                        const unsigned pm = (own >> (8 * h)) & 0xFFu;
                        const int lo = (int)((own >> (h ? 26 : 19)) & 0x7Fu);
                        int hi;   // the half's valid bytes end here
                        if (O16 && md == 1) hi = h ? max(4 * n, 64) - 64 : min(4 * n, 64);
                        else hi = (h == 0 || md == 2) ? (O16 ? 2 : 4) * n : 0;
                        for (int bb = lo; bb < hi; bb += 2)
                            if (!((pm >> (bb >> 3)) & 1u))
                                *reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(A.scratch) + dst + (u64)bb) = reinterpret_cast<const unsigned short *>(orow + 16 * h)[bb >> 1];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // the ranges the prediction relies on, checked where a subframe ends and where a round ends (hb belongs to the subframe)
    auto check_bad = [&]() {
        if (badacc & ~(2u * (unsigned)hb - 1u)) { status = FE_DECLINE; st = S_DONE; }
        badacc = 0;
    };

    // ---- the group loop: MAXO taps for every lane (coefficients beyond a lane's order are zero); WIDE: 64-bit sums and range checks (any
    // coefficients, any shift), otherwise one 24-bit multiply-add per tap — exact while |value| < hb, a power of two with hb * sum |coef| < 2^31
#ifdef AUKIT_FLAC_STATS
    u64 st_rounds = 0, st_outer = 0, st_turns = 0, st_singles = 0, st_lane_turns = 0, st_single_turns = 0;
#define SSTAT(x) x
#define SSTATW(c) { if ((int)__builtin_amdgcn_readfirstlane(lane) == lane) c++; }
#else
#define SSTAT(x)
#define SSTATW(c)
#endif
    int rend = 0;        // the round's values of the running subframe end at this index (a multiple of RV)
    bool slow_lane = false;   // the groups met a code longer than 32 bits: the lane's next value is the generic reader's
    bool rdone = false;
    auto groups_loop = [&](auto MO, auto WD, bool elig) {
        constexpr int MAXO = decltype(MO)::value;
        constexpr bool WIDE = decltype(WD)::value;
        const u64 dfull = (b.pos - 1) >> 5;                     // the dword that holds the last bit read (frames start far beyond bit 0)
        unsigned d = (unsigned)dfull;
        int s = (int)((0u - (unsigned)b.pos) & 31u);            // bits of it not yet read
        unsigned w0 = lw[d & SRING], w1 = lw[(d + 1) & SRING];
        const unsigned whi32 = (unsigned)b.whi;
        int groups = min(remaining, rend - jpos) >> 2;
        bool slow = false;
        bool go = elig && groups > 0 && (int)(whi32 - d) >= SNEED;   // a group reads the dwords d .. d + 6 at most
        // the lane's constants of the run: a value is a Rice code (z zeros, a one, rk bits) or a field of rk bits
        const int zmask = fixed ? 0 : -1, rk1 = fixed ? rk : rk + 1;
        const unsigned krm = 31u - (unsigned)rk, fo = 32u - (unsigned)rk;   // (the bit-field extracts take their offsets modulo 32)
        // ... and of its outputs.  Lanes that do not decorrelate (modes 0 and 1) go through the same arithmetic with a parked value of 0:
        // X - ((Y >> 0) & 0) = 0 for the second channel, that + Y = the value for the first
        const bool any2 = __any(elig && mode == 2) && !(A.dbg & 4);
        const int amask = mode == 2 ? -1 : 0, dm = mode == 2 ? dmask : 0, dshl = mode == 2 ? dsh : 0;
        const bool dsw = mode == 2 && dswap;
        const int g0 = groups;
        unsigned *op = orow + (mode == 1 ? (jpos & (RV - 1)) : ((O16 ? (jpos & (RV - 1)) >> 1 : (jpos & (RV - 1)))));   // where the group's bytes go in the row
        const int ostep = (O16 && mode != 1) ? 2 : 4;           // dwords per group
        const int j0 = jpos;
        v4u tp = v4u{0, 0, 0, 0};
        if (any2) tp = *reinterpret_cast<const v4u *>(park_addr(jpos, go));
        while (go) {
            SSTATW(st_turns) SSTAT(st_lane_turns++;)
            unsigned d_ = d, w0_ = w0, w1_ = w1;
            int s_ = s;
            int totmax = 0;
            int res[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned wn = lw[(d_ + 2) & SRING];             // for a crossing at the END of this value: requested first, used last
                const unsigned hi = __builtin_amdgcn_alignbit(w0_, w1_, (unsigned)s_);
                const int z = hi ? __builtin_clz(hi) : 32;
                const unsigned low = __builtin_amdgcn_ubfe(hi, krm - (unsigned)z, (unsigned)rk);
                const unsigned ur = ((unsigned)z << rk) | low;
                const int v_r = (int)(ur >> 1) ^ -(int)(ur & 1u);
                const int v_f = __builtin_amdgcn_sbfe((int)hi, fo, (unsigned)rk);
                const int tot = (z & zmask) + rk1;
                totmax = max(totmax, tot);                            // beyond 32: a Rice code the generic reader takes (nothing of this group is kept)
                res[j] = fixed ? v_f : v_r;
                s_ -= tot;
                const bool cross = s_ < 0;
                s_ &= 31;
                d_ += cross ? 1u : 0u;
                w0_ = cross ? w1_ : w0_;
                w1_ = cross ? wn : w1_;
            }
            const bool ok = totmax <= 32;
            if (ok) {
                d = d_; s = s_; w0 = w0_; w1 = w1_;
                int nv[4], out[4];
                if constexpr (WIDE) {
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        long long sum = 0;
#pragma unroll
                        for (int q = 0; q < MAXO; q++) { const int t = q < jj ? nv[jj - 1 - q] : hist[q - jj]; sum += (long long)t * (long long)coef[q]; }
                        const long long pr = lshift >= 0 ? (sum >> lshift) : (sum << (-lshift));   // floor(sum / 2^shift)
                        const long long vv = (long long)res[jj] + pr;
                        const long long oo = vv << wasted;
                        if ((unsigned long long)(oo + (1ll << 29)) >= (1ull << 30) || (unsigned long long)(vv + (1ll << 29)) >= (1ull << 30)) badacc |= 0x80000000u;
                        nv[jj] = (int)vv; out[jj] = (int)oo;
                    }
                } else {
                    // the taps on values of BEFORE these four first — four sums that depend on nothing computed here, their multiply-adds interleaved
                    // (one chain per sample: hipcc pads a chain of dependent v_mad_i32_i24 with s_nop) — the taps on the samples just restored last.
                    // Integer sums: any order gives the same bits
                    int sm[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int q = MAXO - 1; q >= 0; q--) {
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) if (q >= jj) sm[jj] = smad24(hist[q - jj], coef[q], sm[jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        int sum = sm[jj];
#pragma unroll
                        for (int q = jj - 1; q >= 0; q--) sum = smad24(nv[jj - 1 - q], coef[q], sum);
                        const int v = res[jj] + (sum >> lshift);
                        badacc |= (unsigned)(v + hb);
                        nv[jj] = v;
                        out[jj] = (int)((unsigned)v << wasted);   // |v| < 2^23, wasted <= 6
                    }
                }
#pragma unroll
                for (int q = MAXO - 1; q >= 4; q--) hist[q] = hist[q - 4];
#pragma unroll
                for (int q = 0; q < 4 && q < MAXO; q++) hist[q] = nv[3 - q];
                // ---- the group's bytes
                int l[4], r[4];
                if (any2) {
                    const int a[4] = {(int)tp.x & amask, (int)tp.y & amask, (int)tp.z & amask, (int)tp.w & amask};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int X = dsw ? out[i] : a[i], Y = dsw ? a[i] : out[i];
                        r[i] = X - ((Y >> dshl) & dm);
                        l[i] = r[i] + Y;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) { l[i] = out[i]; r[i] = 0; }
                }
                if constexpr (O16) {
                    if (depth != 16) {
#pragma unroll
                        for (int i = 0; i < 4; i++) { l[i] = mode == 1 ? l[i] : wrap(l[i]); r[i] = wrap(r[i]); }
                    }
                    // at 16 bits the wrap IS the truncation to int16 (a value of [32768, 65535] keeps its low 16 bits); the reference wraps ONCE, so a
                    // value of [65536, 98304) becomes v - 65536 — what the truncation gives too; below -32768 or from 98304 on the result is no int16.
                    // (fmin / fmax see the values before that truncation: at 16 bits the bounds are -32768 and 98303 — publish() knows)
                    fmin = min(fmin, min(min(l[0], l[1]), min(l[2], l[3]))); fmax = max(fmax, max(max(l[0], l[1]), max(l[2], l[3])));
                    if (any2) { fmin = min(fmin, min(min(r[0], r[1]), min(r[2], r[3]))); fmax = max(fmax, max(max(r[0], r[1]), max(r[2], r[3]))); }
                    if (mode == 1) *reinterpret_cast<v4u *>(op) = v4u{(unsigned)out[0], (unsigned)out[1], (unsigned)out[2], (unsigned)out[3]};
                    else {
                        *reinterpret_cast<v2u *>(op) = v2u{__builtin_amdgcn_perm((unsigned)l[1], (unsigned)l[0], 0x05040100u), __builtin_amdgcn_perm((unsigned)l[3], (unsigned)l[2], 0x05040100u)};
                        if (any2) *reinterpret_cast<v2u *>(op + 16) = v2u{__builtin_amdgcn_perm((unsigned)r[1], (unsigned)r[0], 0x05040100u), __builtin_amdgcn_perm((unsigned)r[3], (unsigned)r[2], 0x05040100u)};
                    }
                } else {
                    if (mode == 1) *reinterpret_cast<v4u *>(op) = v4u{(unsigned)out[0], (unsigned)out[1], (unsigned)out[2], (unsigned)out[3]};
                    else {
                        *reinterpret_cast<v4u *>(op) = v4u{(unsigned)wrap(l[0]), (unsigned)wrap(l[1]), (unsigned)wrap(l[2]), (unsigned)wrap(l[3])};
                        if (any2) *reinterpret_cast<v4u *>(op + 16) = v4u{(unsigned)wrap(r[0]), (unsigned)wrap(r[1]), (unsigned)wrap(r[2]), (unsigned)wrap(r[3])};
                    }
                }
                op += ostep;
                groups--;
            }
            slow = !ok;
            go = ok && groups > 0 && (int)(whi32 - d) >= SNEED;
            if (any2) tp = *reinterpret_cast<const v4u *>(park_addr(j0 + 4 * (g0 - groups), go));   // the next group's parked values, a group ahead of their use
        }
        if (elig) {
            const int done = 4 * (g0 - groups);
            if (done > 0) { if (sv_lo < 0) sv_lo = jpos; sv_hi = jpos + done; }
            jpos += done; remaining -= done;
            b.pos = 32 * (dfull + (u64)(d - (unsigned)dfull) + 1) - (u64)s;
            // the round is over for a lane that reached its end, or that has groups left and not the window for them (the generic path must not
            // take those value by value)
            // (a lane whose run ended WITH the round goes on to the next partition's or subframe's header now: the round after, it runs with the others)
            if ((jpos >= rend && remaining > 0) || (!slow && remaining >= 4 && (int)(whi32 - d) < SNEED)) rdone = true;
        }
        return slow;
    };

    bool more = true;
    while (more) {
        SSTATW(st_rounds)
        // ---- the windows.  (1) the granules requested a round ago move into the ring; (2) a lane that begins a frame fetches its first granules
        // now; (3) every lane asks for the granules behind its ring's end that the ring has room for (what the NEXT round will read)
        {
            if (__any(pf_n > 0)) {
                // what the ring has room for NOW (a granule's slot is free once the dword the reader stands on has left it); the rest is asked for again
                const long long fit = ((long long)(((b.pos - 1) >> 5) + SWD) - (long long)b.whi) >> 2;
                const int nw = (int)max(0ll, min((long long)pf_n, fit));
                const unsigned k0 = (unsigned)(4 * pf_g0);
#pragma unroll
                for (int i = 0; i < SPF; i++) {
                    if (i < nw) {
                        const unsigned zm = ((pf_inr >> i) & 1u) ? 0xFFFFFFFFu : 0u;   // (a granule beyond the batch was read at a dummy address: zeros)
                        *reinterpret_cast<v4u *>(lw + ((k0 + 4u * i) & SRING)) = v4u{__builtin_bswap32(pf[i].x) & zm, __builtin_bswap32(pf[i].y) & zm, __builtin_bswap32(pf[i].z) & zm, __builtin_bswap32(pf[i].w) & zm};
                    }
                }
                if (nw > 0) { b.whi = 4 * (pf_g0 + (u64)nw); if (b.whi > b.wlo + SWD) b.wlo = b.whi - SWD; }
                pf_n = 0;
            }
            if (have_flush) flush();   // the rows of the round before: behind the wait above (loads and stores share vmcnt)
            if (__any(fresh)) {
                const u64 cg = b.pos >> 7;
                v4u t[SPF];
#pragma unroll
                for (int i = 0; i < SPF; i++) {
                    const bool inr = fresh && 2 * (cg + (u64)i) < A.G.safe_words;
                    t[i] = *reinterpret_cast<const v4u *>(A.G.w0 + (inr ? 2 * (cg + (u64)i) : 0ull));
                    if (!inr) t[i] = v4u{0, 0, 0, 0};
                }
                if (fresh) {
#pragma unroll
                    for (int i = 0; i < SPF; i++)
                        *reinterpret_cast<v4u *>(lw + ((unsigned)(4 * (cg + (u64)i)) & SRING)) = v4u{__builtin_bswap32(t[i].x), __builtin_bswap32(t[i].y), __builtin_bswap32(t[i].z), __builtin_bswap32(t[i].w)};
                    b.wlo = 4 * cg; b.whi = 4 * (cg + SPF);
                }
                fresh = false;
            }
            {
                const u64 g0 = b.whi >> 2;
                const int n = (have && st != S_DONE) ? SPF : 0;   // (what of it fits is decided when it has arrived: the round in between makes the room)
                pf_g0 = g0; pf_n = n;
                const bool allin = n > 0 && 2 * (g0 + SPF) <= A.G.safe_words;
                const v4u *src = reinterpret_cast<const v4u *>(A.G.w0) + (allin ? g0 : 0ull);   // (a lane with nothing to fetch reads the batch's first vectors)
                pf_inr = allin ? (1u << SPF) - 1u : 0u;
                if (A.G.safe_words >= 2 * SPF) {
#pragma unroll
                    for (int i = 0; i < SPF; i++) pf[i] = src[i];
                }
                if (__any(n > 0 && !allin)) {   // the batch's last granules: one by one, the ones beyond it not at all
#pragma unroll
                    for (int i = 0; i < SPF; i++) {
                        const bool inr = n > 0 && !allin && 2 * (g0 + (u64)i) < A.G.safe_words;
                        if (inr) { pf_inr |= 1u << i; pf[i] = *(reinterpret_cast<const v4u *>(A.G.w0) + g0 + (u64)i); }
                    }
                }
            }
        }

        // ---- a round: every lane advances its frame by up to RV values (to the next multiple of RV of its subframe: Rice partitions end where rounds end, the lanes of a wave stay in step), or until its window runs low
        rend = (jpos & ~(RV - 1)) + RV;
        rdone = st == S_DONE;
        // a lane whose window reaches the end of its stream's data reads value by value with every check (a stream's last rounds)
        const bool careful = b.end < (b.whi << 5) + 64;
        bool go_on = __any(!rdone);
        while (go_on) {
            SSTATW(st_outer)
            // -- groups of four values: Rice codes (:370-376) or fields of `rk` bits (:405, :423, :457)
            {
                const bool elig = !rdone && st == S_RUN && remaining >= 4 && (jpos & 3) == 0 && jpos >= order && !careful && jpos < rend &&
                                  !(mode == 2 && (bs & 3) != 0);   // (a stream's last frame: its parked values lie at any alignment — value by value)
                if (__any(elig)) {
                    const bool anywide = __any(elig && wide), big = __any(elig && order > 4);
                    bool slow;
                    if (anywide) slow = big ? groups_loop(std::integral_constant<int, 12>(), std::true_type(), elig) : groups_loop(std::integral_constant<int, 4>(), std::true_type(), elig);
                    else slow = big ? groups_loop(std::integral_constant<int, 12>(), std::false_type(), elig) : groups_loop(std::integral_constant<int, 4>(), std::false_type(), elig);
                    if (elig) slow_lane = slow;   // (a code longer than 32 bits: the generic reader's — this value and the ones that bring the run back to a multiple of four)
                }
            }
            // -- everything else.  A lane passes through as many of these steps as follow from each other in one turn (a run's end, the partition
            // header behind it; a subframe's end, the next one's header; a frame header and its first subframe's): the wave's other lanes wait for
            // every turn.  A step that reads wants SLOOK dwords of window in front of it, or the lane's round is over.
            auto window = [&]() -> bool { if ((b.pos >> 5) + SLOOK > b.whi) { rdone = true; return false; } return true; };
            // what reads nothing: a run that ended -> the next partition or the subframe's end; a subframe's end -> the next subframe or the frame's end
            if (!rdone && st == S_RUN && remaining == 0) { st = after; if (after == S_PART) { pi++; if (pi >= nparts) st = S_SUBEND; } }
            if (!rdone && st == S_SUBEND) {
                if (sv_lo >= 0) rdone = true;   // this subframe's last values leave (with this round) before the next header is read: a row has one destination
                else {
                    check_bad();
                    if (st != S_DONE) { ch++; jpos = 0; rend = RV; st = ch < nsub ? S_SUB : S_FRAMEEND; }
                    else rdone = true;
                }
            }
            if (!rdone && st == S_FRAMEEND) {   // :555-557
                    b.pos = (b.pos + 7) & ~7ull;                                  // alignToByte (frames start on byte boundaries of the batch buffer)
                    b.pos = (b.pos + 16 <= b.end) ? b.pos + 16 : b.end;           // readUint(16): a nil here is discarded, the NEXT readByte returns nil
                    end_byte = (b.pos - A.G.base_bit) >> 3;
                    st = S_DONE; rdone = true;
            }
            if (!rdone && st == S_FRAME && window()) {   // decodeFrame header  :510-553
                    int fs = FE_OK;
                    const unsigned t0 = srd_get(b, 8);
                    if (b.eof) fs = FE_EOF_START;
                    const unsigned sync = t0 * 64 + srd_get(b, 6);
                    if (!fs && b.eof) fs = FE_NIL;
                    if (!fs && sync != 0x3FFE) fs = FE_SYNC;
                    srd_get(b, 2);
                    const int bsc = (int)srd_get(b, 4), src_code = (int)srd_get(b, 4);
                    chan_asgn = (int)srd_get(b, 4);
                    srd_get(b, 4);
                    const int t = (int)srd_get(b, 8);
                    if (!fs && b.eof) fs = FE_NIL;
                    int t2 = -1;
                    for (int i = 7; i >= 0; i--) { if (!(t & (1 << i))) break; t2++; }
                    for (int i = 1; i <= t2; i++) srd_get(b, 8);
                    if (bsc == 1) bs = 192;
                    else if (bsc >= 2 && bsc <= 5) bs = 576 << (bsc - 2);
                    else if (bsc == 6) bs = (int)srd_get(b, 8) + 1;
                    else if (bsc == 7) bs = (int)srd_get(b, 16) + 1;
                    else if (bsc >= 8) bs = 256 << (bsc - 8);
                    else { bs = 0; if (!fs) fs = FE_BLOCKSIZE; }
                    if (src_code == 12) srd_get(b, 8);
                    else if (src_code == 13 || src_code == 14) srd_get(b, 16);
                    srd_get(b, 8);   // CRC-8, ignored :553
                    if (!fs && b.eof) fs = FE_NIL;
                    if (!fs) {
                        if (chan_asgn <= 7) nsub = C;
                        else if (chan_asgn <= 10) { nsub = 2; if (C != 2) fs = FE_NIL; }   // result[ch] of a missing / extra channel is nil (:482-507)
                        else fs = FE_CHAN;
                    }
                    status = fs;
                    if (fs != FE_OK) { st = S_DONE; rdone = true; }
                    else {
                        limit = (A.limit_factor > 0 && !nolimit) ? b.pos + (u64)A.limit_factor * (u64)bs * (u64)C * (u64)(depth + 2) / 4 + 4096 : ~0ull;
                        const u64 need = (u64)nsub * (u64)bs;
                        cand_scratch = atomicAdd(A.scratch_cursor, ((need + 3) & ~3ull) + 32);   // (+ 32: see k_flac_extract — frames must not all start at one offset within 16 KiB)
                        store_ok = cand_scratch + need <= A.scratch_cap;
                        ch = 0; jpos = 0;
                        st = S_SUB;
                    }
            }
            if (!rdone && st == S_SUB && window()) {   // decodeSubframe  :443-465
                    srd_get(b, 1);
                    const int type = (int)srd_get(b, 6);
                    wasted = (int)srd_get(b, 1);
                    if (wasted == 1) {   // unary wasted-bits count  :447-449
                        bool gw = true;
                        while (gw) { const unsigned bit = srd_get(b, 1); if (b.eof || bit) gw = false; else wasted++; }
                    }
                    sdepth = depth - wasted;
                    if (chan_asgn >= 8) sdepth += ((chan_asgn == 9) == (ch == 0)) ? 1 : 0;   // the side channel has one more bit  :480-481
                    order = 0; lshift = 0; jpos = 0; rend = RV; lpc = false; hb = 1 << 23;
                    wide = sdepth > 24 || wasted > 6;   // (until S_COEF knows the taps: the bound the warm-up samples are held to)
#pragma unroll
                    for (int q = 0; q < SMAXO; q++) { coef[q] = 0; hist[q] = 0; }
                    set_outputs();
                    if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                    else if (sdepth < 1 || sdepth > 31 || wasted > 24) { status = FE_DECLINE; st = S_DONE; rdone = true; }
                    else if (type == 0) {
                        cval = srd_sget(b, sdepth);
                        if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                        else if (sdepth > 24 || wasted > 6) { status = FE_DECLINE; st = S_DONE; rdone = true; }   // (a constant beyond the ranges of this kernel: not an ordinary stream)
                        else { remaining = bs; st = S_CONST; }
                    } else if (type == 1) { remaining = bs; fixed = true; rk = sdepth; after = S_SUBEND; st = S_RUN; }
                    else if ((type >= 8 && type <= 12) || (type >= 32 && type <= 63)) {
                        order = type <= 12 ? type - 8 : type - 31;
                        lpc = type >= 32;
                        if (order > SMAXO || order > bs) { status = FE_DECLINE; st = S_DONE; rdone = true; }   // (order > bs: the Lua table grows past blockSize)
                        else { remaining = order; fixed = true; rk = sdepth; after = S_COEF; st = S_RUN; }
                    } else { status = FE_SUBTYPE; st = S_DONE; rdone = true; }
            }
            if (!rdone && st == S_COEF && window()) {   // :433-438 / FIXED_PREDICTION_COEFFICIENTS :334-340, then the residual header :381-391
                    if (lpc) {
                        const int precision = (int)srd_get(b, 4) + 1;
                        lshift = srd_sget(b, 5);
#pragma unroll
                        for (int q = 0; q < SMAXO; q++) if (q < order) coef[q] = srd_sget(b, precision);
                    } else {
                        // FIXED_PREDICTION_COEFFICIENTS[order + 1] = {}, {1}, {2, -1}, {3, -3, 1}, {4, -6, 4, -1}: binomials, by arithmetic
                        coef[0] = order;
                        coef[1] = order == 2 ? -1 : (order == 3 ? -3 : (order == 4 ? -6 : 0));
                        coef[2] = order == 3 ? 1 : (order == 4 ? 4 : 0);
                        coef[3] = order == 4 ? -1 : 0;
                    }
                    const int method = (int)srd_get(b, 2);
                    param_bits = method == 0 ? 4 : 5;
                    const int porder = (int)srd_get(b, 4);
                    nparts = 1 << porder;
                    int sabs = 1;
#pragma unroll
                    for (int q = 0; q < SMAXO; q++) sabs += coef[q] < 0 ? -coef[q] : coef[q];
                    const int hbits = min(23, __builtin_clz((unsigned)sabs) - 1);   // 2^hbits * sum |coef| < 2^31
                    hb = 1 << hbits;
                    // values of this subframe have up to sdepth bits: the 24-bit multiply-adds serve it when those fit under hb; else 64-bit sums
                    wide = sdepth - 1 > hbits || lshift < 0 || wasted > 6;   // (the warm-up samples, sdepth-bit fields, fit under hb whenever this says narrow)
                    if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                    else if (method >= 2) { status = FE_RESMETHOD; st = S_DONE; rdone = true; }
                    else if (bs % nparts != 0) { status = FE_PARTITION; st = S_DONE; rdone = true; }
                    else {
                        psize = bs / nparts;
                        pi = 0;
                        if (nparts > 1 && psize < order) { status = FE_DECLINE; st = S_DONE; rdone = true; }   // :400 — later partitions overwrite warm-up entries
                        else st = S_PART;
                    }
            }
            if (!rdone && st == S_PART && window()) {   // :394-406
                    const int escape = param_bits == 4 ? 15 : 31;
                    const int param = (int)srd_get(b, param_bits);
                    const bool esc = param >= escape;
                    int nbits = 0;
                    if (esc) nbits = (int)srd_get(b, 5);
                    if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                    else if (b.pos > limit) { status = FE_LIMIT; st = S_DONE; rdone = true; }
                    else {
                        const int start_i = pi * psize + (pi == 0 ? order : 0), endd = (pi + 1) * psize;
                        remaining = endd > start_i ? endd - start_i : 0;
                        fixed = esc;
                        rk = esc ? nbits : param;
                        after = S_PART;
                        st = S_RUN;   // (an empty partition leaves through S_RUN's remaining == 0)
                    }
            }
            if (!rdone && st == S_CONST) {   // :453-454
                    const int o = (int)((unsigned)cval << wasted);
                    while (remaining > 0 && jpos < rend) { emit1(o); remaining--; jpos++; }
                    if (remaining == 0) st = S_SUBEND; else rdone = true;
            }
            if (!rdone && st == S_RUN && remaining > 0) {
                if (jpos >= rend) rdone = true;
                else if ((slow_lane || !(remaining >= 4 && (jpos & 3) == 0 && jpos >= order && !careful && !(mode == 2 && (bs & 3) != 0))) && window()) {   // (what the groups take is theirs: next turn)
                        slow_lane = false;
                        // one value by the generic reader: warm-up samples, the values that bring a run to a multiple of four, a code longer than 32 bits,
                        // a stream's last bytes; 64-bit sums (:411-419)
                        int r1 = FE_OK, v1 = 0;
                        if (fixed) { v1 = srd_sget(b, rk); if (b.eof) r1 = FE_NIL; }
                        else r1 = srd_rice(b, rk, v1);
                        if (!r1 && b.oow) r1 = FE_DECLINE;
                        if (r1) { status = r1; st = S_DONE; rdone = true; }
                        else {
                            long long sum = 0;
#pragma unroll
                            for (int q = 0; q < SMAXO; q++) sum += (long long)hist[q] * (long long)coef[q];
                            long long pr = lshift >= 0 ? (sum >> lshift) : (sum << (-lshift));
                            if (jpos < order) pr = 0;
                            const long long vv = (long long)v1 + pr;
                            const long long oo = vv << wasted;
                            if (wide) { if ((unsigned long long)(oo + (1ll << 29)) >= (1ull << 30) || (unsigned long long)(vv + (1ll << 29)) >= (1ull << 30)) badacc |= 0x80000000u; }
                            else { if ((unsigned long long)(vv + (long long)hb) >= 2ull * (unsigned long long)hb) badacc |= 0x80000000u; }
#pragma unroll
                            for (int q = SMAXO - 1; q >= 1; q--) hist[q] = hist[q - 1];
                            hist[0] = (int)vv;
                            emit1((int)oo);
                            SSTAT(st_singles++;) SSTATW(st_single_turns)
                            remaining--; jpos++;
                        }
                }
            }
            if (b.oow && st != S_DONE) { status = FE_DECLINE; st = S_DONE; rdone = true; }   // a field beyond the window: not an ordinary stream
            go_on = __any(!rdone);
        }
        if (st != S_DONE && b.pos > limit) { status = FE_LIMIT; st = S_DONE; }
        if (st != S_DONE) check_bad();
        publish();
        have_flush = true;
        if (have && st == S_DONE) finish();
        if (__ballot(have) == 0) take();   // new frames when the whole wave is through with its old ones: the lanes then parse their headers in the same rounds
        more = __ballot(st != S_DONE) != 0;
    }
    if (have_flush) flush();
    if constexpr (O16) { if (__any(bad16 != 0u) && lane == 0) atomicOr(A.flags, 0x100u); }
#ifdef AUKIT_FLAC_STATS
    { u64 *cs[6] = {&st_rounds, &st_outer, &st_turns, &st_singles, &st_lane_turns, &st_single_turns};
      for (int c = 0; c < 6; c++) { u64 sv = *cs[c]; for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o); *cs[c] = sv; } }
    if (lane == 0) { atomicAdd(A.stats + 0, st_rounds); atomicAdd(A.stats + 1, st_outer); atomicAdd(A.stats + 2, st_turns); atomicAdd(A.stats + 3, st_singles); atomicAdd(A.stats + 4, st_lane_turns); atomicAdd(A.stats + 5, st_single_turns); }
#endif
}

int flac_stream_launch(aukit_ctx *ctx, const FusedArgs &A) {
    if (!A.count) return AUKIT_OK;
    AUKIT_HIP_CHECK(hipMemsetAsync(A.ticket, 0, 4, ctx->stream));
    static const int wgs = getenv("AUKIT_FLAC_STREAM_WGS") ? atoi(getenv("AUKIT_FLAC_STREAM_WGS")) : 4 * AUKIT_FS_LB;   // waves per CU (three per SIMD)
    const unsigned grid = std::min<unsigned>((A.count + 63) / 64, (unsigned)ctx->num_cus * (unsigned)std::max(wgs, 1));
    if (A.out16) hipLaunchKernelGGL((k_flac_stream<true>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else hipLaunchKernelGGL((k_flac_stream<false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

}  // namespace aukit
