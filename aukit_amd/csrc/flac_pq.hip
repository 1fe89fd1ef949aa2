// flac_pq.hip — k_flac_pq: a PARSER wave and a PREDICTOR wave per 64 frames (round 6; VERDICT r05 item 1c: small batches).
//
// k_flac_stream (flac_stream.hip) walks a frame with one lane: what a batch costs that does not fill the chip — 256 streams are 432 waves on
// 1024 SIMDs — is that lane's chain: 8192 values x (parse + predict + emit) ≈ 82 instructions each.  Here the chain is cut in two:
//   wave P (threads 0-63): the lane-filled bit window, the state machine of decodeFrame (aukit.lua:510-567), the Rice codes and fields in groups
//     of four — and nothing else: a round's values (RESIDUALS, warm-up samples, constants, as read) go to the lane's row in LDS, a subframe's
//     parameters (order, shift, coefficients, where its values go) to the lane's parameter block;
//   wave Q (threads 64-127): lane l restores the prediction (:411-419) over lane l's row of the round BEFORE, shifts (:467-469), decorrelates
//     (:482-497), wraps (:501-507), packs, and flushes the output rows exactly as k_flac_stream does.
// Two buffers (rows, parameter blocks, counts), one workgroup barrier per round: P fills buffer r while Q works on buffer r - 1.  What Q finds
// beyond its ranges (the multiply-add bound of a subframe, an int16 final) it says in s_bad / its own flag; a frame is reported two rounds behind
// its last value, when Q has seen everything of it.  Same arguments, same outputs as k_flac_stream and k_flac_decode; the host takes this
// kernel when the batch has fewer frames than three workgroups per CU hold at once (flac_pq_launch: its 53 KB of LDS per workgroup and the
// second wave cost a full chip 1.5 x what they win a small batch).
#include "flac_stream_dev.h"

namespace aukit {

namespace {
constexpr int PRS = 36;    // dwords between two lanes' value rows (32 values + 4: 16-byte aligned)
constexpr int PPW = 28;    // dwords of a lane's parameter block
// the parameter block: [0, 12) coefficients, 12 lshift, 13 hb, 14 wide  (part B: S_COEF) | 16 order, 17 wasted, 18 mode | dswap << 2 | store_ok << 3,
// 19 dsh, 20 dmask, 21 bs, 22-23 o_fin, 24-25 o_park, 26 wide0 (the bound of the warm-up samples) (part A: S_SUB)
enum { PF_A = 1, PF_B = 2, PF_NEWSUB = 4 };
}  // namespace

template <bool O16>
__global__ __launch_bounds__(128) void k_flac_pq(const FusedArgs A) {
    constexpr int RV = O16 ? 32 : 16;
    __shared__ __attribute__((aligned(16))) unsigned s_win[64 * SWS];
    __shared__ __attribute__((aligned(16))) unsigned s_res[2][64 * PRS];
    __shared__ __attribute__((aligned(16))) unsigned s_par[2][64 * PPW];
    __shared__ unsigned s_meta[2][64];   // values of the round | index of the first << 8 | flags << 24
    __shared__ unsigned s_bad[64];       // Q -> P: the frame in this lane is beyond this kernel's ranges (FE_DECLINE)
    __shared__ unsigned s_more[2];       // P -> Q: another iteration (by the iteration's parity: P writes the slot again two barriers later)
    __shared__ __attribute__((aligned(16))) unsigned s_out[64 * SOS];
    __shared__ u64 s_dst[64 * 2];
    __shared__ unsigned s_rng[64];
    __shared__ unsigned s_end[64];
    const int lane = threadIdx.x & 63;
    const bool isQ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0;
    const int C = A.C, depth = A.depth;
    const int wrap_half = 1 << (depth - 1), wrap_full = 1 << depth;
    auto wrap = [&](int v) -> int { return v >= wrap_half ? v - wrap_full : v; };   // :504
    if (threadIdx.x < 64) { s_bad[lane] = 0; s_meta[0][lane] = 0; s_meta[1][lane] = 0; }
    if (threadIdx.x == 0) { s_more[0] = 1; s_more[1] = 1; }
    __syncthreads();

    if (!isQ) {
        // =========================================================================================== wave P: bits -> values
        unsigned *const lw = s_win + lane * SWS;
        SRd b;
        b.lw = lw;
        b.wlo = 0; b.whi = 0; b.pos = 0; b.end = 0; b.eof = 0; b.oow = 0;
        u64 limit = ~0ull;
        bool have = false, fresh = false;
        unsigned idx = 0, nolimit = 0;
        int st = S_DONE, status = FE_OK, drain = 0;
        int bs = 0, chan_asgn = 0, nsub = 0, ch = 0;
        int order = 0, wasted = 0, sdepth = 0, after = S_SUBEND;
        int nparts = 0, psize = 0, pi = 0, param_bits = 4, remaining = 0, jpos = 0, rk = 0, cval = 0;
        bool fixed = false, store_ok = false, lpc = false;
        u64 cand_scratch = 0, end_byte = 0;
        unsigned pflags = 0;
        int sv_lo = -1, sv_hi = 0;
        v4u pf[SPF];
        u64 pf_g0 = 0;
        int pf_n = 0;
        unsigned pf_inr = 0;
#pragma unroll
        for (int i = 0; i < SPF; i++) pf[i] = v4u{0, 0, 0, 0};
        constexpr int S_DRAIN = S_DONE + 1;   // the frame's values are all read; Q is two rounds from having seen them

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
            s_bad[lane] = 0;
        };
        auto finish = [&]() {
            CandInfo f;
            f.end_byte = end_byte; f.scratch = cand_scratch; f.sample_off = 0;
            f.blocksize = bs; f.chan_asgn = chan_asgn; f.status = status; f.nsub = nsub;
            f.seq = 0; f.used = 0;
            A.ci[idx] = f;
            have = false;
        };
        auto take = [&]() {
            const u64 m = __ballot(!have);
            if (!m) return;
            unsigned base = 0;
            if (lane == __builtin_ctzll(m)) base = atomicAdd(A.ticket, (unsigned)__builtin_popcountll(m));
            base = __shfl(base, __builtin_ctzll(m));
            const unsigned rel = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1));
            if (!have && rel < A.count) start(rel);
        };
        take();

        int rend = 0;
        bool slow_lane = false, rdone = false;
        unsigned it = 0;
        bool more = true;
        while (more) {
            const unsigned bufi = it & 1u;
            unsigned *const rrow = s_res[bufi] + lane * PRS;
            unsigned *const par = s_par[bufi] + lane * PPW;
            auto put = [&](int v) { rrow[jpos & (RV - 1)] = (unsigned)v; if (sv_lo < 0) sv_lo = jpos; sv_hi = jpos + 1; };
            // ---- the windows (k_flac_stream's: the granules requested a round ago move in, a lane that begins a frame fetches now, every lane asks on)
            {
                if (__any(pf_n > 0)) {
                    const long long fit = ((long long)(((b.pos - 1) >> 5) + SWD) - (long long)b.whi) >> 2;
                    const int nw = (int)max(0ll, min((long long)pf_n, fit));
                    const unsigned k0 = (unsigned)(4 * pf_g0);
#pragma unroll
                    for (int i = 0; i < SPF; i++) {
                        if (i < nw) {
                            const unsigned zm = ((pf_inr >> i) & 1u) ? 0xFFFFFFFFu : 0u;
                            *reinterpret_cast<v4u *>(lw + ((k0 + 4u * i) & SRING)) = v4u{__builtin_bswap32(pf[i].x) & zm, __builtin_bswap32(pf[i].y) & zm, __builtin_bswap32(pf[i].z) & zm, __builtin_bswap32(pf[i].w) & zm};
                        }
                    }
                    if (nw > 0) { b.whi = 4 * (pf_g0 + (u64)nw); if (b.whi > b.wlo + SWD) b.wlo = b.whi - SWD; }
                    pf_n = 0;
                }
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
                    const int n = (have && st < S_DONE) ? SPF : 0;
                    pf_g0 = g0; pf_n = n;
                    const bool allin = n > 0 && 2 * (g0 + SPF) <= A.G.safe_words;
                    const v4u *src = reinterpret_cast<const v4u *>(A.G.w0) + (allin ? g0 : 0ull);
                    pf_inr = allin ? (1u << SPF) - 1u : 0u;
                    if (A.G.safe_words >= 2 * SPF) {
#pragma unroll
                        for (int i = 0; i < SPF; i++) pf[i] = src[i];
                    }
                    if (__any(n > 0 && !allin)) {
#pragma unroll
                        for (int i = 0; i < SPF; i++) {
                            const bool inr = n > 0 && !allin && 2 * (g0 + (u64)i) < A.G.safe_words;
                            if (inr) { pf_inr |= 1u << i; pf[i] = *(reinterpret_cast<const v4u *>(A.G.w0) + g0 + (u64)i); }
                        }
                    }
                }
            }

            // ---- a round
            rend = (jpos & ~(RV - 1)) + RV;
            if (st == S_DRAIN) {   // Q has seen the frame's last values when this has counted two rounds down
                if (--drain <= 0) { if (s_bad[lane]) status = FE_DECLINE; st = S_DONE; }
            }
            rdone = st >= S_DONE;
            const bool careful = b.end < (b.whi << 5) + 64;
            bool go_on = __any(!rdone);
            while (go_on) {
                // -- groups of four values: Rice codes (:370-376) or fields of `rk` bits (:405, :423, :457), as read
                {
                    const bool elig = !rdone && st == S_RUN && remaining >= 4 && (jpos & 3) == 0 && jpos >= order && !careful && jpos < rend;
                    if (__any(elig)) {
                        const u64 dfull = (b.pos - 1) >> 5;
                        unsigned d = (unsigned)dfull;
                        int s = (int)((0u - (unsigned)b.pos) & 31u);
                        unsigned w0 = lw[d & SRING], w1 = lw[(d + 1) & SRING];
                        const unsigned whi32 = (unsigned)b.whi;
                        int groups = min(remaining, rend - jpos) >> 2;
                        bool slow = false;
                        bool go = elig && groups > 0 && (int)(whi32 - d) >= SNEED;
                        const int zmask = fixed ? 0 : -1, rk1 = fixed ? rk : rk + 1;
                        const unsigned krm = 31u - (unsigned)rk, fo = 32u - (unsigned)rk;
                        const int g0 = groups;
                        unsigned *op = rrow + (jpos & (RV - 1));
                        while (go) {
                            unsigned d_ = d, w0_ = w0, w1_ = w1;
                            int s_ = s;
                            int totmax = 0;
                            int res[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const unsigned wn = lw[(d_ + 2) & SRING];
                                const unsigned hi = __builtin_amdgcn_alignbit(w0_, w1_, (unsigned)s_);
                                const int z = hi ? __builtin_clz(hi) : 32;
                                const unsigned low = __builtin_amdgcn_ubfe(hi, krm - (unsigned)z, (unsigned)rk);
                                const unsigned ur = ((unsigned)z << rk) | low;
                                const int v_r = (int)(ur >> 1) ^ -(int)(ur & 1u);
                                const int v_f = __builtin_amdgcn_sbfe((int)hi, fo, (unsigned)rk);
                                const int tot = (z & zmask) + rk1;
                                totmax = max(totmax, tot);
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
                                *reinterpret_cast<v4u *>(op) = v4u{(unsigned)res[0], (unsigned)res[1], (unsigned)res[2], (unsigned)res[3]};
                                op += 4;
                                groups--;
                            }
                            slow = !ok;
                            go = ok && groups > 0 && (int)(whi32 - d) >= SNEED;
                        }
                        if (elig) {
                            const int done = 4 * (g0 - groups);
                            if (done > 0) { if (sv_lo < 0) sv_lo = jpos; sv_hi = jpos + done; }
                            jpos += done; remaining -= done;
                            b.pos = 32 * (dfull + (u64)(d - (unsigned)dfull) + 1) - (u64)s;
                            if ((jpos >= rend && remaining > 0) || (!slow && remaining >= 4 && (int)(whi32 - d) < SNEED)) rdone = true;
                            slow_lane = slow;
                        }
                    }
                }
                // -- everything else (k_flac_stream's steps; a value is only READ here)
                auto window = [&]() -> bool { if ((b.pos >> 5) + SLOOK > b.whi) { rdone = true; return false; } return true; };
                if (!rdone && st == S_RUN && remaining == 0) { st = after; if (after == S_PART) { pi++; if (pi >= nparts) st = S_SUBEND; } }
                if (!rdone && st == S_SUBEND) {
                    if (sv_lo >= 0) rdone = true;   // a round's values are ONE subframe's (its parameters travel with the round)
                    else { ch++; jpos = 0; rend = RV; st = ch < nsub ? S_SUB : S_FRAMEEND; }
                }
                if (!rdone && st == S_FRAMEEND) {   // :555-557
                    b.pos = (b.pos + 7) & ~7ull;
                    b.pos = (b.pos + 16 <= b.end) ? b.pos + 16 : b.end;
                    end_byte = (b.pos - A.G.base_bit) >> 3;
                    st = S_DRAIN; drain = 2; rdone = true;
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
                        else if (chan_asgn <= 10) { nsub = 2; if (C != 2) fs = FE_NIL; }
                        else fs = FE_CHAN;
                    }
                    status = fs;
                    if (fs != FE_OK) { st = S_DONE; rdone = true; }
                    else {
                        limit = (A.limit_factor > 0 && !nolimit) ? b.pos + (u64)A.limit_factor * (u64)bs * (u64)C * (u64)(depth + 2) / 4 + 4096 : ~0ull;
                        const u64 need = (u64)nsub * (u64)bs;
                        cand_scratch = atomicAdd(A.scratch_cursor, ((need + 3) & ~3ull) + 32);
                        store_ok = cand_scratch + need <= A.scratch_cap;
                        ch = 0; jpos = 0;
                        st = S_SUB;
                    }
                }
                if (!rdone && st == S_SUB && window()) {   // decodeSubframe  :443-465
                    srd_get(b, 1);
                    const int type = (int)srd_get(b, 6);
                    wasted = (int)srd_get(b, 1);
                    if (wasted == 1) {
                        bool gw = true;
                        while (gw) { const unsigned bit = srd_get(b, 1); if (b.eof || bit) gw = false; else wasted++; }
                    }
                    sdepth = depth - wasted;
                    if (chan_asgn >= 8) sdepth += ((chan_asgn == 9) == (ch == 0)) ? 1 : 0;   // :480-481
                    order = 0; jpos = 0; rend = RV; lpc = false;
                    {   // part A of the parameters: where this subframe's values go and how
                        const bool decor = C == 2 && chan_asgn >= 8 && chan_asgn <= 10;
                        const int mode = !decor ? 0 : (ch == 0 ? 1 : 2);
                        u64 o_fin, o_park;
                        if constexpr (O16) { o_fin = 2 * cand_scratch + (mode == 0 ? (u64)ch * (u64)bs : 0ull); o_park = cand_scratch + (u64)bs; }
                        else { o_fin = cand_scratch + (mode == 0 ? (u64)ch * (u64)bs : 0ull); o_park = cand_scratch + (u64)bs; }
                        par[17] = (unsigned)wasted;
                        par[18] = (unsigned)mode | (chan_asgn == 9 ? 4u : 0u) | (store_ok ? 8u : 0u);
                        par[19] = chan_asgn == 10 ? 1u : 0u;
                        par[20] = chan_asgn == 9 ? 0u : 0xFFFFFFFFu;
                        par[21] = (unsigned)bs;
                        par[22] = (unsigned)o_fin; par[23] = (unsigned)(o_fin >> 32);
                        par[24] = (unsigned)o_park; par[25] = (unsigned)(o_park >> 32);
                        par[26] = (sdepth > 24 || wasted > 6) ? 1u : 0u;
                        pflags |= PF_A | PF_NEWSUB;
                    }
                    if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                    else if (sdepth < 1 || sdepth > 31 || wasted > 24) { status = FE_DECLINE; st = S_DONE; rdone = true; }
                    else if (type == 0) {
                        cval = srd_sget(b, sdepth);
                        if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                        else if (sdepth > 24 || wasted > 6) { status = FE_DECLINE; st = S_DONE; rdone = true; }
                        else { remaining = bs; st = S_CONST; }
                    } else if (type == 1) { remaining = bs; fixed = true; rk = sdepth; after = S_SUBEND; st = S_RUN; }
                    else if ((type >= 8 && type <= 12) || (type >= 32 && type <= 63)) {
                        order = type <= 12 ? type - 8 : type - 31;
                        lpc = type >= 32;
                        if (order > SMAXO || order > bs) { status = FE_DECLINE; st = S_DONE; rdone = true; }
                        else { remaining = order; fixed = true; rk = sdepth; after = S_COEF; st = S_RUN; }
                    } else { status = FE_SUBTYPE; st = S_DONE; rdone = true; }
                    par[16] = (unsigned)order;
                }
                if (!rdone && st == S_COEF && window()) {   // :433-438 / :334-340, then the residual header :381-391
                    int coef[SMAXO];
#pragma unroll
                    for (int q = 0; q < SMAXO; q++) coef[q] = 0;
                    int lshift = 0;
                    if (lpc) {
                        const int precision = (int)srd_get(b, 4) + 1;
                        lshift = srd_sget(b, 5);
#pragma unroll
                        for (int q = 0; q < SMAXO; q++) if (q < order) coef[q] = srd_sget(b, precision);
                    } else {
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
                    const int hbits = min(23, __builtin_clz((unsigned)sabs) - 1);
#pragma unroll
                    for (int q = 0; q < SMAXO; q++) par[q] = (unsigned)coef[q];
                    par[12] = (unsigned)lshift;
                    par[13] = 1u << hbits;
                    par[14] = (sdepth - 1 > hbits || lshift < 0 || wasted > 6) ? 1u : 0u;
                    pflags |= PF_B;
                    if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                    else if (method >= 2) { status = FE_RESMETHOD; st = S_DONE; rdone = true; }
                    else if (bs % nparts != 0) { status = FE_PARTITION; st = S_DONE; rdone = true; }
                    else {
                        psize = bs / nparts;
                        pi = 0;
                        if (nparts > 1 && psize < order) { status = FE_DECLINE; st = S_DONE; rdone = true; }   // :400
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
                        st = S_RUN;
                    }
                }
                if (!rdone && st == S_CONST) {   // :453-454
                    while (remaining > 0 && jpos < rend) { put(cval); remaining--; jpos++; }
                    if (remaining == 0) st = S_SUBEND; else rdone = true;
                }
                if (!rdone && st == S_RUN && remaining > 0) {
                    if (jpos >= rend) rdone = true;
                    else if ((slow_lane || !(remaining >= 4 && (jpos & 3) == 0 && jpos >= order && !careful)) && window()) {
                        slow_lane = false;
                        int r1 = FE_OK, v1 = 0;
                        if (fixed) { v1 = srd_sget(b, rk); if (b.eof) r1 = FE_NIL; }
                        else r1 = srd_rice(b, rk, v1);
                        if (!r1 && b.oow) r1 = FE_DECLINE;
                        if (r1) { status = r1; st = S_DONE; rdone = true; }
                        else { put(v1); remaining--; jpos++; }
                    }
                }
                if (b.oow && st < S_DONE) { status = FE_DECLINE; st = S_DONE; rdone = true; }
                go_on = __any(!rdone);
            }
            if (st < S_DONE && b.pos > limit) { status = FE_LIMIT; st = S_DONE; }
            // what the round read, for Q
            s_meta[bufi][lane] = (sv_lo >= 0 ? (unsigned)(sv_hi - sv_lo) | ((unsigned)(sv_lo & 0xFFFF) << 8) : 0u) | (pflags << 24);
            sv_lo = -1; pflags = 0;
            if (have && st == S_DONE) finish();
            if (__ballot(have) == 0) take();
            more = __ballot(st != S_DONE) != 0;
            if (lane == 0) s_more[bufi] = more ? 1u : 0u;
            __syncthreads();
            it++;
        }
        return;
    }

    // =============================================================================================== wave Q: values -> finals
    int order = 0, lshift = 0, wasted = 0, hb = 1 << 23, mode = 0, dsh = 0, dmask = 0, bs = 0;
    bool wide = false, dswap = false, store_ok = false;
    u64 o_fin = 0, o_park = 0;
    unsigned badacc = 0, bad16 = 0;
    int coef[SMAXO], hist[SMAXO];
#pragma unroll
    for (int q = 0; q < SMAXO; q++) { coef[q] = 0; hist[q] = 0; }
    int fmin = 0, fmax = 0;
    int sv_lo = -1, sv_hi = 0, jpos = 0;
    unsigned *const orow = s_out + lane * SOS;
    auto staged = [&](int n) { if (sv_lo < 0) sv_lo = jpos; sv_hi = jpos + n; };
    auto emit1 = [&](int o) {
        const int jr = jpos & (RV - 1);
        if constexpr (O16) {
            short *hrow = reinterpret_cast<short *>(orow);
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
    auto park_addr = [&](int j, bool on) -> const int * {
        const u64 e = O16 ? o_park + (u64)j : o_fin + (u64)bs + (u64)j;
        return A.scratch + ((on && mode == 2 && store_ok && (bs & 3) == 0) ? e : 0ull);
    };
    auto publish = [&]() {
        unsigned m = 0;
        u64 dA = 0, dB = 0;
        if (sv_lo >= 0 && store_ok) {
            const int R0 = sv_lo & ~(RV - 1);
            const int lo = sv_lo - R0, hi = sv_hi - R0;
            int la, ha, lb, hb2;
            if (O16 && mode == 1) {
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
            auto pieces = [](int l, int h, u64 d) -> unsigned {
                if (h <= l || (d & 7)) return 0u;
                return ((1u << (h >> 3)) - 1u) & ~((1u << ((l + 7) >> 3)) - 1u);
            };
            const unsigned pa = pieces(la, ha, dA), pb = pieces(lb, hb2, dB);
            const bool rag = (ha > la && (((la | ha) & 7) || (dA & 7))) || (hb2 > lb && (((lb | hb2) & 7) || (dB & 7)));
            m = pa | (pb << 8) | ((dA & 15) == 0 ? 1u << 16 : 0u) | ((dB & 15) == 0 ? 1u << 17 : 0u) | (rag ? 1u << 18 : 0u) | ((unsigned)la << 19) | ((unsigned)lb << 26);
            if constexpr (O16) { if (mode != 1 && (fmin < -32768 || fmax > (depth == 16 ? 98303 : 32767))) bad16 = 1; }
        }
        fmin = 0; fmax = 0;
        s_dst[2 * lane] = dA; s_dst[2 * lane + 1] = dB;
        s_rng[lane] = m;
        s_end[lane] = (unsigned)(sv_lo >= 0 ? (sv_hi - (sv_lo & ~(RV - 1))) : 0) | ((unsigned)mode << 8);
        sv_lo = -1;
    };
    auto flush = [&]() {
        __builtin_amdgcn_wave_barrier();
        const unsigned own = s_rng[lane];
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
                    else {
                        if (bits & 1u) sstore(*reinterpret_cast<const v2u *>(src), reinterpret_cast<v2u *>(g));
                        if (bits & 2u) sstore(*reinterpret_cast<const v2u *>(src + 2), reinterpret_cast<v2u *>(g + 8));
                    }
                }
            }
            if (__any((own >> 18) & 1u)) {
                if ((own >> 18) & 1u) {
                    const u64 dA = s_dst[2 * lane], dB = s_dst[2 * lane + 1];
                    const unsigned e = s_end[lane];
                    const int n = (int)(e & 0xFFu), md = (int)(e >> 8);
                    for (int h = 0; h < 2; h++) {
                        const u64 dst = h ? dB : dA;
                        const unsigned pm = (own >> (8 * h)) & 0xFFu;
                        const int lo = (int)((own >> (h ? 26 : 19)) & 0x7Fu);
                        int hi;
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

    // the groups of four of Q: k_flac_stream's prediction and bytes on values that lie in the row
    auto groups_q = [&](auto MO, auto WD, bool elig, const unsigned *rrow, int &left) {
        constexpr int MAXO = decltype(MO)::value;
        constexpr bool WIDE = decltype(WD)::value;
        int groups = left >> 2;
        bool go = elig && groups > 0;
        const bool any2 = __any(elig && mode == 2);
        const int amask = mode == 2 ? -1 : 0, dm = mode == 2 ? dmask : 0, dshl = mode == 2 ? dsh : 0;
        const bool dsw = mode == 2 && dswap;
        const int g0 = groups;
        unsigned *op = orow + (mode == 1 ? (jpos & (RV - 1)) : ((O16 ? (jpos & (RV - 1)) >> 1 : (jpos & (RV - 1)))));
        const int ostep = (O16 && mode != 1) ? 2 : 4;
        const unsigned *ip = rrow + (jpos & (RV - 1));
        const int j0 = jpos;
        v4u tp = v4u{0, 0, 0, 0};
        if (any2) tp = *reinterpret_cast<const v4u *>(park_addr(jpos, go));
        while (go) {
            const v4u rv = *reinterpret_cast<const v4u *>(ip);
            const int res[4] = {(int)rv.x, (int)rv.y, (int)rv.z, (int)rv.w};
            int nv[4], out[4];
            if constexpr (WIDE) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    long long sum = 0;
#pragma unroll
                    for (int q = 0; q < MAXO; q++) { const int t = q < jj ? nv[jj - 1 - q] : hist[q - jj]; sum += (long long)t * (long long)coef[q]; }
                    const long long pr = lshift >= 0 ? (sum >> lshift) : (sum << (-lshift));
                    const long long vv = (long long)res[jj] + pr;
                    const long long oo = vv << wasted;
                    if ((unsigned long long)(oo + (1ll << 29)) >= (1ull << 30) || (unsigned long long)(vv + (1ll << 29)) >= (1ull << 30)) badacc |= 0x80000000u;
                    nv[jj] = (int)vv; out[jj] = (int)oo;
                }
            } else {
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
                    out[jj] = (int)((unsigned)v << wasted);
                }
            }
#pragma unroll
            for (int q = MAXO - 1; q >= 4; q--) hist[q] = hist[q - 4];
#pragma unroll
            for (int q = 0; q < 4 && q < MAXO; q++) hist[q] = nv[3 - q];
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
            op += ostep; ip += 4;
            groups--;
            go = groups > 0;
            if (any2) tp = *reinterpret_cast<const v4u *>(park_addr(j0 + 4 * (g0 - groups), go));
        }
        if (elig) {
            const int done = 4 * (g0 - groups);
            if (done > 0) { if (sv_lo < 0) sv_lo = jpos; sv_hi = jpos + done; }
            jpos += done; left -= done;
        }
    };

    unsigned it = 0;
    bool more = true;
    while (more) {
        if (it > 0) {
            const unsigned bufi = (it - 1) & 1u;
            const unsigned *const rrow = s_res[bufi] + lane * PRS;
            const unsigned *const par = s_par[bufi] + lane * PPW;
            const unsigned m = s_meta[bufi][lane];
            const unsigned fl = m >> 24;
            int left = (int)(m & 0xFFu);
            jpos = (int)((m >> 8) & 0xFFFFu);
            if (fl & PF_A) {
                order = (int)par[16]; wasted = (int)par[17];
                mode = (int)(par[18] & 3u); dswap = (par[18] & 4u) != 0; store_ok = (par[18] & 8u) != 0;
                dsh = (int)par[19]; dmask = (int)par[20]; bs = (int)par[21];
                o_fin = (u64)par[22] | ((u64)par[23] << 32); o_park = (u64)par[24] | ((u64)par[25] << 32);
                wide = par[26] != 0; hb = 1 << 23; lshift = 0;
#pragma unroll
                for (int q = 0; q < SMAXO; q++) coef[q] = 0;
            }
            if (fl & PF_NEWSUB) {
#pragma unroll
                for (int q = 0; q < SMAXO; q++) hist[q] = 0;
            }
            if (fl & PF_B) {
#pragma unroll
                for (int q = 0; q < SMAXO; q++) coef[q] = (int)par[q];
                lshift = (int)par[12]; hb = (int)par[13]; wide = par[14] != 0;
            }
            // (subframe indices beyond 65535 — block sizes are 16 bits: the index of a round's first value travels in 16 — )
            bool go_on = __any(left > 0);
            while (go_on) {
                const bool elig = left >= 4 && (jpos & 3) == 0 && jpos >= order && !(mode == 2 && (bs & 3) != 0);
                if (__any(elig)) {
                    const bool anywide = __any(elig && wide), big = __any(elig && order > 4);
                    if (anywide) { if (big) groups_q(std::integral_constant<int, 12>(), std::true_type(), elig, rrow, left); else groups_q(std::integral_constant<int, 4>(), std::true_type(), elig, rrow, left); }
                    else { if (big) groups_q(std::integral_constant<int, 12>(), std::false_type(), elig, rrow, left); else groups_q(std::integral_constant<int, 4>(), std::false_type(), elig, rrow, left); }
                }
                if (left > 0 && !(left >= 4 && (jpos & 3) == 0 && jpos >= order && !(mode == 2 && (bs & 3) != 0))) {
                    // one value: warm-up samples, the values that bring a run to a multiple of four, a subframe's last ones; 64-bit sums (:411-419)
                    const int v1 = (int)rrow[jpos & (RV - 1)];
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
                    left--; jpos++;
                }
                go_on = __any(left > 0);
            }
            if (badacc & ~(2u * (unsigned)hb - 1u)) s_bad[lane] = 1;   // (the bound is the subframe's: a round's values are one subframe's)
            badacc = 0;
            publish();
            flush();
        }
        __syncthreads();
        more = s_more[it & 1u] != 0;
        it++;
    }
    // (P's last two rounds carry no values: Q has flushed everything)
    if constexpr (O16) { if (__any(bad16 != 0u) && lane == 0) atomicOr(A.flags, 0x100u); }
}

int flac_pq_launch(aukit_ctx *ctx, const FusedArgs &A) {
    if (!A.count) return AUKIT_OK;
    AUKIT_HIP_CHECK(hipMemsetAsync(A.ticket, 0, 4, ctx->stream));
    const unsigned grid = std::min<unsigned>((A.count + 63) / 64, (unsigned)ctx->num_cus * 3u);
    if (A.out16) hipLaunchKernelGGL((k_flac_pq<true>), dim3(grid), dim3(128), 0, ctx->stream, A);
    else hipLaunchKernelGGL((k_flac_pq<false>), dim3(grid), dim3(128), 0, ctx->stream, A);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

}  // namespace aukit
