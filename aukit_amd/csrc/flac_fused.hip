// flac_fused.hip — FLAC frames decoded to FINAL integers by the lane that reads their bits (round 4).
//
// flac.hip's first design splits decodeFrame (aukit.lua:510-567) in two: k_flac_extract turns bit fields into residuals (7.8 GB written for
// BASELINE config 5), k_flac_restore_fast reads them back (9.8 GB), predicts, decorrelates, wraps and writes the samples (8.0 GB) — 13.4 of the
// step's 20.4 ms.  Here one kernel does all of it:
//   * a lane owns one candidate frame (k_flac_find's list, a ticket counter, a persistent grid — as before) and walks it like decodeFrame;
//   * the hot loop is ONE loop over Rice codes and fixed-width fields (escape partitions, warm-up, VERBATIM): a value = alignbit, count-leading-
//     zeros, two bit-field extracts, a select.  Everything else — frame / subframe / partition headers, LPC coefficients, codes longer than 32
//     bits, the last bytes of a stream — is generic code outside it, entered a few times per subframe (the old loop folded the headers in as
//     "fields": 105 instructions per field);
//   * rounds of 32 values are aligned to the subframe, so Rice partitions (multiples of 32 values for every ordinary stream) end where rounds end;
//   * after a round's values are in the lane's LDS row the SAME lane restores the prediction (:411-419) with its history in registers
//     (v_mad_i32_i24 per tap while sum |coef| * |value| provably fits 32 bits, else 64-bit sums), shifts by the wasted bits (:467-469);
//   * the round leaves through LDS as 16-byte stores: independent channels wrapped (:501-507) and final; the first subframe of a decorrelating
//     stereo frame parked (raw) in the SECOND channel's place; the second subframe's rounds read that back, apply :482-497 and write both
//     channels — final values are written once, no descriptor table, no residual array, no second kernel.
// What this kernel does not serve it DECLINES (FE_DECLINE in the frame's CandInfo): predictor orders above 12, values beyond the ranges checked
// below, the :400 quirk (a Rice partition shorter than the predictor order), sample depths of a subframe outside 1..31.  If the chain walk
// (k_flac_chain) needs a declined frame the whole batch is decoded by the first design — so nothing here can change a result.
#include <algorithm>
#include <type_traits>
#include "flac_dev.h"

namespace aukit {

constexpr int FWN = 16;             // 64-bit words of bit-stream window per lane (128 bytes, a ring of 32 dwords)
constexpr int FLPW = FWN / 2;       // lanes that refill one window, 16 bytes each
constexpr int FWS = 2 * FWN + 1;    // row stride of the window array in dwords (odd: lanes at one offset hit distinct banks)
constexpr int FMAXO = 12;           // predictor orders served
constexpr unsigned FRING = 2 * FWN - 1;

enum { S_FRAME = 0, S_SUB, S_RUN, S_CONST, S_COEF, S_PART, S_SUBEND, S_FRAMEEND, S_DONE };

// MSB-first reader of the generic (rare) code, on the lane's ring window in LDS ONLY: a generic step starts with 32 bytes of window in front of it
// (the round ends otherwise and the window slides), which covers every header of an ordinary stream; a read beyond it — a unary run of
// hundreds of zero bits — raises `oow` and the frame is declined.  (With a global-memory path in here hipcc put `s_waitcnt vmcnt(0)` behind
// every field read — a join — and each one waited for the window lines requested at the top of the round.)
struct FRd {
    const unsigned *lw;    // ring: lw[k & 31] = big-endian dword k of the batch (counted from G.w0) for the dwords [2 win_lo, 2 win_lo + 32)
    u64 win_lo, pos, end;
    int eof, oow;
};
AUKIT_DEV unsigned rd_dword(FRd &r, u64 k) {
    if (k - 2 * r.win_lo >= (u64)(2 * FWN)) r.oow = 1;
    return r.lw[(unsigned)k & FRING];
}
AUKIT_DEV unsigned rd_peek(FRd &r) {   // the next 32 bits
    const u64 d = r.pos >> 5;
    const unsigned u = (unsigned)r.pos & 31u;
    const unsigned a = rd_dword(r, d);
    if (u == 0) return a;
    return (a << u) | (rd_dword(r, d + 1) >> (32 - u));
}
AUKIT_DEV unsigned rd_get(FRd &r, int n) {   // BitInputStream.readUint(n), 0 <= n <= 31  (:351-364)
    if (n == 0) return 0;
    if (r.pos + (u64)n > r.end) { r.eof = 1; return 0; }
    const unsigned v = rd_peek(r) >> (32 - n);
    r.pos += (u64)n;
    return v;
}
AUKIT_DEV int rd_sget(FRd &r, int n) {       // readSignedInt(n)  (:365-369)
    const unsigned v = rd_get(r, n);
    return n > 0 ? ((int)(v << (32 - n)) >> (32 - n)) : 0;
}

// the three combinations of :482-497 (first subframe a, second subframe b)
AUKIT_DEV void flac_decor(int asg, int a, int b, int &c0, int &c1) {
    const int right = a - (b >> 1);                                                           // mid / side: floor(side / 2)
    c0 = asg == 8 ? a : (asg == 9 ? a + b : right + b);
    c1 = asg == 8 ? a - b : (asg == 9 ? b : right);
}

// A 16-byte load that only the lanes with `on` make, in straight-line code: hipcc has no such thing (a load under an `if` is a branch, and at the
// join it waits for everything in flight).  The compiler does not know this load exists: the caller waits for it (`s_waitcnt vmcnt(0)`) before
// it reads `dst`, and nothing else may touch `dst` in between (tests/test_isa_schedule.py checks the registers in the built code object).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
// The rows leave with the non-temporal hint: a lane's two 128-byte store runs per round would otherwise push its bit-stream line out of the XCD's
// L2 before the lane comes back for the line's next 16 bytes (PMC, config 5: 8.3 GB fetched for 3.1 GB of bit stream with plain stores, 4.8 GB
// with these; -DAUKIT_FLAC_PLAIN_STORES for the A/B)
__device__ __forceinline__ void st16(int *dst, unsigned x, unsigned y, unsigned z, unsigned w) {
#ifdef AUKIT_FLAC_PLAIN_STORES
    *reinterpret_cast<v4u *>(dst) = v4u{x, y, z, w};
#else
    __builtin_nontemporal_store(v4u{x, y, z, w}, reinterpret_cast<v4u *>(dst));
#endif
}
// O16: two int16 finals per dword (the values were range-checked by the caller)
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st8(short *dst, int a, int b, int c, int d) {
    __builtin_nontemporal_store(v2u{((unsigned)a & 0xFFFFu) | ((unsigned)b << 16), ((unsigned)c & 0xFFFFu) | ((unsigned)d << 16)}, reinterpret_cast<v2u *>(dst));
}
__device__ __forceinline__ unsigned out16(int a, int b, int c, int d) {   // nonzero: a value that does not fit int16
    return (((unsigned)(a + 32768) | (unsigned)(b + 32768) | (unsigned)(c + 32768) | (unsigned)(d + 32768)) >> 16);
}
__device__ __forceinline__ void masked_load16(v4u &dst, const void *addr, bool on) {
    unsigned long long save;
    asm volatile("v_cmp_ne_u32_e32 vcc, 0, %2\n\t"
                 "s_and_saveexec_b64 %1, vcc\n\t"
                 "global_load_dwordx4 %0, %3, off\n\t"
                 "s_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(save) : "v"((unsigned)on), "v"(addr) : "vcc", "memory");
}

__device__ __forceinline__ int fmad24(int a, int b, int c) {   // v_mad_i32_i24 named outright (cf. flac.hip)
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// readRiceSignedInt (:370-376) with every check: codes longer than 32 bits, the end of the data.  Returns FE_OK (value in v), FE_NIL, FE_DECLINE.
AUKIT_DEV int flac_rice_slow(FRd &b, int k, int &v) {
    u64 zeros = 0;
    bool gz = true, dec = false;
    while (gz) {   // (single-exit: see k_flac_chain)
        if (b.pos >= b.end) { b.eof = 1; gz = false; }
        else {
            const unsigned hi = rd_peek(b);
            const u64 avail = b.end - b.pos;
            const int z = hi ? __builtin_clz(hi) : 32;
            if ((u64)z >= avail) { b.eof = 1; gz = false; }   // ran off the end inside the unary prefix
            else if (z < 32) { zeros += (u64)z; b.pos += (u64)z + 1; gz = false; }
            else { zeros += 32; b.pos += 32; if (zeros > (1u << 26)) { dec = true; gz = false; } }
        }
    }
    const unsigned lowb = rd_get(b, k);
    const u64 u = (zeros << k) + lowb;
    if (b.eof) return FE_NIL;
    if (dec || (u >> 31)) return FE_DECLINE;   // beyond int32: the first design's overflow path knows what to do
    v = (int)(unsigned)(u >> 1) ^ -(int)(unsigned)(u & 1);
    return FE_OK;
}

// restoreLinearPrediction (:411-419) over the first `cnt` values of a lane's row, then result[i] * 2^shift (:467-469).  hist[q] = value i - 1 - q.
// WARM: the row may hold warm-up samples (index below `order`: copied).  WIDE: 64-bit sums and range checks (any coefficients, any shift);
// otherwise one 24-bit multiply-add per tap — exact while |value| < hb, a power of two with hb * sum |coef| < 2^31 (checked: `badacc`).
template <int MAXO, bool WARM, bool WIDE>
AUKIT_DEV void flac_predict(int *row, int cnt, int jpos0, int order, int lshift, int wasted, const int (&coef)[FMAXO], int (&hist)[FMAXO], int hb, unsigned &badacc) {
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {
        const int res[4] = {row[k], row[k + 1], row[k + 2], row[k + 3]};
        int nv[4], out[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            int v, o;
            if constexpr (WIDE) {
                long long sum = 0;
#pragma unroll
                for (int q = 0; q < MAXO; q++) { const int t = q < jj ? nv[jj - 1 - q] : hist[q - jj]; sum += (long long)t * (long long)coef[q]; }
                long long pr = lshift >= 0 ? (sum >> lshift) : (sum << (-lshift));   // floor(sum / 2^shift)
                if (WARM && jpos0 + k + jj < order) pr = 0;
                const long long vv = (long long)res[jj] + pr;
                const long long oo = vv << wasted;
                if ((unsigned long long)(oo + (1ll << 29)) >= (1ull << 30) || (unsigned long long)(vv + (1ll << 29)) >= (1ull << 30)) badacc |= 0x80000000u;
                v = (int)vv; o = (int)oo;
            } else {
                // the taps on values of BEFORE these four first, in two chains (they depend on nothing computed here: the four samples' old-tap
                // sums overlap), the taps on the samples just restored last — as one chain of twelve dependent multiply-adds from the newest
                // tap down, every sample waited for the one before it through all twelve (hipcc even pads that chain with s_nop).  Integer
                // sums: any order gives the same bits
#ifdef AUKIT_FLAC_OLD_PRED
                int sum = 0;
#pragma unroll
                for (int q = 0; q < MAXO; q++) { const int t = q < jj ? nv[jj - 1 - q] : hist[q - jj]; sum = fmad24(t, coef[q], sum); }
#else
                int sa = 0, sb = 0;
#pragma unroll
                for (int q = MAXO - 1; q >= jj; q--) { if ((q - jj) & 1) sb = fmad24(hist[q - jj], coef[q], sb); else sa = fmad24(hist[q - jj], coef[q], sa); }
                int sum = sa + sb;
#pragma unroll
                for (int q = jj - 1; q >= 0; q--) sum = fmad24(nv[jj - 1 - q], coef[q], sum);
#endif
                int pr = sum >> lshift;
                if (WARM && jpos0 + k + jj < order) pr = 0;
                v = res[jj] + pr;
                badacc |= (unsigned)(v + hb);
                o = (int)((unsigned)v << wasted);   // |v| < 2^23, wasted <= 6
            }
            nv[jj] = v;
            out[jj] = o;
        }
        row[k] = out[0]; row[k + 1] = out[1]; row[k + 2] = out[2]; row[k + 3] = out[3];
#pragma unroll
        for (int q = MAXO - 1; q >= 4; q--) hist[q] = hist[q - 4];
#pragma unroll
        for (int q = 0; q < 4 && q < MAXO; q++) hist[q] = nv[3 - q];
    }
    for (; k < cnt; k++) {   // a short round's last one to three values
        long long sum = 0;
#pragma unroll
        for (int q = 0; q < MAXO; q++) sum += (long long)hist[q] * (long long)coef[q];
        long long pr = lshift >= 0 ? (sum >> lshift) : (sum << (-lshift));
        if (jpos0 + k < order) pr = 0;
        const long long vv = (long long)row[k] + pr;
        const long long oo = vv << wasted;
        if constexpr (WIDE) { if ((unsigned long long)(oo + (1ll << 29)) >= (1ull << 30) || (unsigned long long)(vv + (1ll << 29)) >= (1ull << 30)) badacc |= 0x80000000u; }
        else { if ((unsigned long long)(vv + (long long)hb) >= 2ull * (unsigned long long)hb) badacc |= 0x80000000u; }
        row[k] = (int)oo;
#pragma unroll
        for (int q = MAXO - 1; q >= 1; q--) hist[q] = hist[q - 1];
        hist[0] = (int)vv;
    }
}

// FNC values per lane and round.  PF: the window lines and the parked first-subframe values are requested a round ahead into registers (80 VGPRs).
// <32, true> holds 237 VGPRs and 17.9 KB of LDS: two waves per SIMD, each issuing an instruction every 8.4 cycles (PMC: 54 % of a wave's cycles
// active, a third parked at s_waitcnt) — a latency-bound instruction stream with too few waves to cover it.  <16, false> fits three (13.3 KB, no
// prefetch registers, __launch_bounds__ tells hipcc so): more per-round overhead per value, and the loads it no longer prefetches are covered by the
// third wave instead.
// O16 (round 4, late; depths <= 16, the loader's resample path): the FINAL values leave as int16 — a frame's region of nsub * bs int32 slots holds
// [channel 0: bs int16][channel 1: bs int16] in its first half and, for decorrelated stereo frames, the parked first subframe (17 bits: int32) in
// its second half.  7.2 + 7.1 GB of config 5's int32 rows become 3.6 + 3.6.  A final that does not fit int16 (garbage input: the reference
// carries whatever the arithmetic gives) raises a flag and the batch is decoded again without O16.
template <int FNC, bool PF, bool O16>
__global__ __launch_bounds__(64, PF ? 2 : 3) void k_flac_decode(const FusedArgs A) {
    constexpr int FOS = FNC + 1;        // row stride of the value array
    constexpr int LPR = FNC / 4;        // lanes that store one row (16 bytes each)
    constexpr int RPI = 64 / LPR;       // rows per store instruction
    __shared__ unsigned s_win[64 * FWS];
    __shared__ int s_val[64 * FOS];
    __shared__ u64 s_ptr[64];                              // where the round's finals go (O16: in int16 units)
    __shared__ u64 s_ptr2[O16 ? 64 : 1];                   // O16: where the parked first subframe lies (int32 units)
    __shared__ unsigned s_meta[64];   // values of the round | mode << 6 | channel assignment << 8 | block size << 12   (mode 0: wrap and store, 1: park the first subframe, 2: decorrelate)
    const int lane = threadIdx.x;
    const int C = A.C, depth = A.depth;
    const int wrap_half = 1 << (depth - 1), wrap_full = 1 << depth;   // 1 <= depth <= 24 (the host sends nothing else here)
    auto wrap = [&](int v) -> int { return v >= wrap_half ? v - wrap_full : v; };   // :504

    FRd b;
    b.lw = s_win + lane * FWS;
    b.win_lo = 0; b.pos = 0; b.end = 0; b.eof = 0; b.oow = 0;
    u64 limit = ~0ull;
    bool have = false, fresh = true;
    unsigned idx = 0, nolimit = 0;
    int st = S_DONE, status = FE_OK;
    int bs = 0, chan_asgn = 0, nsub = 0, ch = 0;
    int order = 0, wasted = 0, sdepth = 0, lshift = 0, after = S_SUBEND;
    int nparts = 0, psize = 0, pi = 0, param_bits = 4, remaining = 0, jpos = 0, rk = 0, cval = 0, hb = 1 << 23;
    bool fixed = false, wide = false, store_ok = false, lpc = false;
    u64 cand_scratch = 0, end_byte = 0;
    int coef[FMAXO], hist[FMAXO];
#pragma unroll
    for (int q = 0; q < FMAXO; q++) { coef[q] = 0; hist[q] = 0; }
    int *const orow = s_val + lane * FOS;

    auto start = [&](unsigned rel) {
        have = true;
        idx = A.first + rel;
        const Cand c = A.cands[idx];
        const FlacStreamInfo si = A.G.info[c.stream];
        (void)si;
        b.end = A.G.base_bit + 8 * A.G.off[c.stream + 1];
        b.pos = A.G.base_bit + 8 * c.byte;
        b.eof = 0; b.oow = 0;
        b.win_lo = 0;
        limit = ~0ull;
        nolimit = c.nolimit;
        st = S_FRAME; status = FE_OK; fresh = true;
        bs = 0; chan_asgn = 0; nsub = 0; ch = 0; jpos = 0; remaining = 0;
        store_ok = false;
        cand_scratch = 0; end_byte = 0;
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

    // ---- a round's values leave at the top of the NEXT round, behind the wait for that round's window lines (loads and stores share vmcnt: stores
    // issued in front of that wait would be waited for as well)
    bool have_flush = false, flush_fast = false;
    uint4 tpre[PF ? LPR : 1];   // mode 2: the parked first-subframe values of the rows this lane stores for, requested while the round is predicted
#pragma unroll
    for (int i = 0; i < (PF ? LPR : 1); i++) tpre[i] = make_uint4(0, 0, 0, 0);
    auto flush = [&]() {
        [[maybe_unused]] unsigned bad16 = 0;
        if (flush_fast) {
            // every requested value is awaited HERE, once, before the first store: hipcc cannot count loads across the branches below and would
            // wait vmcnt(0) at each use — behind the stores of the iteration before it, i.e. for those stores (2.5 of the first version's 10 ms)
            if constexpr (PF) {
#pragma unroll
                for (int i = 0; i < LPR; i++) asm volatile("" : "+v"(tpre[i].x), "+v"(tpre[i].y), "+v"(tpre[i].z), "+v"(tpre[i].w));
            }
            const int grp = lane / LPR, q4 = 4 * (lane % LPR);
#pragma unroll
            for (int i = 0; i < LPR; i++) {
                const int s = RPI * i + grp;
                const unsigned m = s_meta[s];
                const int cn = (int)(m & 0x3Fu), mode = (int)((m >> 6) & 3u), asg = (int)((m >> 8) & 15u), rbs = (int)(m >> 12);
                if (q4 < cn) {
                    const int *v = s_val + s * FOS + q4;
                    const int a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                    if constexpr (O16) {
                        short *dst = reinterpret_cast<short *>(A.scratch) + s_ptr[s] + q4;
                        int *park = A.scratch + s_ptr2[s] + q4;
                        if (mode == 0) { const int w0 = wrap(a0), w1 = wrap(a1), w2 = wrap(a2), w3 = wrap(a3); bad16 |= out16(w0, w1, w2, w3); st8(dst, w0, w1, w2, w3); }
                        else if (mode == 1) st16(park, (unsigned)a0, (unsigned)a1, (unsigned)a2, (unsigned)a3);
                        else {
                            int l0, l1, l2, l3, r0, r1, r2, r3;
                            uint4 tp;
                            if constexpr (PF) tp = tpre[i]; else tp = *reinterpret_cast<const uint4 *>(park);
                            flac_decor(asg, (int)tp.x, a0, l0, r0); flac_decor(asg, (int)tp.y, a1, l1, r1);
                            flac_decor(asg, (int)tp.z, a2, l2, r2); flac_decor(asg, (int)tp.w, a3, l3, r3);
                            l0 = wrap(l0); l1 = wrap(l1); l2 = wrap(l2); l3 = wrap(l3); r0 = wrap(r0); r1 = wrap(r1); r2 = wrap(r2); r3 = wrap(r3);
                            bad16 |= out16(l0, l1, l2, l3) | out16(r0, r1, r2, r3);
                            st8(dst, l0, l1, l2, l3);
                            st8(dst + rbs, r0, r1, r2, r3);
                        }
                    } else {
                    int *dst = A.scratch + s_ptr[s] + q4;
                    if (mode == 0) st16(dst, (unsigned)wrap(a0), (unsigned)wrap(a1), (unsigned)wrap(a2), (unsigned)wrap(a3));
                    else if (mode == 1) st16(dst, (unsigned)a0, (unsigned)a1, (unsigned)a2, (unsigned)a3);
                    else {
                        int l0, l1, l2, l3, r0, r1, r2, r3;
                        uint4 tp;
                        if constexpr (PF) tp = tpre[i]; else tp = *reinterpret_cast<const uint4 *>(dst + rbs);   // (no prefetch: read where it is needed; the other waves cover it)
                        flac_decor(asg, (int)tp.x, a0, l0, r0); flac_decor(asg, (int)tp.y, a1, l1, r1);
                        flac_decor(asg, (int)tp.z, a2, l2, r2); flac_decor(asg, (int)tp.w, a3, l3, r3);
                        st16(dst, (unsigned)wrap(l0), (unsigned)wrap(l1), (unsigned)wrap(l2), (unsigned)wrap(l3));
                        st16(dst + rbs, (unsigned)wrap(r0), (unsigned)wrap(r1), (unsigned)wrap(r2), (unsigned)wrap(r3));
                    }
                    }
                }
            }
            if constexpr (O16) { if (__any(bad16 != 0u) && lane == 0) atomicOr(A.flags, 0x100u); }
            return;
        }
        const int part = lane / FNC, k = lane % FNC;   // any alignment, any count: 64 / FNC rows per instruction
        for (int i = 0; i < FNC; i++) {
            const int s = (64 / FNC) * i + part;
            const unsigned m = s_meta[s];
            const int cn = (int)(m & 0x3Fu), mode = (int)((m >> 6) & 3u), asg = (int)((m >> 8) & 15u), rbs = (int)(m >> 12);
            if (k < cn) {
                const int a = s_val[s * FOS + k];
                if constexpr (O16) {
                    short *dst = reinterpret_cast<short *>(A.scratch) + s_ptr[s] + k;
                    int *park = A.scratch + s_ptr2[s] + k;
                    if (mode == 0) { const int w = wrap(a); bad16 |= out16(w, 0, 0, 0); *dst = (short)w; }
                    else if (mode == 1) *park = a;
                    else { int l, r; flac_decor(asg, *park, a, l, r); l = wrap(l); r = wrap(r); bad16 |= out16(l, r, 0, 0); dst[0] = (short)l; dst[rbs] = (short)r; }
                } else {
                int *dst = A.scratch + s_ptr[s] + k;
                if (mode == 0) *dst = wrap(a);
                else if (mode == 1) *dst = a;
                else { int l, r; flac_decor(asg, dst[rbs], a, l, r); dst[0] = wrap(l); dst[rbs] = wrap(r); }
                }
            }
        }
        if constexpr (O16) { if (__any(bad16 != 0u) && lane == 0) atomicOr(A.flags, 0x100u); }
    };
    v4u pf[PF ? FLPW : 1];        // per window this lane loads for: the line that will replace its slot's line, already requested
    u64 pf_line[PF ? FLPW : 1];
#pragma unroll
    for (int i = 0; i < (PF ? FLPW : 1); i++) { pf[i] = v4u{0, 0, 0, 0}; pf_line[i] = ~0ull; }

#ifdef AUKIT_FLAC_STATS
    u64 st_rounds = 0, st_outer = 0, st_turns = 0, st_values = 0, st_lane_turns = 0, st_live = 0;
#endif
    bool more = true;
    while (more) {
        // ---- slide the LDS windows of the lanes that have used a quarter of theirs: 8 lanes × 16 bytes per window, 8 windows per load.
        // Three passes over the eight windows a lane loads for, so that ONE wait covers the round's loads: hipcc cannot count loads across the
        // branches of a pass and waits vmcnt(0) at every use — with "use line i, request its successor" in one pass every iteration waited for
        // the request of the iteration before it (eight memory latencies per round: half of the first version's 9.5 ms).
        {
            const u64 wi = b.pos >> 6;
            const bool want = st != S_DONE && (fresh || (wi - b.win_lo) >= (u64)(FWN / 4));
            const u64 new_lo = wi & ~1ull;
            const u64 keep_from = fresh ? ~0ull : b.win_lo / 2 + FWN / 2;   // first 16-byte line the ring does not hold yet
            if (want) b.win_lo = new_lo;
            const int sub8 = lane % FLPW, grp = lane / FLPW;
            unsigned act = 0;
            if constexpr (PF) {
            // the lines requested a round ago (masked_load16 in pass 3: loads hipcc does not see) are awaited here, once and unconditionally
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < FLPW; i++) asm volatile("" : "+v"(pf[i]));
            if (__any(want)) {
#pragma unroll
                for (int i = 0; i < FLPW; i++) {   // pass 1: which lines move in; the ones that were not requested a round ago (a new frame) are requested now
                    const int s = i * (64 / FLPW) + grp;
                    const int w = __shfl((int)want, s);
                    const u64 ws = __shfl(new_lo, s);
                    const u64 kf = __shfl(keep_from, s);
                    const u64 l0 = ws / 2, line = l0 + (((u64)sub8 - l0) & (u64)(FLPW - 1));   // ring slot sub8 holds the line congruent to sub8
                    if (w && (kf == ~0ull || line >= kf || line < kf - FWN / 2)) {
                        act |= 1u << i;
                        if (pf_line[i] != line) {
                            pf_line[i] = line;
                            pf[i] = v4u{0, 0, 0, 0};
                            if (2 * line < A.G.safe_words) pf[i] = *reinterpret_cast<const v4u *>(A.G.w0 + 2 * line);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < FLPW; i++) {   // pass 2: into the rings
                    if (act & (1u << i)) {
                        unsigned *wrow = s_win + (i * (64 / FLPW) + grp) * FWS + 4 * sub8;
                        const bool inr = 2 * pf_line[i] < A.G.safe_words;   // (a line beyond the batch was requested at a dummy address: zeros)
                        wrow[0] = inr ? __builtin_bswap32(pf[i].x) : 0u; wrow[1] = inr ? __builtin_bswap32(pf[i].y) : 0u;
                        wrow[2] = inr ? __builtin_bswap32(pf[i].z) : 0u; wrow[3] = inr ? __builtin_bswap32(pf[i].w) : 0u;
                    }
                }
            }
            } else if (__any(want)) {
                // no prefetch registers: every line that moves in is read now, all eight requests in flight before the first is used (straight-line:
                // an inactive slot reads the batch's first vector) — the wait is covered by the SIMD's other waves
                uint4 ld[FLPW];
                unsigned actm = 0;
#pragma unroll
                for (int i = 0; i < FLPW; i++) {
                    const int s = i * (64 / FLPW) + grp;
                    const int w = __shfl((int)want, s);
                    const u64 ws = __shfl(new_lo, s);
                    const u64 kf = __shfl(keep_from, s);
                    const u64 l0 = ws / 2, line = l0 + (((u64)sub8 - l0) & (u64)(FLPW - 1));
                    const bool on = w && (kf == ~0ull || line >= kf || line < kf - FWN / 2);
                    const bool inr = on && 2 * line < A.G.safe_words;
                    actm |= (on ? 1u : 0u) << i;
                    actm |= (inr ? 1u : 0u) << (8 + i);
                    ld[i] = *reinterpret_cast<const uint4 *>(A.G.w0 + (inr ? 2 * line : 0ull));
                }
#pragma unroll
                for (int i = 0; i < FLPW; i++) {
                    if (actm & (1u << i)) {
                        unsigned *wrow = s_win + (i * (64 / FLPW) + grp) * FWS + 4 * sub8;
                        const bool inr = (actm >> (8 + i)) & 1u;
                        wrow[0] = inr ? __builtin_bswap32(ld[i].x) : 0u; wrow[1] = inr ? __builtin_bswap32(ld[i].y) : 0u;
                        wrow[2] = inr ? __builtin_bswap32(ld[i].z) : 0u; wrow[3] = inr ? __builtin_bswap32(ld[i].w) : 0u;
                    }
                }
            }
            __syncthreads();
            if (have_flush && !(A.dbg & 2)) flush();
            if constexpr (PF)
#pragma unroll
            for (int i = 0; i < FLPW; i++) {       // pass 3: a slot that moved asks for its next line now, a round or more before it is needed.
                // Only the slots that moved (masked_load16): until round 4's last pass every slot asked every round — "a hit in L2" for the ones
                // that kept their line, it said here; PMC said 16.1 GB fetched for 3.1 GB of bit stream: with two 128-byte store runs per lane
                // in flight the per-XCD L2 (4 MiB for 16 K lanes) does not keep a lane's line from one round to the next
                const bool mv = (act >> i) & 1u;
                const u64 nl = pf_line[i] + (mv ? (u64)FLPW : 0ull);
                pf_line[i] = nl;
                masked_load16(pf[i], A.G.w0 + 2 * nl, mv && 2 * nl < A.G.safe_words);
            }
            __syncthreads();
            fresh = false;
        }

        // ---- phase 1: every lane advances its frame until it has 32 values of ONE subframe (or its window runs low, or the subframe ends)
        int cnt = 0;
        int lim = FNC - (jpos & (FNC - 1));   // rounds end at multiples of 32 values of the subframe: where Rice partitions end
        bool rdone = st == S_DONE;
        const unsigned dwbase = (unsigned)(2 * b.win_lo);
        // a lane whose window reaches the end of its stream's data reads value by value with every check (a stream's last rounds)
        const bool careful = b.end < ((b.win_lo + FWN) << 6) + 64;
        bool go_on = __any(!rdone);
#ifdef AUKIT_FLAC_STATS
        st_rounds++; st_live += (u64)__builtin_popcountll(__ballot(have && st != S_DONE));
#endif
        while (go_on) {
#ifdef AUKIT_FLAC_STATS
            st_outer++;
#endif
            // -- the run loop: Rice codes (:370-376) or fields of `rk` bits (:405, :423, :457), one value per turn
            const bool run = !rdone && st == S_RUN && remaining > 0 && !careful;
            if (__any(run)) {
                const u64 dfull = (b.pos - 1) >> 5;                     // the dword that holds the last bit read (frames start far beyond bit 0)
                unsigned d = (unsigned)dfull;
                int s = (int)((0u - (unsigned)b.pos) & 31u);            // bits of it not yet read
                unsigned w0 = b.lw[d & FRING], w1 = b.lw[(d + 1) & FRING];
                asm volatile("" : "+v"(w0), "+v"(w1));   // awaited HERE: pending at the loop header they would put an lgkmcnt(0) there — which also waits for every turn's store
                const int cnt0 = cnt;
                bool slow = false;
                const unsigned dlim = dwbase + 24u;                     // the window is low beyond this dword
                int n_it = min(lim - cnt, remaining);                   // values this lane may still take in this round
                int *wp = orow + cnt;
                bool go = run && n_it > 0 && (int)(d - dlim) < 0;
                // (tried, round 4 late: a second copy of this loop without the field arithmetic for Rice-only rounds, and a turn that commits
                // unconditionally with the last one taken back — 36 -> 31 VALU instructions per value on paper, 8.0 -> 8.4 / 8.5 ms measured)
                while (go) {
#ifdef AUKIT_FLAC_STATS
                    st_turns++; st_lane_turns += (u64)__builtin_popcountll(__ballot(true));
#endif
                    const unsigned wn = b.lw[(d + 2) & FRING];          // for a crossing at the END of this turn: requested first, used last
                    const unsigned hi = __builtin_amdgcn_alignbit(w0, w1, (unsigned)s);
                    const int z = hi ? __builtin_clz(hi) : 32;
                    const int tot_r = z + 1 + rk;
                    const unsigned low = __builtin_amdgcn_ubfe(hi, (unsigned)(31 - rk - z) & 31u, (unsigned)rk);
                    const unsigned ur = ((unsigned)z << rk) | low;
                    const int v_r = (int)(ur >> 1) ^ -(int)(ur & 1u);
                    const int v_f = __builtin_amdgcn_sbfe((int)hi, (unsigned)(32 - rk) & 31u, (unsigned)rk);
                    const int tot = fixed ? rk : tot_r;
                    const bool ok = tot <= 32;              // else a Rice code longer than 32 bits: the generic reader takes this one (nothing moves)
                    *wp = fixed ? v_f : v_r;                // (a value that is not ok is overwritten by the generic reader's)
                    s -= ok ? tot : 0;
                    const bool cross = s < 0;
                    s &= 31;
                    d += cross ? 1u : 0u;
                    w0 = cross ? w1 : w0;
                    w1 = cross ? wn : w1;
                    wp += ok ? 1 : 0; n_it -= ok ? 1 : 0;
                    slow = !ok;
                    go = ok && n_it > 0 && (int)(d - dlim) < 0;
                }
                if (run) { const int got = (int)(wp - orow) - cnt; cnt += got; remaining -= got; }
                (void)cnt0;
                if (run) {
                    b.pos = 32 * (dfull + (u64)(d - (unsigned)dfull) + 1) - (u64)s;
                    jpos += cnt - cnt0;
                    if (slow) {   // this one value by the generic reader
                        int v1 = 0;
                        int r1 = flac_rice_slow(b, rk, v1);
                        if (!r1 && b.oow) r1 = FE_DECLINE;
                        if (r1) { status = r1; st = S_DONE; rdone = true; }
                        else { orow[cnt] = v1; cnt++; remaining--; jpos++; }
                    }
                    if (!rdone && remaining > 0 && (cnt >= lim || ((b.pos >> 5) - (u64)2 * b.win_lo) >= 24)) rdone = true;   // the round is full, or the window low
                }
            }
            // -- everything else, one transition per turn (divergent, rare)
            if (!rdone && ((b.pos >> 5) - (u64)2 * b.win_lo) >= 24 && st != S_SUBEND && st != S_FRAMEEND && !(st == S_RUN && remaining == 0)) rdone = true;   // a step that reads wants 32 bytes of window
            if (!rdone) {
                if (st == S_RUN) {
                    if (remaining == 0) { st = after; if (after == S_PART) { pi++; if (pi >= nparts) st = S_SUBEND; } }
                    else if (cnt >= lim) rdone = true;
                    else if (careful) {   // value by value near the end of the data
                        int r1 = FE_OK, v1 = 0;
                        if (fixed) { v1 = rd_sget(b, rk); if (b.eof) r1 = FE_NIL; }
                        else r1 = flac_rice_slow(b, rk, v1);
                        if (r1) { status = r1; st = S_DONE; rdone = true; }
                        else { orow[cnt] = v1; cnt++; remaining--; jpos++; }
                        if (!rdone && remaining > 0 && (cnt >= lim || ((b.pos >> 5) - (u64)2 * b.win_lo) >= 24)) rdone = true;
                    }
                } else if (st == S_CONST) {   // :453-454
                    while (remaining > 0 && cnt < lim) { orow[cnt++] = cval; remaining--; jpos++; }
                    if (remaining == 0) st = S_SUBEND; else rdone = true;
                } else if (st == S_PART) {   // :394-406
                    const int escape = param_bits == 4 ? 15 : 31;
                    const int param = (int)rd_get(b, param_bits);
                    const bool esc = param >= escape;
                    int nbits = 0;
                    if (esc) nbits = (int)rd_get(b, 5);
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
                } else if (st == S_SUB) {   // decodeSubframe  :443-465
                    rd_get(b, 1);
                    const int type = (int)rd_get(b, 6);
                    wasted = (int)rd_get(b, 1);
                    if (wasted == 1) {   // unary wasted-bits count  :447-449
                        bool gw = true;
                        while (gw) { const unsigned bit = rd_get(b, 1); if (b.eof || bit) gw = false; else wasted++; }
                    }
                    sdepth = depth - wasted;
                    if (chan_asgn >= 8) sdepth += ((chan_asgn == 9) == (ch == 0)) ? 1 : 0;   // the side channel has one more bit  :480-481
                    order = 0; lshift = 0; jpos = 0; lim = FNC; lpc = false; wide = false; hb = 1 << 23;
#pragma unroll
                    for (int q = 0; q < FMAXO; q++) { coef[q] = 0; hist[q] = 0; }
                    if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                    else if (sdepth < 1 || sdepth > 31 || wasted > 24) { status = FE_DECLINE; st = S_DONE; rdone = true; }
                    else if (type == 0) {
                        cval = rd_sget(b, sdepth);
                        wide = sdepth > 24 || wasted > 6;
                        if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                        else { remaining = bs; st = S_CONST; }
                    } else if (type == 1) { wide = sdepth > 24 || wasted > 6; remaining = bs; fixed = true; rk = sdepth; after = S_SUBEND; st = S_RUN; }
                    else if ((type >= 8 && type <= 12) || (type >= 32 && type <= 63)) {
                        order = type <= 12 ? type - 8 : type - 31;
                        lpc = type >= 32;
                        if (order > FMAXO || order > bs) { status = FE_DECLINE; st = S_DONE; rdone = true; }   // (order > bs: the Lua table grows past blockSize)
                        else { remaining = order; fixed = true; rk = sdepth; after = S_COEF; st = S_RUN; }
                    } else { status = FE_SUBTYPE; st = S_DONE; rdone = true; }
                } else if (st == S_COEF) {   // :433-438 / FIXED_PREDICTION_COEFFICIENTS :334-340, then the residual header :381-391
                    if (lpc) {
                        const int precision = (int)rd_get(b, 4) + 1;
                        lshift = rd_sget(b, 5);
#pragma unroll
                        for (int q = 0; q < FMAXO; q++) if (q < order) coef[q] = rd_sget(b, precision);
                    } else {
                        // FIXED_PREDICTION_COEFFICIENTS[order + 1] = {}, {1}, {2, -1}, {3, -3, 1}, {4, -6, 4, -1}: binomials, by arithmetic (a table
                        // would be a load from constant memory, and a vector-memory load in here puts a wait for the window prefetch behind it)
                        coef[0] = order;
                        coef[1] = order == 2 ? -1 : (order == 3 ? -3 : (order == 4 ? -6 : 0));
                        coef[2] = order == 3 ? 1 : (order == 4 ? 4 : 0);
                        coef[3] = order == 4 ? -1 : 0;
                    }
                    const int method = (int)rd_get(b, 2);
                    param_bits = method == 0 ? 4 : 5;
                    const int porder = (int)rd_get(b, 4);
                    nparts = 1 << porder;
                    int sabs = 1;
#pragma unroll
                    for (int q = 0; q < FMAXO; q++) sabs += coef[q] < 0 ? -coef[q] : coef[q];
                    const int hbits = min(23, __builtin_clz((unsigned)sabs) - 1);   // 2^hbits * sum |coef| < 2^31
                    hb = 1 << hbits;
                    // values of this subframe have up to sdepth bits: the 24-bit multiply-adds serve it when those fit under hb; else 64-bit sums
                    wide = sdepth - 1 > hbits || lshift < 0 || wasted > 6;
                    if (b.eof) { status = FE_NIL; st = S_DONE; rdone = true; }
                    else if (method >= 2) { status = FE_RESMETHOD; st = S_DONE; rdone = true; }
                    else if (bs % nparts != 0) { status = FE_PARTITION; st = S_DONE; rdone = true; }
                    else {
                        psize = bs / nparts;
                        pi = 0;
                        if (nparts > 1 && psize < order) { status = FE_DECLINE; st = S_DONE; rdone = true; }   // :400 — later partitions overwrite warm-up entries
                        else st = S_PART;
                    }
                } else if (st == S_SUBEND) {
                    if (cnt > 0) rdone = true;   // this subframe's last values are predicted and stored before the next header is read
                    else { ch++; jpos = 0; lim = FNC; st = ch < nsub ? S_SUB : S_FRAMEEND; }
                } else if (st == S_FRAME) {   // decodeFrame header  :510-553
                    int fs = FE_OK;
                    const unsigned t0 = rd_get(b, 8);
                    if (b.eof) fs = FE_EOF_START;
                    const unsigned sync = t0 * 64 + rd_get(b, 6);
                    if (!fs && b.eof) fs = FE_NIL;
                    if (!fs && sync != 0x3FFE) fs = FE_SYNC;
                    rd_get(b, 2);
                    const int bsc = (int)rd_get(b, 4), src_code = (int)rd_get(b, 4);
                    chan_asgn = (int)rd_get(b, 4);
                    rd_get(b, 4);
                    const int t = (int)rd_get(b, 8);
                    if (!fs && b.eof) fs = FE_NIL;
                    int t2 = -1;
                    for (int i = 7; i >= 0; i--) { if (!(t & (1 << i))) break; t2++; }
                    for (int i = 1; i <= t2; i++) rd_get(b, 8);
                    if (bsc == 1) bs = 192;
                    else if (bsc >= 2 && bsc <= 5) bs = 576 << (bsc - 2);
                    else if (bsc == 6) bs = (int)rd_get(b, 8) + 1;
                    else if (bsc == 7) bs = (int)rd_get(b, 16) + 1;
                    else if (bsc >= 8) bs = 256 << (bsc - 8);
                    else { bs = 0; if (!fs) fs = FE_BLOCKSIZE; }
                    if (src_code == 12) rd_get(b, 8);
                    else if (src_code == 13 || src_code == 14) rd_get(b, 16);
                    rd_get(b, 8);   // CRC-8, ignored :553
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
                        ch = 0; jpos = 0; lim = FNC;
                        st = S_SUB;
                    }
                } else if (st == S_FRAMEEND) {   // :555-557
                    b.pos = (b.pos + 7) & ~7ull;                                  // alignToByte (frames start on byte boundaries of the batch buffer)
                    b.pos = (b.pos + 16 <= b.end) ? b.pos + 16 : b.end;           // readUint(16): a nil here is discarded, the NEXT readByte returns nil
                    end_byte = (b.pos - A.G.base_bit) >> 3;
                    st = S_DONE; rdone = true;
                }
                if (b.oow && st != S_DONE) { status = FE_DECLINE; st = S_DONE; rdone = true; }   // a field beyond the window: not an ordinary stream
            }
            go_on = __any(!rdone);
        }
        if (st != S_DONE && b.pos > limit) { status = FE_LIMIT; st = S_DONE; }

#ifdef AUKIT_FLAC_STATS
        { int sv = cnt; for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o); st_values += (u64)sv; }
#endif
        // ---- where the round's values go
        const bool decor = C == 2 && chan_asgn >= 8 && chan_asgn <= 10;
        const int mode = !decor ? 0 : (ch == 0 ? 1 : 2);
        const int jpos0 = jpos - cnt;   // the subframe index of the row's first value (a round never mixes subframes)
        // (O16: finals in int16 units from twice the region's int32 offset; the parking area is the region's second half, in int32 units)
        const u64 gptr = O16 ? 2 * cand_scratch + (u64)jpos0 + (mode == 0 ? (u64)ch * (u64)bs : 0ull)
                             : cand_scratch + (u64)jpos0 + (mode == 0 ? (u64)ch * (u64)bs : (mode == 1 ? (u64)bs : 0ull));
        if constexpr (O16) s_ptr2[lane] = cand_scratch + (u64)bs + (u64)jpos0;
        const bool live = have && cnt > 0 && status == FE_OK;
        const bool stores = live && store_ok;
        s_meta[lane] = stores ? ((unsigned)cnt | ((unsigned)mode << 6) | ((unsigned)chan_asgn << 8) | ((unsigned)bs << 12)) : 0u;
        s_ptr[lane] = gptr;
        // the 16-byte path needs whole vectors at aligned places (block sizes are multiples of 4 but for a stream's last frame)
        flush_fast = __all(!stores || ((cnt & 3) == 0 && (gptr & 3) == 0 && ((mode != 2 && !(O16 && mode == 1)) || (bs & 3) == 0)));
        __syncthreads();
        {   // the parked first-subframe values of the rounds that decorrelate: requested now, used by the flush at the top of the next round.
            // Straight-line and unconditional (a lane with nothing to fetch reads the scratch's first vector): inside an `if` hipcc merges the loaded
            // registers with the old ones at the join — moves that wait for the loads right here
            if constexpr (PF) {
            const bool ff = flush_fast && !(A.dbg & 4);
            const int grp = lane / LPR, q4 = 4 * (lane % LPR);
#pragma unroll
            for (int i = 0; i < LPR; i++) {
                const int s = RPI * i + grp;
                const unsigned m = s_meta[s];
                const bool need = ff && ((m >> 6) & 3u) == 2u && q4 < (int)(m & 0x3Fu);
                if constexpr (O16) tpre[i] = *reinterpret_cast<const uint4 *>(A.scratch + (need ? s_ptr2[s] + (u64)q4 : 0ull));
                else tpre[i] = *reinterpret_cast<const uint4 *>(A.scratch + (need ? s_ptr[s] + (u64)(m >> 12) + (u64)q4 : 0ull));
            }
            }
        }
        // ---- phase 2: the prediction, by the lane that decoded the values
        if (__any(live) && !(A.dbg & 1)) {
            unsigned badacc = 0;
            const int pcnt = live ? cnt : 0;
            const bool anywide = __any(live && wide);
            const bool warm = __any(live && jpos0 < order);
            const bool big = __any(live && order > 4);
#define AUKIT_PRED(MO, W, WD) flac_predict<MO, W, WD>(orow, pcnt, jpos0, order, lshift, wasted, coef, hist, hb, badacc)
            if (anywide) { if (big) { if (warm) AUKIT_PRED(12, true, true); else AUKIT_PRED(12, false, true); } else { if (warm) AUKIT_PRED(4, true, true); else AUKIT_PRED(4, false, true); } }
            else { if (big) { if (warm) AUKIT_PRED(12, true, false); else AUKIT_PRED(12, false, false); } else { if (warm) AUKIT_PRED(4, true, false); else AUKIT_PRED(4, false, false); } }
#undef AUKIT_PRED
            const bool bad = anywide ? (badacc >> 31) != 0 : ((badacc & ~(2u * (unsigned)hb - 1u)) != 0);
            if (live && bad) { status = FE_DECLINE; st = S_DONE; s_meta[lane] = 0u; }
        }
        have_flush = true;
        if (have && st == S_DONE) finish();
        if (__ballot(have) == 0) take();   // new frames when the whole wave is through with its old ones: the lanes then parse their headers in the same rounds
        more = __ballot(st != S_DONE) != 0;
    }
    __syncthreads();
    flush();
#ifdef AUKIT_FLAC_STATS
    if (lane == 0) { atomicAdd(A.stats + 0, st_rounds); atomicAdd(A.stats + 1, st_outer); atomicAdd(A.stats + 2, st_turns); atomicAdd(A.stats + 3, st_values); atomicAdd(A.stats + 4, st_lane_turns); atomicAdd(A.stats + 5, st_live); }
#endif
}

int flac_fused_launch(aukit_ctx *ctx, const FusedArgs &A) {
    if (!A.count) return AUKIT_OK;
    AUKIT_HIP_CHECK(hipMemsetAsync(A.ticket, 0, 4, ctx->stream));
    static const int variant = getenv("AUKIT_FLAC_FUSED_VARIANT") ? atoi(getenv("AUKIT_FLAC_FUSED_VARIANT")) : 0;   // 0: <32, prefetch> two waves per SIMD; 1: <16, none> three
    static const int wgs = getenv("AUKIT_FLAC_FUSED_WGS") ? atoi(getenv("AUKIT_FLAC_FUSED_WGS")) : (variant == 1 ? 12 : 8);
    const unsigned grid = std::min<unsigned>((A.count + 63) / 64, (unsigned)ctx->num_cus * (unsigned)std::max(wgs, 1));
    if (A.out16) hipLaunchKernelGGL((k_flac_decode<32, true, true>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (variant == 1) hipLaunchKernelGGL((k_flac_decode<16, false, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else hipLaunchKernelGGL((k_flac_decode<32, true, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

__global__ __launch_bounds__(64) void k_flac_frames(const Cand *cands, const CandInfo *ci, unsigned ncand, const u64 *frame_base, FrameRec *frames, const u64 *stream_off) {
    const unsigned k = blockIdx.x * 64 + threadIdx.x;
    if (k >= ncand) return;
    const CandInfo f = ci[k];
    if (!f.used) return;
    const unsigned s = cands[k].stream;
    const u64 rel = f.end_byte - stream_off[s];
    frames[frame_base[s] + f.seq] = FrameRec{f.sample_off, f.scratch, f.blocksize, f.chan_asgn, s, rel < 0xFFFFFFFFull ? (unsigned)rel : 0u};
}
int flac_frames_launch(aukit_ctx *ctx, const Cand *cands, const CandInfo *ci, unsigned ncand, const u64 *frame_base, FrameRec *frames, const u64 *stream_off) {
    if (!ncand) return AUKIT_OK;
    hipLaunchKernelGGL(k_flac_frames, dim3((ncand + 63) / 64), dim3(64), 0, ctx->stream, cands, ci, ncand, frame_base, frames, stream_off);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

// chained frames: scratch → rows.  One workgroup per frame; 16 bytes per thread and turn where everything is aligned.  OUT = int: the decoder's
// integers as they are; float / double: the loader's `s / 2^depth` (:505; an exact scaling) straight into an audio's rows.
template <typename OUT, bool S16 = false>
__global__ __launch_bounds__(256) void k_flac_gather(const FrameRec *frames, int C, const u64 *row_off, const u64 *a_meta, unsigned n, const int *scratch, OUT *rows, double inv_full) {
    const FrameRec f = frames[blockIdx.x];
    const int nch = f.chan_asgn >= 8 ? 2 : C;
    if constexpr (S16) {   // int16 finals (k_flac_decode<..., O16>) to int32 rows: the consumers that want rows want them as before
        const short *base = reinterpret_cast<const short *>(scratch) + 2 * f.scratch;
        for (int c = 0; c < nch; c++) {
            const short *src = base + (u64)c * (u64)f.bs;
            OUT *dst = rows + (row_off ? row_off[(size_t)f.stream * C + c] : a_meta[n + f.stream] + (u64)c * a_meta[2 * (size_t)n + f.stream]) + f.sample_off;
            if constexpr (std::is_same<OUT, int>::value) { for (int i = threadIdx.x; i < f.bs; i += 256) dst[i] = (OUT)src[i]; }
            else if ((((uintptr_t)src) & 7) == 0 && (((uintptr_t)dst) & 15) == 0 && (f.bs & 3) == 0) {   // four samples a turn: 8 bytes in, 16 / 32 out
                for (int i = threadIdx.x; i < f.bs / 4; i += 256) {
                    const uint2 v = reinterpret_cast<const uint2 *>(src)[i];
                    typedef OUT ov4 __attribute__((ext_vector_type(4), aligned(16)));
                    ov4 w;
                    w[0] = (OUT)((double)(short)(v.x & 0xFFFF) * inv_full); w[1] = (OUT)((double)((int)v.x >> 16) * inv_full);
                    w[2] = (OUT)((double)(short)(v.y & 0xFFFF) * inv_full); w[3] = (OUT)((double)((int)v.y >> 16) * inv_full);
                    *reinterpret_cast<ov4 *>(dst + 4 * i) = w;
                }
            } else { for (int i = threadIdx.x; i < f.bs; i += 256) dst[i] = (OUT)((double)src[i] * inv_full); }
        }
        return;
    }
    for (int c = 0; c < nch; c++) {
        const int *src = scratch + f.scratch + (u64)c * (u64)f.bs;
        OUT *dst = rows + (row_off ? row_off[(size_t)f.stream * C + c] : a_meta[n + f.stream] + (u64)c * a_meta[2 * (size_t)n + f.stream]) + f.sample_off;
        if constexpr (std::is_same<OUT, int>::value) {
            if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (f.bs & 3) == 0) {
                for (int i = threadIdx.x; i < f.bs / 4; i += 256) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
            } else {
                for (int i = threadIdx.x; i < f.bs; i += 256) dst[i] = src[i];
            }
        } else {
            if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (f.bs & 3) == 0) {
                for (int i = threadIdx.x; i < f.bs / 4; i += 256) {
                    const uint4 v = reinterpret_cast<const uint4 *>(src)[i];
                    typedef OUT ov4 __attribute__((ext_vector_type(4), aligned(16)));
                    ov4 w;
                    w[0] = (OUT)((double)(int)v.x * inv_full); w[1] = (OUT)((double)(int)v.y * inv_full);
                    w[2] = (OUT)((double)(int)v.z * inv_full); w[3] = (OUT)((double)(int)v.w * inv_full);
                    *reinterpret_cast<ov4 *>(dst + 4 * i) = w;
                }
            } else {
                for (int i = threadIdx.x; i < f.bs; i += 256) dst[i] = (OUT)((double)src[i] * inv_full);
            }
        }
    }
}
int flac_gather_launch(aukit_ctx *ctx, const FrameRec *frames, u64 nfr, int C, const u64 *row_off, const int *scratch, int *rows, bool scratch16) {
    if (!nfr) return AUKIT_OK;
    if (scratch16) hipLaunchKernelGGL((k_flac_gather<int, true>), dim3((unsigned)nfr), dim3(256), 0, ctx->stream, frames, C, row_off, (const u64 *)nullptr, 0u, scratch, rows, 1.0);
    else hipLaunchKernelGGL((k_flac_gather<int>), dim3((unsigned)nfr), dim3(256), 0, ctx->stream, frames, C, row_off, (const u64 *)nullptr, 0u, scratch, rows, 1.0);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}
int flac_gather_convert_launch(aukit_ctx *ctx, const FrameRec *frames, u64 nfr, int C, const int *scratch, const u64 *a_meta, unsigned n, void *out, int dtype, double full, bool scratch16) {
    if (!nfr) return AUKIT_OK;
    if (scratch16) {
        if (dtype == AUKIT_F32) hipLaunchKernelGGL((k_flac_gather<float, true>), dim3((unsigned)nfr), dim3(256), 0, ctx->stream, frames, C, (const u64 *)nullptr, a_meta, n, scratch, reinterpret_cast<float *>(out), 1.0 / full);
        else hipLaunchKernelGGL((k_flac_gather<double, true>), dim3((unsigned)nfr), dim3(256), 0, ctx->stream, frames, C, (const u64 *)nullptr, a_meta, n, scratch, reinterpret_cast<double *>(out), 1.0 / full);
        AUKIT_HIP_CHECK(hipGetLastError());
        return AUKIT_OK;
    }
    if (dtype == AUKIT_F32) hipLaunchKernelGGL((k_flac_gather<float>), dim3((unsigned)nfr), dim3(256), 0, ctx->stream, frames, C, (const u64 *)nullptr, a_meta, n, scratch, reinterpret_cast<float *>(out), 1.0 / full);
    else hipLaunchKernelGGL((k_flac_gather<double>), dim3((unsigned)nfr), dim3(256), 0, ctx->stream, frames, C, (const u64 *)nullptr, a_meta, n, scratch, reinterpret_cast<double *>(out), 1.0 / full);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

}  // namespace aukit
