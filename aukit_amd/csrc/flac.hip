// flac.hip — frame-parallel FLAC decode (aukit.lua:311-619: decodeFLAC, a port of Nayuki's simple decoder) on gfx950.
//
// FLAC frames carry no length, and the reference simply decodes them one after another (aukit.lua:615), ignoring
// the CRCs (:553, :557).  To decode frames in parallel without changing which bytes are treated as frames:
//   1. k_flac_header   one lane per stream walks the metadata blocks (:573-606) → STREAMINFO + first frame byte;
//   2. k_flac_find     every byte position that starts with the 14-bit sync code 0x3FFE (:518) and carries a valid
//                      header CRC-8 is a candidate; candidates go into a list and a byte-position hash table;
//   3. k_flac_extract  one lane per candidate walks the frame exactly like decodeFrame does and writes, per subframe,
//                      the warm-up samples + Rice residuals (integers, :380-409) to a scratch array and the predictor
//                      (order, coefficients, shifts) to a descriptor — no prediction yet, so every lane of a wave runs
//                      the same bit-reading code whatever the subframe types are;
//   4. k_flac_chain    one lane per stream follows end → start links through the hash table from the first frame:
//                      precisely the frames the serial decoder visits; positions that are not in the table (CRC filter,
//                      no sync) are extracted on demand by the host loop, so the filter never changes the result;
//   5. k_flac_jobs     chained subframes → job list grouped by predictor-order class;
//   6. k_flac_restore  restoreLinearPrediction (:411-419) + wasted-bits shift (:467-469), one lane per subframe;
//   7. k_flac_finish   stereo decorrelation (:482-497) and wrap (:501-507, Q14) in place.
// Steps 3 and 6 stage everything through LDS so that global loads/stores are contiguous 128-byte runs per subframe
// (round 1 had each lane walk its own bytes from global memory: 48 + 36 ms per 3.6 GB batch, both latency-bound).
// Rows are int32 when the bit depth is ≤ 24 and no value overflows (checked everywhere; a flagged batch is redone with
// double rows and the Lua's own floating-point prediction, so even absurd values round the way the reference rounds them).  The division by 2^depth (:505) happens in the consumer.
#include <algorithm>
#include <chrono>
#include <type_traits>
#include "resample.h"
#include "stream_tail.h"
#include "flac_dev.h"
#include "resample_dev.h"

namespace aukit {

int audio_from_int_rows(aukit_ctx *ctx, int src_kind, const void *rows_dev, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len,
                        uint32_t n, int channels, double rate, double new_rate, int interp, bool do_resample, int dtype, double norm_pos,
                        double norm_neg, aukit_audio **out);


#ifndef AUKIT_FLAC_WN
#define AUKIT_FLAC_WN 16
#endif
constexpr int WN = AUKIT_FLAC_WN;  // 64-bit words of bit-stream window per lane in LDS (16: 128 bytes); 8 or 16
constexpr int LPW = WN / 2;        // lanes that refill one window with 16 bytes each
constexpr int WSTR = 2 * WN + 1;  // row stride of the window array in 32-bit words (odd: lanes at the same offset hit distinct banks)
constexpr unsigned LOW_BITS = (WN - 4) * 64;  // a lane stops for a refill once it is this far into its window
constexpr int NC = 32;    // values a lane produces (extract) / restores (restore) per round
constexpr int OSTR = 33;  // row stride of the value array
#ifndef AUKIT_FLAC_TAKE_EAGER
#define AUKIT_FLAC_TAKE_EAGER 0
#endif
#ifndef AUKIT_FLAC_SPLIT_TAPS
#define AUKIT_FLAC_SPLIT_TAPS 0     // 1: k_flac_restore_fast always splits a tap into two 24-bit multiply-adds (A/B)
#endif
#ifndef AUKIT_FLAC_CAREFUL_LOOP
#define AUKIT_FLAC_CAREFUL_LOOP 0   // 1: k_flac_extract uses its end-of-data-aware field loop in every round (A/B)
#endif
#ifndef AUKIT_FLAC_NCX
#define AUKIT_FLAC_NCX 32
#endif
constexpr int NCX = AUKIT_FLAC_NCX;  // values a lane extracts per round (k_flac_extract); a power of two <= 64
constexpr int OSTRX = NCX + 1;


// MSB-first bit reader over [first, end) of the batch buffer.  Three big-endian 64-bit words are kept in registers
// (current, next, next-but-one); they are refilled from the lane's LDS window, or from global memory when a field
// reaches outside the window (long warm-up / coefficient runs, the 44-bit look-back of readUint(n >= 32)).
struct Bits {
    const u64 *w0;
    const unsigned *lw; // this lane's window row, a ring: lw[k mod 2 WN] = big-endian 32-bit word k of the stream (counted from w0), valid for the
                        // 64-bit words [win_lo, win_lo + WN); a slide of the window loads only the 16-byte lines that are new
    u64 safe_words;
    u64 first, end;     // bit offsets (relative to w0) of the BitInputStream start and of the end of the string
    u64 pos, limit;
    u64 cur, nxt, nxt2, wi, win_lo;
    int eof;
};
AUKIT_DEV u64 be64(u64 v) { return __builtin_bswap64(v); }
AUKIT_DEV u64 fetch(const Bits &b, u64 wi) {
    const u64 d = wi - b.win_lo;
    if (d < (u64)WN) { const unsigned k = (unsigned)(2 * wi) & (2 * WN - 1); return (u64)b.lw[k] << 32 | b.lw[k + 1]; }
    return wi < b.safe_words ? be64(b.w0[wi]) : 0ull;
}
AUKIT_DEV void bits_seek(Bits &b, u64 pos) {
    b.pos = pos;
    b.wi = pos >> 6;
    b.cur = fetch(b, b.wi);
    b.nxt = fetch(b, b.wi + 1);
    b.nxt2 = fetch(b, b.wi + 2);
}
AUKIT_DEV u64 peek(const Bits &b) {
    const unsigned s = (unsigned)b.pos & 63;
    return s ? (b.cur << s) | (b.nxt >> (64 - s)) : b.cur;
}
AUKIT_DEV void skip(Bits &b, unsigned n) {  // n <= 64
    b.pos += n;
    if ((b.pos >> 6) != b.wi) {
        b.wi++;
        b.cur = b.nxt;
        b.nxt = b.nxt2;
        b.nxt2 = fetch(b, b.wi + 2);
    }
}
AUKIT_DEV u64 peek_at(const Bits &b, u64 pos) {  // uncached, any position
    const u64 wi = pos >> 6;
    const unsigned s = (unsigned)pos & 63;
    const u64 hi = fetch(b, wi);
    if (s == 0) return hi;
    return (hi << s) | (fetch(b, wi + 1) >> (64 - s));
}
// BitInputStream.readUint(n)  :351-364, n <= 57.  n >= 32 keeps the stale high bits of the 44-bit buffer like the Lua does.
AUKIT_DEV long long read_uint(Bits &b, int n) {
    if (n == 0) return 0;
    if (b.pos + (u64)n > b.end) { b.eof = 1; return 0; }  // str_byte → nil
    if (n < 32) {
        const u64 v = peek(b) >> (64 - n);
        skip(b, (unsigned)n);
        return (long long)v;
    }
    const u64 after = b.pos + n;
    const int L = (int)((8 - ((after - b.first) & 7)) & 7);  // bits left in the buffer after this read
    int width = 44 - L;
    u64 s = after - (u64)width;
    if (after < b.first + (u64)width) { width = (int)(after - b.first); s = b.first; }
    const u64 v = peek_at(b, s) >> (64 - width);
    skip(b, (unsigned)n);
    return (long long)v;
}
AUKIT_DEV long long read_sint(Bits &b, int n) {  // :365-369
    long long v = read_uint(b, n);
    if (n > 0 && v >= (1ll << (n - 1))) v -= (1ll << n);
    return v;
}
AUKIT_DEV long long read_rice(Bits &b, int param) {  // :370-376
    long long val = 0;
    for (;;) {
        if (b.pos >= b.end) { b.eof = 1; return 0; }
        const u64 w = peek(b);
        const u64 avail = b.end - b.pos;
        int z = w ? __builtin_clzll(w) : 64;
        if ((u64)z >= avail) { b.eof = 1; return 0; }  // ran off the end inside the unary prefix
        if (z < 64) { val += z; skip(b, (unsigned)z + 1); break; }
        val += 64;
        skip(b, 64);
    }
    val = (val << param) + read_uint(b, param);
    return (val & 1) ? -(val >> 1) - 1 : (val >> 1);
}

// decodeFLAC header + metadata blocks  :569-606
__global__ __launch_bounds__(64) void k_flac_header(const unsigned char *src, const u64 *off, unsigned n, FlacStreamInfo *out) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const unsigned char *p = src + off[s];
    const u64 nb = off[s + 1] - off[s];
    FlacStreamInfo r{};
    r.status = 0;
    if (nb < 4) { r.status = 1; out[s] = r; return; }                                                    // nil arithmetic in intunpack
    if (!(p[0] == 0x66 && p[1] == 0x4C && p[2] == 0x61 && p[3] == 0x43)) { r.status = 2; out[s] = r; return; }  // "Invalid magic string"
    u64 pos = 4;
    bool last = false, have = false;
    while (!last) {
        if (pos + 4 > nb) { r.status = 1; out[s] = r; return; }
        const int t = p[pos];
        last = (t & 0x80) != 0;
        const int type = t & 0x7F;
        const u64 length = (u64)p[pos + 1] << 16 | (u64)p[pos + 2] << 8 | p[pos + 3];
        pos += 4;
        if (type == 0) {
            if (pos + 34 > nb) { r.status = 1; out[s] = r; return; }
            const unsigned char *q = p + pos;
            r.rate = (double)(q[10] << 8 | q[11]) * 16 + (q[12] >> 4);
            r.channels = ((q[12] >> 1) & 7) + 1;
            r.depth = (q[12] & 1) * 16 + (q[13] >> 4) + 1;
            r.nsamples = (double)((unsigned)q[14] << 24 | (unsigned)q[15] << 16 | (unsigned)q[16] << 8 | q[17]) + (double)(q[13] & 15) * 4294967296.0;
            pos += 34;
            have = true;
        } else pos += length;
    }
    if (!have) r.status = 3;                    // "Stream info metadata block absent"
    else if (r.depth % 8 != 0) r.status = 4;    // "Sample depth not supported"
    r.first_byte = pos > nb ? nb : pos;
    out[s] = r;
}

// Speed-only plausibility filter for sync candidates: the header fields a real encoder writes for this stream (reserved bits
// clear, channel assignment and sample size matching STREAMINFO) and the header CRC-8 (poly 0x07) over the bytes decodeFrame
// walks (:518-553).  The reference checks none of this, so nothing here may change results: see k_flac_chain's miss path.
AUKIT_DEV bool flac_header_plausible(const unsigned char *src, u64 p, u64 end, int channels, int depth) {
    if (p + 6 > end) return false;
    if (src[p + 1] & 2) return false;
    const unsigned b2 = src[p + 2], b3 = src[p + 3];
    const unsigned bsc = b2 >> 4, src_code = b2 & 15;
    if (bsc == 0 || src_code == 15 || (b3 & 1)) return false;
    const unsigned asgn = b3 >> 4, ssc = (b3 >> 1) & 7;
    if (asgn <= 7 ? (int)asgn != channels - 1 : (asgn > 10 || channels != 2)) return false;
    const int ssd = ssc == 1 ? 8 : ssc == 2 ? 12 : ssc == 4 ? 16 : ssc == 5 ? 20 : ssc == 6 ? 24 : ssc == 7 ? 32 : 0;
    if (ssc == 3 || (ssc != 0 && ssd != depth)) return false;
    u64 idx = p + 4;
    const unsigned t = src[idx];
    int lead = 0;
    for (int i = 7; i >= 0; i--) { if (!(t & (1u << i))) break; lead++; }
    idx += 1 + (lead > 1 ? lead - 1 : 0);
    if (bsc == 6) idx += 1; else if (bsc == 7) idx += 2;
    if (src_code == 12) idx += 1; else if (src_code == 13 || src_code == 14) idx += 2;
    if (idx >= end) return false;
    unsigned crc = 0;
    for (u64 q = p; q < idx; q++) {
        crc ^= src[q];
        for (int k = 0; k < 8; k++) crc = (crc & 0x80) ? ((crc << 1) ^ 0x07) & 0xFF : (crc << 1) & 0xFF;
    }
    return crc == src[idx];
}


// One 16-byte vector (+ the first byte of the next one) per thread and step: 16 byte positions are tested for the sync code
// (round 1 read two bytes per thread and launched 4 M workgroups: 6 ms for 3.6 GB; this is one pass at HBM speed).
__global__ __launch_bounds__(256) void k_flac_find(const FlacGlobals G, u64 total, unsigned n, Cand *cands, u64 cap, u64 *count, CandHash H) {
    // candidates are collected per workgroup in LDS and published with ONE atomic on the global counter: with an atomicAdd per
    // candidate (220 000 of them on one word, ≈ 88 per µs: MI355X_MICROARCH.md, dequeue) the counter alone cost 2.5 of the kernel's 2.8 ms
    constexpr unsigned LMAX = 1024, RMAX = 1024;
    __shared__ Cand s_list[LMAX];
    __shared__ u64 s_raw[RMAX];
    __shared__ unsigned s_n, s_nraw;
    __shared__ u64 s_base;
    if (threadIdx.x == 0) { s_n = 0; s_nraw = 0; }
    __syncthreads();
    const u64 base_off = G.base_bit >> 3;  // G.src = (bytes of G.w0) + base_off
    auto examine = [&](u64 pa) {   // a position that holds the sync code: is it a plausible frame start of a stream?
        if (pa < base_off || pa - base_off + 1 >= total) return;
        const u64 p = pa - base_off;  // byte position in the batch
        unsigned lo = 0, hi = n;      // stream: off[s] <= p < off[s + 1]
        while (hi - lo > 1) { const unsigned m = (lo + hi) >> 1; if (G.off[m] <= p) lo = m; else hi = m; }
        const unsigned s = lo;
        const FlacStreamInfo si = G.info[s];
        const u64 b1 = G.off[s + 1];
        if (si.status || p < G.off[s] + si.first_byte || p + 1 >= b1) return;
        // Speed-only filter: a frame that fails it is still decoded — the chain asks for any position that is not in the table.
        if (!flac_header_plausible(G.src, p, b1, si.channels, si.depth)) return;
        const unsigned kl = atomicAdd(&s_n, 1u);
        if (kl < LMAX) s_list[kl] = Cand{s, 0, p};
        else {  // the workgroup's list is full: publish this one directly
            const u64 k = atomicAdd(count, 1ull);
            if (k < cap) { cands[k] = Cand{s, 0, p}; hash_insert(H, p, (unsigned)k); }
        }
    };
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(G.w0);
    const u64 nvec = G.safe_words >> 1;
    // four vectors per thread and trip, all loads issued before the first is looked at: with one load per trip the kernel had
    // 16 KiB in flight per CU and ran at 1.3 TB/s (latency-bound, not VALU-bound: the SWAR test below changed nothing by itself)
    constexpr int UN = 4;
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 c0 = (u64)blockIdx.x * 256 + threadIdx.x; c0 < nvec; c0 += stride * UN) {
      uint4 vv[UN];
      unsigned nn[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) {
          const u64 c = c0 + stride * u;
          vv[u] = make_uint4(0, 0, 0, 0);
          nn[u] = 0;
          if (c < nvec) {
              vv[u] = *reinterpret_cast<const uint4 *>(wb + 16 * c);
              if (c + 1 < nvec) nn[u] = *reinterpret_cast<const unsigned *>(wb + 16 * c + 16);
          }
      }
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const u64 c = c0 + stride * u;
        if (c >= nvec) break;
        const uint4 v = vv[u];
        const unsigned nxt = nn[u];
        const unsigned w[5] = {v.x, v.y, v.z, v.w, nxt};
        // byte 0 = 0xFF, byte 1 = 0xF8 / 0xF9 (0x3FFE :518, reserved bit clear), four positions per dword at once: z has a zero byte
        // exactly where both hold, and the carry-free zero-byte test marks those bytes (round 1 tested the 16 positions one by one
        // with 64-bit shifts: 2.8 ms for 3.6 GB, VALU-bound)
        unsigned hits = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned t = w[q], u = __builtin_amdgcn_alignbit(w[q + 1], w[q], 8);  // u: every byte's successor
            const unsigned z = ~t | ((u & 0xFEFEFEFEu) ^ 0xF8F8F8F8u);
            const unsigned zb = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);  // 0x80 in every zero byte of z, nowhere else
            if (zb) hits |= (((zb >> 7) & 1u) | ((zb >> 14) & 2u) | ((zb >> 21) & 4u) | ((zb >> 28) & 8u)) << (4 * q);
        }
        // A hit is only noted here (round 3): what follows it — a binary search over the stream offsets, the header's plausibility test with
        // its CRC-8 loop: a dozen dependent loads — ran inside the streaming loop, on one or two lanes of a wave whose other lanes waited; a third
        // of the wave-trips met one (0xFF 0xF8 turns up every 32 KiB of noise, a real frame every 14 KiB).  The workgroup examines its hits
        // afterwards, one per thread, all at once.
        while (hits) {
            const int i = __builtin_ctz(hits);
            hits &= hits - 1;
            const u64 pa = 16 * c + (u64)i;
            const unsigned kr = atomicAdd(&s_nraw, 1u);
            if (kr < RMAX) s_raw[kr] = pa;
            else examine(pa);   // (the list is full: on the spot, as before)
        }
      }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < min(s_nraw, RMAX); i += 256) examine(s_raw[i]);
    __syncthreads();
    const unsigned nl = min(s_n, LMAX);
    if (threadIdx.x == 0) s_base = nl ? atomicAdd(count, (u64)nl) : 0ull;
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nl; i += 256) {
        const u64 k = s_base + i;
        if (k < cap) { cands[k] = s_list[i]; hash_insert(H, s_list[i].byte, (unsigned)k); }
    }
}
__global__ __launch_bounds__(64) void k_flac_hash_insert(const Cand *cands, unsigned first, unsigned count, CandHash H) {
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    if (i < count) hash_insert(H, cands[first + i].byte, first + i);
}


enum { ST_FRAME = 0, ST_SUB, ST_WARM, ST_CONST, ST_COEF, ST_PART, ST_CODES, ST_SUBEND, ST_FRAMEEND, ST_DIRECT_WAIT, ST_DONE };

template <typename R> struct ExtractArgs {
    FlacGlobals G;
    const Cand *cands;
    unsigned first, count;  // candidates [first, first + count)
    CandInfo *ci;
    SubDesc *sd;
    int C;                  // channels of the batch = descriptors per candidate
    R *scratch;
    u64 scratch_cap;        // elements
    u64 *scratch_cursor;
    unsigned *flags;
    int limit_factor;
    unsigned *ticket;       // next candidate (relative to first) nobody has taken yet
};

// decodeFrame (:510-557) without prediction: header, per-subframe warm-up / residual values → scratch, predictor → SubDesc.
#ifdef AUKIT_FLAC_EXTRACT_WAVES
#define AUKIT_EXTRACT_OCC __attribute__((amdgpu_waves_per_eu(AUKIT_FLAC_EXTRACT_WAVES, 8)))
#else
#define AUKIT_EXTRACT_OCC
#endif
#ifndef AUKIT_FLAC_EXTRACT_WGS
#define AUKIT_FLAC_EXTRACT_WGS 8   // workgroups per CU in k_flac_extract's persistent grid
#endif
#ifndef AUKIT_FLAC_NOPF
#define AUKIT_FLAC_NOPF 0   // 1: no register prefetch of the window lines (48 VGPRs less: A/B for a third wave per SIMD)
#endif
template <typename R>
__global__ __launch_bounds__(64) AUKIT_EXTRACT_OCC void k_flac_extract(const ExtractArgs<R> A) {
    __shared__ unsigned s_win[64 * WSTR];
    __shared__ R s_out[64 * OSTRX];
    __shared__ u64 s_ptr[64];
    __shared__ int s_cnt[64];
    const int lane = threadIdx.x;
    const int C = A.C;
    // A lane owns one candidate at a time and takes the next one off a ticket counter when its frame is done (`take` below, at round
    // boundaries): frames differ a lot in length — a VERBATIM subframe has four times the bits of a Rice-coded one — and with a fixed
    // candidate per lane a wave lived as long as its longest frame, most of its lanes idle for the second half; the grid is sized to what
    // the chip holds at once, so there is no thinly populated second generation of workgroups either.
    bool have = false;
    unsigned idx = 0;
    Cand c{};
    FlacStreamInfo si{};
    int depth = 0;
    Bits b;
    b.w0 = A.G.w0;
    b.lw = s_win + lane * WSTR;
    b.safe_words = A.G.safe_words;
    b.first = b.end = b.pos = b.wi = 0;
    b.eof = 0;
    b.limit = ~0ull;
    b.win_lo = 0; b.cur = b.nxt = b.nxt2 = 0;
    int st = ST_DONE;
    bool fresh = true;
    int status = FE_OK, bs = 0, chan_asgn = 0, nsub = 0, ch = 0;
    int order = 0, type = 0, wasted = 0, sdepth = 0, lshift = 0, after = ST_SUBEND, resume = ST_SUB;
    int nparts = 0, psize = 0, pi = 0, param = 0, nbits = 0, param_bits = 4, escape = 15, remaining = 0, jpos = 0;
    bool esc = false, direct = false, store_ok = false, ovf = false, gen_once = false, gen_part = false;
    long long cval = 0;
    u64 cand_scratch = 0, gcur = 0, end_byte = 0;
    SubDesc *sd = A.sd;
    R *const orow = s_out + lane * OSTRX;
    auto start = [&](unsigned rel) {   // this lane's next candidate: everything a frame's decoding keeps between rounds, as at kernel entry
        have = true;
        idx = A.first + rel;
        c = A.cands[idx];
        si = A.G.info[c.stream];
        depth = si.depth;
        b.first = A.G.base_bit + 8 * (A.G.off[c.stream] + si.first_byte);
        b.end = A.G.base_bit + 8 * A.G.off[c.stream + 1];
        b.eof = 0;
        b.limit = ~0ull;
        b.pos = A.G.base_bit + 8 * c.byte;
        b.wi = b.pos >> 6;
        b.win_lo = 0; b.cur = b.nxt = b.nxt2 = 0;
        st = ST_FRAME;
        fresh = true;
        status = FE_OK; bs = 0; chan_asgn = 0; nsub = 0; ch = 0;
        order = 0; type = 0; wasted = 0; sdepth = 0; lshift = 0; after = ST_SUBEND; resume = ST_SUB;
        nparts = 0; psize = 0; pi = 0; param = 0; nbits = 0; param_bits = 4; escape = 15; remaining = 0; jpos = 0;
        esc = false; direct = false; store_ok = false; ovf = false; gen_once = false; gen_part = false;
        cval = 0;
        cand_scratch = 0; gcur = 0; end_byte = 0;
        sd = A.sd + (size_t)idx * C;
    };
    auto finish = [&]() {              // the frame is done (or failed): what the chain walk needs to know about it
        CandInfo f;
        f.end_byte = end_byte;
        f.scratch = cand_scratch;
        f.sample_off = 0;
        f.blocksize = bs; f.chan_asgn = chan_asgn; f.status = status; f.nsub = nsub;
        f.seq = 0; f.used = 0;
        A.ci[idx] = f;
        if (ovf && status == FE_OK) atomicOr(A.flags, (unsigned)FLAG_OVERFLOW);
        have = false;
    };
    auto take = [&]() {                // every lane without a frame takes a ticket: one atomic per wave
        const u64 m = __ballot(!have);
        if (!m) return;
        unsigned base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(A.ticket, (unsigned)__builtin_popcountll(m));
        base = __shfl(base, __builtin_ctzll(m));
        const unsigned rel = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1));
        if (!have && rel < A.count) start(rel);
    };
    take();
    // the values of stream s go to 128 contiguous bytes; two streams per store instruction.  A round's values are stored at the top of the NEXT
    // round, behind the wait for that round's window lines: loads and stores share one counter (vmcnt), so stores issued in front of that wait
    // would be waited for as well; issued behind it they drain while the next round is decoded.
    auto flush = [&]() {
        // the usual round — every lane of the wave produced NCX values (or none) and writes them to a 16-byte aligned place — goes out as
        // 16 bytes per lane, 64 / (NCX / 4) streams per store instruction
        if (sizeof(R) == 4 && __all((s_cnt[lane] == NCX && (s_ptr[lane] & 3) == 0) || s_cnt[lane] == 0)) {
            constexpr int LPS = NCX / 4, SPI = 64 / LPS;   // lanes per stream, streams per instruction
            const int grp = lane / LPS, q4 = 4 * (lane % LPS);
#pragma unroll
            for (int i = 0; i < LPS; i++) {
                const int s = SPI * i + grp;
                if (s_cnt[s]) {
                    const R *v = s_out + s * OSTRX + q4;
                    *reinterpret_cast<uint4 *>(A.scratch + s_ptr[s] + q4) = make_uint4((unsigned)v[0], (unsigned)v[1], (unsigned)v[2], (unsigned)v[3]);
                }
            }
            return;
        }
        constexpr int PER = 64 / NCX;  // streams per store instruction
        const int part = lane / NCX, k = lane % NCX;
        for (int i = 0; i < NCX; i++) {
            const int s = PER * i + part;
            if (k < s_cnt[s]) A.scratch[s_ptr[s] + k] = s_out[s * OSTRX + k];
        }
    };
    bool have_flush = false;
    uint4 pf[LPW];      // per stream this lane loads for: the line that will replace its slot's line, already requested
    u64 pf_line[LPW];
#pragma unroll
    for (int i = 0; i < LPW; i++) { pf[i] = make_uint4(0, 0, 0, 0); pf_line[i] = ~0ull; }

    for (;;) {
        // ---- slide the LDS windows of the lanes that have used a quarter of theirs: 8 lanes × 16 bytes per stream, 8 streams per load
        {
            const bool want = st != ST_DONE && (fresh || (b.wi - b.win_lo) >= (u64)(WN / 4));
            const u64 new_lo = b.wi & ~1ull;
            // lines (16 bytes = two 64-bit words) of the new window that the old one did not hold: all of them for a fresh lane or a jump
            const u64 keep_from = fresh ? ~0ull : b.win_lo / 2 + WN / 2;   // first line index the ring does not hold yet
            if (want) b.win_lo = new_lo;
            const int sub8 = lane % LPW, grp = lane / LPW;
#pragma unroll
            for (int i = 0; i < LPW; i++) {
                const int s = i * (64 / LPW) + grp;
                const int w = __shfl((int)want, s);
                const u64 ws = __shfl(new_lo, s);
                const u64 kf = __shfl(keep_from, s);
                if (w) {
                    // ring slot sub8 holds the line of the new window that is congruent to sub8 mod LPW
                    const u64 l0 = ws / 2, line = l0 + (((u64)sub8 - l0) & (u64)(LPW - 1));
                    if (kf == ~0ull || line >= kf || line < kf - WN / 2) {
                        // the line that follows a slot's line into the slot (eight lines on) was requested when that one arrived: a round ago or
                        // more, so it is here — unless the lane is new or jumped, then it is fetched now
                        uint4 v = pf[i];
                        if (AUKIT_FLAC_NOPF || pf_line[i] != line) {
                            v = make_uint4(0, 0, 0, 0);
                            if (2 * line < A.G.safe_words) v = *reinterpret_cast<const uint4 *>(A.G.w0 + 2 * line);
                        }
                        unsigned *wrow = s_win + s * WSTR + 4 * sub8;
                        wrow[0] = __builtin_bswap32(v.x); wrow[1] = __builtin_bswap32(v.y);
                        wrow[2] = __builtin_bswap32(v.z); wrow[3] = __builtin_bswap32(v.w);
                        if (!AUKIT_FLAC_NOPF) {
                            const u64 nl = line + LPW;
                            pf_line[i] = nl;
                            pf[i] = make_uint4(0, 0, 0, 0);
                            if (2 * nl < A.G.safe_words) pf[i] = *reinterpret_cast<const uint4 *>(A.G.w0 + 2 * nl);
                        }
                    }
                }
            }
            if (have_flush) flush();
            __syncthreads();
            if (fresh && st != ST_DONE) { bits_seek(b, b.pos); fresh = false; }
        }
        // ---- every lane advances its own frame until it has produced NCX values or its window runs low
        int cnt = 0;
        // a round that starts at an odd place produces as many values less as bring the next one back to a multiple of four: the 16-byte
        // flush wants aligned rows, and one short round (first of a frame, after a run of long codes) would otherwise spoil every later one
        const int lim = NCX - (int)(gcur & 3);
        if (st == ST_DIRECT_WAIT) { __threadfence(); direct = true; st = resume; }
        while (st != ST_DONE && cnt < lim && (b.wi - b.win_lo) < (u64)(WN - 4)) {
            if (!direct && !gen_once && !gen_part && (st == ST_CODES || st == ST_PART || (st == ST_WARM && sdepth < 32))) {
                // ---- fast path for everything that is a run of bit fields: Rice partitions incl. their headers and escape-coded
                // partitions (:393-407), warm-up samples and VERBATIM subframes (:422-424, :456-458).  A 64-bit shift register is fed
                // from the LDS window one 32-bit word ahead; one loop, one exit condition, selects instead of branches (hipcc's
                // exec-mask bookkeeping for a loop with many exits cost more than the decoding).  One iteration = one field.  The
                // rare cases (Rice parameter > 26, a residual longer than 32 bits) stop the loop and go to the generic reader.
                const u64 wbase = b.win_lo << 6;
                unsigned rp = (unsigned)(b.pos - wbase);
                const unsigned end_rel = (unsigned)min(b.end - wbase, (u64)1 << 30);
                const unsigned limit_rel = (unsigned)min(b.limit > wbase ? b.limit - wbase : 0ull, (u64)1 << 30);
                unsigned wd = (rp >> 5) + 2;  // next 32-bit word of the window to shift in
                const unsigned ring0 = (unsigned)(2 * b.win_lo);  // absolute index of the window's first 32-bit word (the ring index is that mod 2 WN)
                u64 buf = ((u64)b.lw[(ring0 + wd - 2) & (2 * WN - 1)] << 32 | b.lw[(ring0 + wd - 1) & (2 * WN - 1)]) << (rp & 31);
                int avail = 64 - (int)(rp & 31);  // valid bits at the top of buf; >= 32 at the top of every iteration
                unsigned wnext = b.lw[(ring0 + wd) & (2 * WN - 1)];  // loaded one iteration ahead: LDS latency stays off the dependency chain (past the window's end: stale words nobody uses)
                const bool warm = st == ST_WARM;
                int why = 0;  // 1: hand over to the generic reader, 2: ran off the end of the data, 3: over the bit budget
                bool go = true;
                // The window does not reach the end of the data for any lane of the wave (all rounds but a stream's last few): a leaner copy
                // of the loop below — no end-of-data test, two window words in registers and a third one on its way instead of a 64-bit shift
                // register, mode bits as integers, one rare exit in front of the store.  Same fields, same state afterwards.
                const bool near_end = end_rel < (unsigned)(2 * WN * 32 + 64);
                if (!__any(near_end) && !AUKIT_FLAC_CAREFUL_LOOP) {
                    unsigned k = rp >> 5, sb = rp & 31u;          // word of the window and bit within it
                    unsigned w0 = b.lw[(ring0 + k) & (2 * WN - 1)], w1 = b.lw[(ring0 + k + 1) & (2 * WN - 1)];
                    unsigned wn = wnext;                          // word k + 2
                    wd = k + 2;
                    int fixed = (warm || esc) ? 1 : 0, nfix = warm ? sdepth : nbits;
                    const int count_first = psize - min(order, psize), count_rest = psize, pbs = 32 - param_bits;
                    const int warm_i = warm ? 1 : 0;
                    for (;;) {
                        const unsigned hi = (unsigned)((((u64)w0 << 32) | w1) << sb >> 32);
                        const int hdr = (remaining == 0) & (warm_i ^ 1);
                        const int z = __builtin_clz(hi | 1u);
                        const int tot_r = z + 1 + param;
                        const unsigned ur = ((unsigned)z << param) | ((((hi << z) << 1) >> 1) >> (31 - param));
                        const int v_rice = (int)(ur >> 1) ^ -(int)(ur & 1u);
                        const int v_fix = nfix ? ((int)hi >> ((32 - nfix) & 31)) : 0;
                        const int pv = (int)(hi >> pbs);
                        const int is_esc = pv >= escape ? 1 : 0;
                        const int total = hdr ? param_bits + 5 * is_esc : (fixed ? nfix : tot_r);
                        const unsigned nrp = rp + (unsigned)total;
                        const int bad_h = ((is_esc ^ 1) & (pv > 26)) | ((nrp > limit_rel) ? 2 : 0);   // 1: generic reader, 2/3: over the bit budget
                        const int bad_v = (fixed ^ 1) & ((hi == 0u) | (tot_r > 32));
                        const int bad = hdr ? bad_h : bad_v;
                        if (bad) { why = hdr ? ((bad_h & 1) ? 1 : 3) : 1; break; }
                        orow[hdr ? NCX : cnt] = (R)(fixed ? v_fix : v_rice);   // (slot NCX: nobody's)
                        const unsigned s2 = sb + (unsigned)total;
                        const bool cross = s2 >= 32u;
                        sb = s2 & 31u;
                        w0 = cross ? w1 : w0;
                        w1 = cross ? wn : w1;
                        wd += cross ? 1u : 0u;
                        wn = b.lw[(ring0 + wd) & (2 * WN - 1)];
                        rp = nrp;
                        const int nb5 = (int)((hi << param_bits) >> 27);
                        param = hdr ? pv : param;
                        fixed = hdr ? is_esc : fixed;
                        nfix = hdr ? (is_esc ? nb5 : 0) : nfix;
                        remaining = hdr ? (pi == 0 ? count_first : count_rest) : remaining - 1;
                        cnt += hdr ^ 1;
                        pi += (remaining == 0) & (warm_i ^ 1);   // a finished (or empty) partition
                        if (!((cnt < lim) & (rp < LOW_BITS) & ((remaining > 0) | ((warm_i ^ 1) & (pi < nparts))))) break;
                    }
                    if (!warm) { esc = fixed != 0; nbits = nfix; }
                }
                else
                while (go) {
                    const bool hdr = !warm & (remaining == 0);
                    const bool fixed = warm | esc;
                    const int nfix = warm ? sdepth : nbits;
                    const unsigned hi = (unsigned)(buf >> 32);
                    const int z = __builtin_clz(hi | 1u);
                    // partition header: Rice parameter, or the escape code followed by a 5-bit field width
                    const int pv = (int)(hi >> (32 - param_bits));
                    const bool is_esc = pv >= escape;
                    const int nb5 = (int)((hi << param_bits) >> 27);
                    const int total = hdr ? param_bits + (is_esc ? 5 : 0) : (fixed ? nfix : z + 1 + param);
                    const unsigned nrp = rp + (unsigned)total;
                    const bool b_gen = hdr ? (!is_esc & (pv > 26)) : (!fixed & ((hi == 0) | (total > 32)));
                    const bool b_eof = nrp > end_rel, b_lim = hdr & (nrp > limit_rel);
                    const int bad = b_gen ? 1 : (b_eof ? 2 : (b_lim ? 3 : 0));
                    const bool good = bad == 0;
                    why = bad;
                    // the value: `param` bits after the unary prefix (zig-zag), or a sign-extended nfix-bit field
                    const unsigned low = ((((hi << z) << 1) >> 1) >> (31 - param));
                    const unsigned u = ((unsigned)z << param) | low;
                    const int v_rice = (int)(u >> 1) ^ -(int)(u & 1);
                    const int v_fix = nfix ? ((int)hi >> (32 - nfix)) : 0;
                    orow[min(cnt, NCX - 1)] = (R)(fixed ? v_fix : v_rice);
                    const int tot = good ? total : 0;
                    buf <<= tot;
                    avail -= tot;
                    rp += (unsigned)tot;
                    const bool need = avail < 32;
                    buf |= need ? (u64)wnext << (32 - avail) : 0ull;
                    avail += need ? 32 : 0;
                    wd += need ? 1u : 0u;
                    wnext = b.lw[(ring0 + wd) & (2 * WN - 1)];
                    // a header sets the coding of partition pi and its number of residuals (:394-395, :403); a value is stored
                    const int count = psize - (pi == 0 ? min(order, psize) : 0);
                    const bool gh = good & hdr;
                    cnt += (good & !hdr) ? 1 : 0;
                    param = gh ? pv : param;
                    esc = gh ? is_esc : esc;
                    nbits = gh ? (is_esc ? nb5 : 0) : nbits;
                    remaining = good ? (hdr ? count : remaining - 1) : remaining;
                    pi += (good & !warm & (remaining == 0)) ? 1 : 0;  // a finished (or empty) partition
                    go = good & (cnt < lim) & (rp < LOW_BITS) & ((remaining > 0) | (!warm & (pi < nparts)));
                }
                bits_seek(b, wbase + rp);
                if (why == 1) gen_once = true;
                else if (why == 2) { b.eof = 1; status = FE_NIL; st = ST_DONE; }
                else if (why == 3 || (warm && rp > limit_rel)) { status = FE_LIMIT; st = ST_DONE; }
                if (st != ST_DONE) {
                    if (warm) st = remaining == 0 ? after : ST_WARM;
                    else st = remaining > 0 ? ST_CODES : (pi < nparts ? ST_PART : ST_SUBEND);
                }
            } else if (st == ST_CODES) {
                while (remaining > 0 && cnt < lim && (b.wi - b.win_lo) < (u64)(WN - 4)) {
                    const long long v = esc ? read_sint(b, nbits) : read_rice(b, param);
                    if constexpr (sizeof(R) == 4) { if (v != (long long)(R)v) ovf = true; }
                    if (direct) { if (store_ok) A.scratch[cand_scratch + (u64)ch * bs + jpos] = (R)v; }
                    else orow[cnt++] = (R)v;
                    jpos++;
                    remaining--;
                    if (gen_once) { gen_once = false; break; }
                }
                if (b.eof) { status = FE_NIL; st = ST_DONE; }
                else if (remaining == 0) { pi++; gen_part = false; st = pi < nparts ? ST_PART : ST_SUBEND; }
            } else if (st == ST_WARM) {  // warm-up samples (:422-424, :430-432) or a VERBATIM subframe (:456-458)
                while (remaining > 0 && cnt < lim && (b.wi - b.win_lo) < (u64)(WN - 4)) {
                    const long long v = read_sint(b, sdepth);
                    if constexpr (sizeof(R) == 4) { if (v != (long long)(R)v) ovf = true; }
                    if (direct) { if (store_ok && jpos < bs) A.scratch[cand_scratch + (u64)ch * bs + jpos] = (R)v; }
                    else orow[cnt++] = (R)v;
                    jpos++;
                    remaining--;
                    if ((jpos & 255) == 0 && b.pos > b.limit) { status = FE_LIMIT; st = ST_DONE; break; }
                }
                if (st != ST_DONE) {
                    if (b.eof) { status = FE_NIL; st = ST_DONE; }
                    else if (remaining == 0) st = after;
                }
            } else if (st == ST_CONST) {  // :453-454
                if constexpr (sizeof(R) == 4) { if (cval != (long long)(R)cval) ovf = true; }
                while (remaining > 0 && cnt < lim) { orow[cnt++] = (R)cval; remaining--; }
                if (remaining == 0) st = ST_SUBEND;
            } else if (st == ST_PART) {  // :394-406
                gen_once = false;
                gen_part = false;
                param = (int)read_uint(b, param_bits);
                esc = param >= escape;
                nbits = 0;
                if (esc) nbits = (int)read_uint(b, 5);
                if (b.eof) { status = FE_NIL; st = ST_DONE; }
                else if (b.pos > b.limit) { status = FE_LIMIT; st = ST_DONE; }
                else {
                    const int start = pi * psize + (pi == 0 ? order : 0), endd = (pi + 1) * psize;
                    remaining = endd > start ? endd - start : 0;
                    jpos = start;
                    gen_part = !esc && param > 26;
                    if (remaining > 0) st = ST_CODES;
                    else { pi++; st = pi < nparts ? ST_PART : ST_SUBEND; }
                }
            } else if (st == ST_SUB) {  // decodeSubframe  :443-465
                read_uint(b, 1);
                type = (int)read_uint(b, 6);
                wasted = (int)read_uint(b, 1);
                if (wasted == 1) {  // unary wasted-bits count  :447-449
                    for (;;) {
                        const long long bit = read_uint(b, 1);
                        if (b.eof || bit) break;
                        wasted++;
                    }
                }
                sdepth = depth - wasted;
                if (chan_asgn >= 8) sdepth += ((chan_asgn == 9) == (ch == 0)) ? 1 : 0;  // the side channel has one more bit  :483-497
                direct = false;
                order = 0; lshift = 0; jpos = 0;
                if (b.eof || sdepth < 0 || sdepth > 57) { status = FE_NIL; st = ST_DONE; }  // 2^(n-1) with a negative n misbehaves in the Lua too
                else if (type == 0) {
                    cval = read_sint(b, sdepth);
                    if (b.eof) { status = FE_NIL; st = ST_DONE; }
                    else { remaining = bs; st = ST_CONST; }
                } else if (type == 1) { remaining = bs; after = ST_SUBEND; st = ST_WARM; }
                else if ((type >= 8 && type <= 12) || (type >= 32 && type <= 63)) {
                    order = type <= 12 ? type - 8 : type - 31;
                    remaining = order;
                    after = ST_COEF;
                    st = order > 0 ? ST_WARM : ST_COEF;
                    // more warm-up samples than the block holds: the Lua table simply grows past blockSize; take the explicit-index path
                    if (order > bs) { resume = st; st = ST_DIRECT_WAIT; break; }
                } else { status = FE_SUBTYPE; st = ST_DONE; }
            } else if (st == ST_COEF) {  // :433-438 / FIXED_PREDICTION_COEFFICIENTS :334-340, then the residual header :381-391
                if (type >= 32) {
                    const int precision = (int)read_uint(b, 4) + 1;
                    lshift = (int)read_sint(b, 5);
                    for (int i = 0; i < order; i++) sd[ch].coef[i] = (short)read_sint(b, precision);
                } else {
                    const short fc[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
                    for (int i = 0; i < order; i++) sd[ch].coef[i] = fc[order][i];
                }
                const int method = (int)read_uint(b, 2);
                param_bits = method == 0 ? 4 : 5;
                escape = method == 0 ? 0xF : 0x1F;
                const int porder = (int)read_uint(b, 4);
                nparts = 1 << porder;
                if (b.eof) { status = FE_NIL; st = ST_DONE; }
                else if (method >= 2) { status = FE_RESMETHOD; st = ST_DONE; }
                else if (bs % nparts != 0) { status = FE_PARTITION; st = ST_DONE; }
                else {
                    psize = bs / nparts;
                    pi = 0;
                    st = ST_PART;
                    // a partition smaller than the predictor order makes later partitions overwrite warm-up entries (:400): explicit-index path
                    if (!direct && nparts > 1 && psize < order) { resume = ST_PART; st = ST_DIRECT_WAIT; break; }
                }
            } else if (st == ST_SUBEND) {
                sd[ch].order = order;
                sd[ch].lshift = lshift;
                sd[ch].wasted = wasted;
                sd[ch].kind = order == 0 ? 0 : (order <= 4 ? 1 : (order <= 12 ? 2 : 3));
                if (direct) { gcur = cand_scratch + (u64)(ch + 1) * bs; direct = false; }  // cnt == 0 here: direct mode starts at a round boundary
                ch++;
                st = ch < nsub ? ST_SUB : ST_FRAMEEND;
            } else if (st == ST_FRAME) {  // decodeFrame header  :510-553
                const long long t0 = read_uint(b, 8);
                if (b.eof) { status = FE_EOF_START; st = ST_DONE; continue; }
                const long long sync = t0 * 64 + read_uint(b, 6);
                if (b.eof) { status = FE_NIL; st = ST_DONE; continue; }
                if (sync != 0x3FFE) { status = FE_SYNC; st = ST_DONE; continue; }
                read_uint(b, 2);
                const int bsc = (int)read_uint(b, 4), src_code = (int)read_uint(b, 4);
                chan_asgn = (int)read_uint(b, 4);
                read_uint(b, 4);
                const int t = (int)read_uint(b, 8);
                if (b.eof) { status = FE_NIL; st = ST_DONE; continue; }
                int t2 = -1;
                for (int i = 7; i >= 0; i--) { if (!(t & (1 << i))) break; t2++; }
                for (int i = 1; i <= t2; i++) read_uint(b, 8);
                if (bsc == 1) bs = 192;
                else if (bsc >= 2 && bsc <= 5) bs = 576 << (bsc - 2);
                else if (bsc == 6) bs = (int)read_uint(b, 8) + 1;
                else if (bsc == 7) bs = (int)read_uint(b, 16) + 1;
                else if (bsc >= 8) bs = 256 << (bsc - 8);
                else { status = FE_BLOCKSIZE; st = ST_DONE; continue; }
                if (src_code == 12) read_uint(b, 8);
                else if (src_code == 13 || src_code == 14) read_uint(b, 16);
                read_uint(b, 8);  // CRC-8, ignored :553
                if (b.eof) { status = FE_NIL; st = ST_DONE; continue; }
                if (chan_asgn <= 7) nsub = C;
                else if (chan_asgn <= 10) {
                    nsub = 2;
                    if (C != 2) { status = FE_NIL; st = ST_DONE; continue; }  // result[ch] of a missing / extra channel is nil (:482-507)
                } else { status = FE_CHAN; st = ST_DONE; continue; }
                // bit budget: `limit_factor` quarters of the size of an all-VERBATIM frame (no encoder emits a bigger one); a false
                // candidate crawling through garbage on one lane would otherwise hold its whole wave for tens of milliseconds
                b.limit = (A.limit_factor > 0 && !c.nolimit) ? b.pos + (u64)A.limit_factor * (u64)bs * (u64)C * (u64)(depth + 2) / 4 + 4096 : ~0ull;
                const u64 need = (u64)nsub * (u64)bs;
                // + 32 values: block sizes are powers of two times the channel count, and without the skew every candidate's values would start at the
                // same offset within 16 KiB — the 64 lanes of a wave here, and of a wave of k_flac_restore, would then ask the same few memory
                // channels for their 128 bytes each (measured: the restore kernels moved 2.8 TB/s that way)
                cand_scratch = atomicAdd(A.scratch_cursor, ((need + 3) & ~3ull) + 32);
                store_ok = cand_scratch + need <= A.scratch_cap;
                gcur = cand_scratch;
                ch = 0;
                st = ST_SUB;
            } else if (st == ST_FRAMEEND) {  // :555-557
                b.pos = b.first + (((b.pos - b.first) + 7) & ~7ull);  // alignToByte
                // readUint(16): a nil here is discarded, the NEXT readByte then returns nil  :557
                b.pos = (b.pos + 16 <= b.end) ? b.pos + 16 : b.end;
                end_byte = (b.pos - A.G.base_bit) >> 3;
                st = ST_DONE;
            } else break;  // ST_DIRECT_WAIT: resume after this round's flush
        }
        // ---- this round's values: where they go (stored by the next round's flush, or right here after the last round)
        s_cnt[lane] = store_ok ? cnt : 0;
        s_ptr[lane] = gcur;
        gcur += (u64)cnt;
        __syncthreads();
        have_flush = true;
        if (have && st == ST_DONE) finish();
        // new frames when the whole wave is through with its old ones: the lanes then parse their frame and subframe headers (the slow,
        // generic part) in the same rounds — block sizes rarely differ within a batch; taking a frame per lane as soon as it is free
        // measured 10 % more instructions for that reason — and AUKIT_FLAC_TAKE_EAGER lets them anyway (streams of mixed block sizes)
        if (AUKIT_FLAC_TAKE_EAGER || __ballot(have) == 0) take();
        if (__ballot(st != ST_DONE) == 0) { flush(); break; }
    }
}


// decodeFLAC's frame loop (:615): follow end → start links from the first frame
__global__ __launch_bounds__(64) void k_flac_chain(const FlacGlobals G, unsigned n, CandHash H, CandInfo *ci, const SubDesc *sd, int C, ChainOut *out, u64 *kind_count) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    ChainOut r{};
    u64 at = G.off[s] + G.info[s].first_byte;
    const u64 endb = G.off[s + 1];
    u64 sp = 0;
    unsigned nf = 0;
    int bs0 = 0, prev_bs = 0, uniform = 1;
    unsigned kc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // jobs per (order class, channel assignment): class * 4 + (0 = independent, 1..3 = 8..10)
    int stream_status = FE_OK;
    bool go = at < endb;  // readByte() → nil → decodeFrame returns false
    while (go) {          // (single-exit loop: the multi-break form of this walk was miscompiled by hipcc 7.2 — status lost)
        const unsigned k = hash_lookup(H, at);
        if (k == ~0u) { r.miss_kind = 1; r.miss_at = at; go = false; }
        else {
            const CandInfo f = ci[k];
            if (f.status == FE_OK) {
                ci[k].sample_off = sp;
                ci[k].seq = nf;
                ci[k].used = 1;
                if (!sd) {}   // (the fused decoder, flac_fused.hip: no prediction jobs to count)
                else if (C == 2) kc[4 * max(sd[(size_t)k * 2].kind & 3, sd[(size_t)k * 2 + 1].kind & 3) + (f.chan_asgn >= 8 ? f.chan_asgn - 7 : 0)] += 2;  // a stereo frame's two jobs share a wave (k_flac_jobs)
                else for (int c = 0; c < f.nsub; c++) kc[4 * (sd[(size_t)k * C + c].kind & 3)]++;
                if (nf == 0) bs0 = f.blocksize;
                else if (prev_bs != bs0) uniform = 0;   // a frame that is not the last differs from the first
                prev_bs = f.blocksize;
                sp += (u64)f.blocksize;
                nf++;
                at = f.end_byte;
                go = at < endb;
            } else {
                if (f.status == FE_LIMIT) { r.miss_kind = 2; r.miss_ci = k; }
                else if (f.status != FE_EOF_START) stream_status = f.status;
                go = false;
            }
        }
    }
    r.L = sp;
    r.bs0 = bs0;
    r.uniform = (uniform && prev_bs <= bs0) ? 1 : 0;
    r.nframes = nf;
    r.status = stream_status;
    out[s] = r;
    for (int q = 0; q < 16; q++) if (kc[q]) atomicAdd(&kind_count[q], (u64)kc[q]);
}


// The same walk for the register decoders (round 6), one dependent load per frame instead of three: k_flac_links (a lane per candidate, all at once)
// looks every candidate's successor up — the candidate that starts where it ends — and puts what the walk wants of a frame into 16 bytes; the walk
// follows those.  (k_flac_chain waits for the hash slot, the slot's value and the frame's record, one after the other, 108 times per ten-second
// stream: 0.23 ms whatever the batch — a thirteenth of config 5's step at 256 streams.)
struct ChainLink { unsigned next; int blocksize; int status; unsigned more; };   // more: the frame ends inside its stream's data (the walk goes on)
__global__ __launch_bounds__(256) void k_flac_links(const FlacGlobals G, const Cand *cands, const CandInfo *ci, unsigned ncand, CandHash H, ChainLink *links) {
    const unsigned k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ncand) return;
    const CandInfo f = ci[k];
    ChainLink l;
    l.blocksize = f.blocksize; l.status = f.status;
    l.more = (f.status == FE_OK && f.end_byte < G.off[cands[k].stream + 1]) ? 1u : 0u;
    l.next = l.more ? hash_lookup(H, f.end_byte) : ~0u;
    links[k] = l;
}
__global__ __launch_bounds__(64) void k_flac_chain_links(const FlacGlobals G, unsigned n, CandHash H, CandInfo *ci, const ChainLink *links, ChainOut *out) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    ChainOut r{};
    const u64 at0 = G.off[s] + G.info[s].first_byte;
    u64 sp = 0;
    unsigned nf = 0;
    int bs0 = 0, prev_bs = 0, uniform = 1;
    int stream_status = FE_OK;
    bool go = at0 < G.off[s + 1];  // readByte() -> nil -> decodeFrame returns false
    unsigned k = go ? hash_lookup(H, at0) : ~0u, kprev = ~0u;
    while (go) {   // (single-exit loop, as k_flac_chain)
        if (k == ~0u) { r.miss_kind = 1; r.miss_at = kprev == ~0u ? at0 : ci[kprev].end_byte; go = false; }
        else {
            const ChainLink l = links[k];
            if (l.status == FE_OK) {
                ci[k].sample_off = sp;
                ci[k].seq = nf;
                ci[k].used = 1;
                if (nf == 0) bs0 = l.blocksize;
                else if (prev_bs != bs0) uniform = 0;
                prev_bs = l.blocksize;
                sp += (u64)l.blocksize;
                nf++;
                kprev = k;
                k = l.next;
                go = l.more != 0;
            } else {
                if (l.status == FE_LIMIT) { r.miss_kind = 2; r.miss_ci = k; }
                else if (l.status != FE_EOF_START) stream_status = l.status;
                go = false;
            }
        }
    }
    r.L = sp;
    r.bs0 = bs0;
    r.uniform = (uniform && prev_bs <= bs0) ? 1 : 0;
    r.nframes = nf;
    r.status = stream_status;
    out[s] = r;
}


__global__ __launch_bounds__(256) void k_flac_jobs(const Cand *cands, const CandInfo *ci, const SubDesc *sd, unsigned ncand, int C, const u64 *row_off,
                                                  const u64 *frame_base, const u64 *kind_base, u64 *kind_fill, SubJob *jobs, FrameRec *frames) {
    // thread t takes candidate (t mod 64) * (threads / 64) + t / 64: the 64 candidates of a wave lie far apart (other streams), and since a wave's
    // jobs get neighbouring slots, the subframes one wave of k_flac_restore works on do not sit at one offset within their rows' 16 KiB blocks
    const unsigned t = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;
    const unsigned k = (t & 63) * (nthr / 64) + t / 64;
    CandInfo f{};
    if (k < ncand) f = ci[k];
    const bool used = k < ncand && f.used;
    const unsigned s = used ? cands[k].stream : 0;
    const int lane = threadIdx.x & 63;
    if (C == 2) {
        // Stereo: both subframes of a frame go to adjacent slots (even = channel 0) of the order class of the larger predictor, so that
        // k_flac_restore can decorrelate them out of LDS — every class count is even, so slots keep their parity.
        const int kind = used ? 4 * max(sd[(size_t)k * 2].kind & 3, sd[(size_t)k * 2 + 1].kind & 3) + (f.chan_asgn >= 8 ? f.chan_asgn - 7 : 0) : -1;
        u64 todo = __ballot(kind >= 0);
        while (todo) {   // one turn per class present in the wave (a handful), one atomic each
            const int q = __shfl(kind, __builtin_ctzll(todo));
            const u64 m = __ballot(kind == q);
            todo &= ~m;
            u64 base = 0;
            if (lane == __builtin_ctzll(m)) base = atomicAdd(&kind_fill[q], 2 * (u64)__builtin_popcountll(m));
            base = __shfl(base, __builtin_ctzll(m));
            if (kind == q) {
                const u64 slot = kind_base[q] + base + 2 * (u64)__builtin_popcountll(m & ((1ull << lane) - 1));
                const int asgn = f.chan_asgn >= 8 ? f.chan_asgn : 0;
                jobs[slot] = SubJob{f.scratch, row_off[(size_t)s * 2] + f.sample_off, k * 2u, f.blocksize, asgn, 0};
                jobs[slot + 1] = SubJob{f.scratch + (u64)f.blocksize, row_off[(size_t)s * 2 + 1] + f.sample_off, k * 2u + 1, f.blocksize, asgn, 0};
            }
        }
    } else
    for (int c = 0; c < C; c++) {  // wave-uniform trip count; one atomic per wave and order class instead of one per subframe
        const bool have = used && c < f.nsub;
        const int kind = have ? 4 * (sd[(size_t)k * C + c].kind & 3) : -1;
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            const u64 m = __ballot(kind == q);
            if (!m) continue;
            u64 base = 0;
            if (lane == __builtin_ctzll(m)) base = atomicAdd(&kind_fill[q], (u64)__builtin_popcountll(m));
            base = __shfl(base, __builtin_ctzll(m));
            if (kind == q) {
                const u64 slot = kind_base[q] + base + (u64)__builtin_popcountll(m & ((1ull << lane) - 1));
                jobs[slot] = SubJob{f.scratch + (u64)c * f.blocksize, row_off[(size_t)s * C + c] + f.sample_off, k * (unsigned)C + c, f.blocksize, 0, 0};
            }
        }
    }
    if (used) frames[frame_base[s] + f.seq] = FrameRec{f.sample_off, 0ull, f.blocksize, f.chan_asgn, s, 0};
}

// A value on its way out of the prediction (int32 rows): decorrelated against the partner lane's value of the same sample — lanes 2p / 2p+1
// hold channel 0 / 1 of one stereo frame and walk in step (k_flac_jobs) — and wrapped at 2^(depth-1)  (:482-507, Q14).  `o` and the partner's
// value are below 2^29 in magnitude (the callers see to it), so 32-bit arithmetic is exact; depth > 30 goes to the double rows.
struct FlacTail {
    int asg, asg_u, wrap_half, wrap_full, depth;
    bool odd, uniform;
    __device__ __forceinline__ void init(int lane, int asgn, bool idle, int d) {
        asg = asgn; odd = lane & 1; depth = d;
        asg_u = __builtin_amdgcn_readfirstlane(asgn);
        uniform = __all(idle || asgn == asg_u);   // jobs of one class are sorted by assignment: all but a few waves
        wrap_half = d > 0 && d <= 30 ? 1 << (d - 1) : 0x7FFFFFFF;
        wrap_full = d > 0 && d <= 30 ? 1 << d : 0;
    }
    __device__ __forceinline__ int operator()(int o) const {
        if (depth <= 0) return o;
        int v = o;
        if (uniform) {
            if (asg_u) {
                const int par = __builtin_amdgcn_update_dpp(0, o, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]: lane ^ 1
                if (asg_u == 8) v = odd ? par - o : o;                                   // left/side
                else if (asg_u == 9) v = odd ? o : o + par;                              // side/right
                else { const int side = odd ? o : par; v = (odd ? par : o + par) - (side >> 1); }   // mid/side: right = mid - floor(side / 2), left = right + side
            }
        } else {
            const int par = __builtin_amdgcn_update_dpp(0, o, 0xB1, 0xF, 0xF, false);
            if (asg == 8) v = odd ? par - o : o;
            else if (asg == 9) v = odd ? o : o + par;
            else if (asg == 10) { const int side = odd ? o : par; v = (odd ? par : o + par) - (side >> 1); }
        }
        return v >= wrap_half ? v - wrap_full : v;
    }
};

// restoreLinearPrediction (:411-419) + result[i] * 2^shift (:467-469) + the tail above, int32 rows, the FAST version: one lane per subframe;
// 32 values per lane and round travel global → LDS → lane → LDS → global so that both directions move 128 contiguous bytes per subframe.
// The taps are two full-rate 24-bit multiply-adds each — coef = 2^S ch + cl, S = min(L, 8) — instead of one quarter-rate 64-bit
// multiply-add: floor((2^S Sh + Sl) / 2^L) = (Sh + (Sl >> S)) >> (L - S), exact as long as every restored value stays below a bound hb chosen
// per subframe so that neither partial sum can leave 32 bits.  A wave that cannot promise that (negative shift, many wasted bits) or meets a
// value at or beyond the bound (24-bit audio, garbage) marks itself in `redo` and leaves; k_flac_restore does those waves with 64-bit sums.
// ONE: a wave whose coefficients are small enough that the whole sum stays inside 32 bits for every value the bit depth allows
// (sum |coef| * 2^depth < 2^31: 16-bit audio with up to 14 bits of coefficient magnitude in total) runs one multiply-add per tap.
// Kept lean on purpose (no cross-round prefetch, no 64-bit state): at ~100 VGPRs five waves share a SIMD and cover each other's loads.
__device__ __forceinline__ int mad24(int a, int b, int c) {   // named outright: from __mul24 hipcc derives sign extensions (v_bfe_i32) it does not need
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// the tail for a wave whose jobs share one channel assignment (ASG = 0, 8, 9, 10), or per lane (ASG = -1)
template <int ASG> __device__ __forceinline__ int flac_tail(int o, bool odd, int asg, int wrap_half, int wrap_full) {
    int v = o;
    if constexpr (ASG != 0) {
        const int par = __builtin_amdgcn_update_dpp(0, o, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]: lane ^ 1
        if constexpr (ASG == 8) v = odd ? par - o : o;                                   // left/side
        else if constexpr (ASG == 9) v = odd ? o : o + par;                              // side/right
        else if constexpr (ASG == 10) { const int side = odd ? o : par; v = (odd ? par : o + par) - (side >> 1); }   // mid/side
        else {
            const int side = odd ? o : par;
            const int v8 = odd ? par - o : o, v9 = odd ? o : o + par, v10 = (odd ? par : o + par) - (side >> 1);
            v = asg == 8 ? v8 : (asg == 9 ? v9 : (asg == 10 ? v10 : o));
        }
    }
    return v >= wrap_half ? v - wrap_full : v;
}

// One wave's rounds (k_flac_restore_fast).  Returns false when a value reached the bound.
template <int MAXO, int ASG, bool ONE>
__device__ __forceinline__ bool flac_fast_rounds(const int *scratch, int *rows, int *s_v, const u64 *s_src, const u64 *s_dst, const int *s_bs, int lane, int bs, int maxbs,
                                                 int order, int S, int sh8, int wasted, int hb, int asg, int depth, const int (&cl)[MAXO], const int (&chh)[MAXO]) {
    const bool odd = lane & 1;
    const int wrap_half = 1 << (depth - 1), wrap_full = 1 << depth;   // 1 <= depth <= 30
    const int grp = lane >> 3, sub4 = 4 * (lane & 7);
    int *row = s_v + lane * OSTR;
    int hist[MAXO];
#pragma unroll
    for (int q = 0; q < MAXO; q++) hist[q] = 0;
    unsigned badacc = 0;   // bits at and above 2 hb of (v + hb): zero while every value is inside [-hb, hb)
    // four values: taps named statically, the history moves by four registers once; WARM: the first `order` values are copied (:409)
    auto turn = [&](int k, int i0, auto warm) {
        constexpr bool WARM = decltype(warm)::value;
        const int r0 = row[k], r1 = row[k + 1], r2 = row[k + 2], r3 = row[k + 3];
        const int res[4] = {r0, r1, r2, r3};
        int nv[4], out[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            int sl = 0, sh = 0;
#pragma unroll
            for (int q = 0; q < MAXO; q++) {
                const int t = q < jj ? nv[jj - 1 - q] : hist[q - jj];
                if constexpr (!ONE) sl = mad24(t, cl[q], sl);
                sh = mad24(t, chh[q], sh);
            }
            int pr = ONE ? sh >> sh8 : (sh + (sl >> S)) >> sh8;
            if constexpr (WARM) pr = i0 + jj >= order ? pr : 0;
            const int v = res[jj] + pr;
            badacc |= (unsigned)(v + hb);
            nv[jj] = v;
            out[jj] = flac_tail<ASG>((int)((unsigned)v << wasted), odd, asg, wrap_half, wrap_full);   // |v| < 2^23, wasted < 6
        }
        row[k] = out[0]; row[k + 1] = out[1]; row[k + 2] = out[2]; row[k + 3] = out[3];
#pragma unroll
        for (int q = MAXO - 1; q >= 4; q--) hist[q] = hist[q - 4];
#pragma unroll
        for (int q = 0; q < 4 && q < MAXO; q++) hist[q] = nv[3 - q];
    };
    // Memory schedule of a round: the next round's residuals are requested before this round is predicted, and this round's results wait in
    // registers until the NEXT round's data has been awaited — loads and stores share one counter (vmcnt) and complete out of order with
    // respect to each other, so a wave that waits for loads with stores behind it in flight waits for those stores as well; issued right after
    // the wait, the stores have a whole prediction phase to drain before the next one.
    // The element offsets (below 2^32: the kernel checked) and block sizes of the eight subframes a lane moves data for are fetched once, so a
    // round's requests and stores need no look-ups.
    uint4 pre[8], out[8];
    unsigned src[8], dst[8], bs2[4];   // bs2: two 16-bit block sizes per register
#pragma unroll
    for (int i = 0; i < 8; i++) { src[i] = (unsigned)s_src[8 * i + grp] + sub4; dst[i] = (unsigned)s_dst[8 * i + grp] + sub4; }
#pragma unroll
    for (int i = 0; i < 4; i++) bs2[i] = (unsigned)s_bs[16 * i + grp] | ((unsigned)s_bs[16 * i + 8 + grp] << 16);
    auto bs_of = [&](int i) -> int { return (int)((i & 1) ? bs2[i >> 1] >> 16 : bs2[i >> 1] & 0xFFFFu); };
    auto request = [&](int base) {   // 8 lanes × 4 values per subframe, 8 subframes per instruction
#pragma unroll
        for (int i = 0; i < 8; i++) { pre[i] = make_uint4(0, 0, 0, 0); if (base + sub4 < bs_of(i)) pre[i] = *reinterpret_cast<const uint4 *>(scratch + src[i] + base); }
    };
    auto store = [&](int base) {
#pragma unroll
        for (int i = 0; i < 8; i++) if (base + sub4 < bs_of(i)) *reinterpret_cast<uint4 *>(rows + dst[i] + base) = out[i];
    };
    request(0);
    for (int base = 0; base < maxbs; base += NC) {
#pragma unroll
        for (int i = 0; i < 8; i++) {   // (waits for the requested residuals)
            int *d = s_v + (8 * i + grp) * OSTR + sub4;
            d[0] = (int)pre[i].x; d[1] = (int)pre[i].y; d[2] = (int)pre[i].z; d[3] = (int)pre[i].w;
        }
        if (base > 0) store(base - NC);
        if (base + NC < maxbs) request(base + NC);
        __syncthreads();
        const int nmine = min(NC, bs - base);   // a multiple of 4
        if (base == 0) { for (int k = 0; k < nmine; k += 4) turn(k, k, std::true_type{}); }
        else { for (int k = 0; k < nmine; k += 4) turn(k, 0, std::false_type{}); }
        if (__any((badacc & ~(2u * (unsigned)hb - 1u)) != 0)) return false;   // nothing of this round has left (earlier rounds were exact)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int *d = s_v + (8 * i + grp) * OSTR + sub4;
            out[i] = make_uint4((unsigned)d[0], (unsigned)d[1], (unsigned)d[2], (unsigned)d[3]);
        }
        __syncthreads();
    }
    store(((maxbs - 1) / NC) * NC);
    return true;
}

template <int MAXO>
__global__ __launch_bounds__(64) void k_flac_restore_fast(const SubJob *jobs, u64 njobs, const SubDesc *sd, const int *scratch, int *rows, unsigned *redo, int depth) {
    __shared__ int s_v[64 * OSTR];
    __shared__ u64 s_src[64], s_dst[64];
    __shared__ int s_bs[64];
    const int lane = threadIdx.x;
    const u64 j = (u64)blockIdx.x * 64 + lane;
    SubJob job{0, 0, 0, 0, 0, 0};
    if (j < njobs) job = jobs[j];
    int order = 0, lshift = 0, wasted = 0;
    int cl[MAXO], chh[MAXO];
#pragma unroll
    for (int q = 0; q < MAXO; q++) { cl[q] = 0; chh[q] = 0; }
    int scl = 1, sch = 1, sabs = 1;
    if (j < njobs) {
        const SubDesc &d = sd[job.desc];
        order = d.order; lshift = d.lshift; wasted = d.wasted;
#pragma unroll
        for (int q = 0; q < MAXO; q++) if (q < order) { const int c = (int)d.coef[q]; sabs += c < 0 ? -c : c; }
    }
    const int hb1 = 1 << min(23, __clz(sabs) - 1);   // hb1 * sum |coef| < 2^31
    const bool one = __all(hb1 >= (1 << min(max(depth, 1), 23))) && !AUKIT_FLAC_SPLIT_TAPS;
    if (j < njobs) {
        const SubDesc &d = sd[job.desc];
        const int S = one ? 0 : min(max(lshift, 0), 8);
#pragma unroll
        for (int q = 0; q < MAXO; q++)
            if (q < order) { const int c = (int)d.coef[q]; cl[q] = c & ((1 << S) - 1); chh[q] = c >> S; scl += cl[q]; sch += chh[q] < 0 ? -chh[q] : chh[q]; }
    }
    const int hb = one ? hb1 : min(1 << 23, 1 << (29 - (31 - __clz(max(scl, sch)))));   // a power of two; hb * max(sum cl, sum |ch|) < 2^30
    // 16-byte accesses (block sizes are multiples of 4 but for a stream's last frame), 32-bit element offsets, 16-bit block sizes — else the general kernel
    const bool vec = __all((job.src & 3) == 0 && (job.dst & 3) == 0 && (job.bs & 3) == 0 && job.bs < 65536 && (job.src >> 32) == 0 && (job.dst >> 32) == 0);
    if (!__all(j >= njobs || (lshift >= 0 && wasted < 6)) || !vec || depth > 30 || depth < 1) { if (lane == 0) redo[blockIdx.x] = 1; return; }
    s_src[lane] = job.src; s_dst[lane] = job.dst; s_bs[lane] = job.bs;
    int maxbs = job.bs;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) maxbs = max(maxbs, __shfl_xor(maxbs, m));
    __syncthreads();
    const int S = one ? 0 : min(lshift, 8), sh8 = lshift - S;
    const int asg_u = __builtin_amdgcn_readfirstlane(job.asgn);
    const bool uniform = __all(j >= njobs || job.asgn == asg_u);   // jobs of one class are sorted by assignment: all but a few waves
    bool ok;
#define AUKIT_ROUNDS(A)                                                                                                                                      \
    do {                                                                                                                                                     \
        if (one) ok = flac_fast_rounds<MAXO, A, true>(scratch, rows, s_v, s_src, s_dst, s_bs, lane, job.bs, maxbs, order, S, sh8, wasted, hb, job.asgn, depth, cl, chh); \
        else ok = flac_fast_rounds<MAXO, A, false>(scratch, rows, s_v, s_src, s_dst, s_bs, lane, job.bs, maxbs, order, S, sh8, wasted, hb, job.asgn, depth, cl, chh);   \
    } while (0)
    if (!uniform) AUKIT_ROUNDS(-1);
    else if (asg_u == 0) AUKIT_ROUNDS(0);
    else if (asg_u == 8) AUKIT_ROUNDS(8);
    else if (asg_u == 9) AUKIT_ROUNDS(9);
    else AUKIT_ROUNDS(10);
#undef AUKIT_ROUNDS
    if (!ok && lane == 0) redo[blockIdx.x] = 1;
}

// The general version: int32 rows with 64-bit sums and overflow detection (the waves k_flac_restore_fast left in `redo`, every wave when
// redo is null), or double rows with the Lua's own arithmetic (they leave the tail to k_flac_finish).
template <typename R, int MAXO>
__global__ __launch_bounds__(64) void k_flac_restore(const SubJob *jobs, u64 njobs, const SubDesc *sd, const R *scratch, R *rows, unsigned *flags, int depth,
                                                     const unsigned *redo) {
    constexpr bool INT = std::is_same<R, int>::value;  // int32 rows: exact integer prediction with overflow detection; double rows: the Lua's own arithmetic
    if (redo && !redo[blockIdx.x]) return;
    __shared__ R s_v[64 * OSTR];
    __shared__ u64 s_src[64], s_dst[64];
    __shared__ int s_bs[64];
    const int lane = threadIdx.x;
    const u64 j = (u64)blockIdx.x * 64 + lane;
    SubJob job{0, 0, 0, 0, 0, 0};
    if (j < njobs) job = jobs[j];
    int order = 0, lshift = 0, wasted = 0;
    R coef[MAXO > 0 ? MAXO : 1], hist[MAXO > 0 ? MAXO : 1];
#pragma unroll
    for (int q = 0; q < (MAXO > 0 ? MAXO : 1); q++) { coef[q] = 0; hist[q] = 0; }
    if (j < njobs) {
        const SubDesc &d = sd[job.desc];
        order = d.order; lshift = d.lshift; wasted = d.wasted;
#pragma unroll
        for (int q = 0; q < MAXO; q++) if (q < order) coef[q] = (R)d.coef[q];
    }
    const double div = ldexp(1.0, lshift), mul = ldexp(1.0, wasted);
    s_src[lane] = job.src; s_dst[lane] = job.dst; s_bs[lane] = job.bs;
    bool ovf = INT && depth > 30;
    int maxbs = job.bs;
    [[maybe_unused]] FlacTail fin;
    fin.init(lane, job.asgn, j >= njobs, INT ? depth : 0);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) maxbs = max(maxbs, __shfl_xor(maxbs, m));
    __syncthreads();
    const int half = lane >> 5, k32 = lane & 31;
    // 16-byte path: 8 lanes × 4 values per subframe and round, 8 subframes per instruction; the next round's loads are in flight
    // while this round is predicted.  Needs 4-element alignment of every subframe of the wave (block sizes are multiples of 4 but for
    // a stream's last frame); otherwise 4-byte accesses, two subframes per instruction.
    const bool vec = INT && __all((job.src & 3) == 0 && (job.dst & 3) == 0 && (job.bs & 3) == 0);
    const int grp = lane >> 3, sub4 = 4 * (lane & 7);
    uint4 pre[8];
    auto prefetch = [&](int base) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int s = 8 * i + grp;
            pre[i] = make_uint4(0, 0, 0, 0);
            if (base + sub4 < s_bs[s]) pre[i] = *reinterpret_cast<const uint4 *>(scratch + s_src[s] + base + sub4);
        }
    };
    if (vec) prefetch(0);
    for (int base = 0; base < maxbs; base += NC) {
        if (vec) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                R *d = s_v + (8 * i + grp) * OSTR + sub4;
                d[0] = (R)(int)pre[i].x; d[1] = (R)(int)pre[i].y; d[2] = (R)(int)pre[i].z; d[3] = (R)(int)pre[i].w;
            }
            __syncthreads();
            if (base + NC < maxbs) prefetch(base + NC);
        } else {
            for (int i = 0; i < 32; i++) {
                const int s = 2 * i + half;
                if (base + k32 < s_bs[s]) s_v[s * OSTR + k32] = scratch[s_src[s] + base + k32];
            }
            __syncthreads();
        }
        const int nmine = min(NC, job.bs - base);
        for (int k = 0; k < nmine; k++) {
            const int i = base + k;
            if constexpr (INT) {
                long long v = (long long)s_v[lane * OSTR + k];
                if constexpr (MAXO > 0) {
                    if (i >= order) {
                        long long sum = 0;
#pragma unroll
                        for (int q = 0; q < MAXO; q++) sum += (long long)hist[q] * (long long)coef[q];
                        v += lshift >= 0 ? (sum >> lshift) : (sum << (-lshift));  // floor(sum / 2^shift)
                    }
                    if (v != (long long)(int)v) ovf = true;
#pragma unroll
                    for (int q = MAXO - 1; q > 0; q--) hist[q] = hist[q - 1];
                    hist[0] = (int)v;
                }
                const long long o = v << wasted;
                if (o != (long long)(int)o) ovf = true;
                if (job.asgn && (unsigned long long)(o + (1ll << 29)) >= (1ull << 30)) ovf = true;   // the decorrelation is done in 32 bits
                s_v[lane * OSTR + k] = fin((int)o);
            } else {
                double v = s_v[lane * OSTR + k];
                if constexpr (MAXO > 0) {
                    if (i >= order) {
                        double sum = 0;  // sum = sum + result[i - j] * coefs[j + 1], j = 0 ..  :413-416 (padded taps add exact zeros)
#pragma unroll
                        for (int q = 0; q < MAXO; q++) sum = sum + hist[q] * coef[q];
                        v = v + floor(sum / div);
                    }
#pragma unroll
                    for (int q = MAXO - 1; q > 0; q--) hist[q] = hist[q - 1];
                    hist[0] = v;
                }
                s_v[lane * OSTR + k] = v * mul;
            }
        }
        __syncthreads();
        if (vec) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int s = 8 * i + grp;
                const R *d = s_v + s * OSTR + sub4;
                if (base + sub4 < s_bs[s]) *reinterpret_cast<uint4 *>(rows + s_dst[s] + base + sub4) = make_uint4((unsigned)(int)d[0], (unsigned)(int)d[1], (unsigned)(int)d[2], (unsigned)(int)d[3]);
            }
        } else {
            for (int i = 0; i < 32; i++) {
                const int s = 2 * i + half;
                if (base + k32 < s_bs[s]) rows[s_dst[s] + base + k32] = s_v[s * OSTR + k32];
            }
        }
        __syncthreads();
    }
    if (ovf) atomicOr(flags, (unsigned)FLAG_OVERFLOW);
}

// decodeSubframes tail  :482-507: decorrelate and wrap, in place, one workgroup per chained frame.  int32 rows keep the
// integer (the consumer divides by 2^depth); double rows get the reference's doubles, computed the way the Lua does.
template <typename R>
__global__ __launch_bounds__(256) void k_flac_finish(const FrameRec *frames, const u64 *row_off, int C, R *data, int depth, unsigned *flags) {
    const FrameRec fr = frames[blockIdx.x];
    bool ovf = false;
    if constexpr (std::is_same<R, int>::value) {
        const long long half = 1ll << (depth - 1), full = 1ll << depth;
        auto put = [&](u64 at, long long v) {
            if (v >= half) v -= full;
            if (v != (long long)(int)v) ovf = true;
            data[at] = (int)v;
        };
        if (fr.chan_asgn >= 8) {
            const u64 o0 = row_off[(size_t)fr.stream * C] + fr.sample_off, o1 = row_off[(size_t)fr.stream * C + 1] + fr.sample_off;
            for (int i = threadIdx.x; i < fr.bs; i += 256) {
                long long a = (long long)data[o0 + i], s = (long long)data[o1 + i];
                if (fr.chan_asgn == 8) s = a - s;                                   // left/side
                else if (fr.chan_asgn == 9) a = a + s;                              // side/right
                else { const long long side = s; const long long right = a - (side >> 1); s = right; a = right + side; }  // mid/side, floor(side / 2)
                put(o0 + i, a);
                put(o1 + i, s);
            }
        } else {
            for (int c = 0; c < C; c++) {
                const u64 o = row_off[(size_t)fr.stream * C + c] + fr.sample_off;
                for (int i = threadIdx.x; i < fr.bs; i += 256) put(o + i, (long long)data[o + i]);
            }
        }
    } else {
        const double half = ldexp(1.0, depth - 1), full = ldexp(1.0, depth);
        if (fr.chan_asgn >= 8) {
            const u64 o0 = row_off[(size_t)fr.stream * C] + fr.sample_off, o1 = row_off[(size_t)fr.stream * C + 1] + fr.sample_off;
            for (int i = threadIdx.x; i < fr.bs; i += 256) {
                double a = data[o0 + i], s = data[o1 + i];
                if (fr.chan_asgn == 8) s = a - s;
                else if (fr.chan_asgn == 9) a = a + s;
                else { const double side = s; const double right = a - floor(side / 2); s = right; a = right + side; }
                if (a >= half) a -= full;
                if (s >= half) s -= full;
                data[o0 + i] = a / full;
                data[o1 + i] = s / full;
            }
        } else {
            for (int c = 0; c < C; c++) {
                const u64 o = row_off[(size_t)fr.stream * C + c] + fr.sample_off;
                for (int i = threadIdx.x; i < fr.bs; i += 256) { double a = data[o + i]; if (a >= half) a -= full; data[o + i] = a / full; }
            }
        }
    }
    if (ovf) atomicOr(flags, (unsigned)FLAG_OVERFLOW);
}

struct FlacDecoded {
    hipEvent_t entry = nullptr; // ctx->stream as it stood when the call began (flac_decode_rows)
    DevBuf *set = nullptr;      // round 6: the call's set of tables (aukit_ctx::flac_set) when its first stages run on the look-ahead stream `pre`; null: tmp_buf2, ctx->stream
    hipStream_t pre = nullptr;
    std::vector<FlacStreamInfo> info;
    std::vector<std::vector<std::pair<uint64_t, int>>> frames;  // per stream: (sample offset, blocksize) of every decoded frame, in order
    std::vector<int> status;                                     // per stream: 0 = clean end, else FlacErr of the frame that failed
    std::vector<uint64_t> row_off, row_len;                      // (stream, channel) rows in ctx->tmp_buf
    int channels = 0, depth = 0;
    double rate = 0;
    bool wide = false;  // rows are doubles (already / 2^depth) instead of int32
    // what the decoder left on the device (valid until the next call that uses the context's scratch tables): the frame records in stream
    // order, each stream's first record, each (stream, channel) row's offset; and per stream its frame count
    const struct FrameRec *d_frames = nullptr;
    const u64 *d_fbase = nullptr, *d_rowoff = nullptr;
    std::vector<uint64_t> fbase;
    std::vector<unsigned> nframes;
    uint64_t nfr = 0;
    // the fused decoder (flac_fused.hip) leaves every frame's final integers where it decoded them — ctx->tmp_buf3, at FrameRec::scratch — and
    // the consumers that can follow the frame records read them there (the loader's conversion, the deferred resample + one-pole pass,
    // stream.flac's tail jobs); flac_rows_materialize() gathers contiguous rows into ctx->tmp_buf for the others
    bool in_scratch = false;
    std::vector<std::vector<uint32_t>> frame_end;   // fused decoder, want_frames: the byte behind every frame, relative to its stream's start (0: unknown)
    std::vector<uint64_t> first_frame;             // ... and where the stream's first frame starts (behind the metadata blocks)
    std::vector<uint2> brief;                      // fused decoder, want_frames: (block size, end offset) of every frame in stream order — `frames` / `frame_end` are made of it on demand (frames_from_brief)
    bool scratch16 = false;   // ... as int16 (FusedArgs::out16: asked for by the loader's F32 resample path, depths <= 16)
    bool want16 = false;
    uint64_t tot_elems = 0;          // elements of the contiguous rows (row_off / row_len describe them whether they exist yet or not)
    std::vector<int> bs0;            // per stream: the first frame's block size
    bool uniform = false;            // every stream: all frames but the last have bs0, the last no more
};


static bool g_flac_force_wide() { const char *e = getenv("AUKIT_FLAC_WIDE"); return e && atoi(e) != 0; }

// returns AUKIT_OK, an error, or 1 = "int32 rows overflowed, run again with R = double"
template <typename R>
static int flac_run(aukit_ctx *ctx, const aukit_batch *in, FlacDecoded &D, bool want_frames) {
    const uint32_t n = in->n;
    const int C = D.channels;
    int rc;
    const FlacStreamInfo *d_info = reinterpret_cast<const FlacStreamInfo *>(ctx->misc_buf.p);
    const uintptr_t dptr = reinterpret_cast<uintptr_t>(in->data());
    FlacGlobals G;
    G.src = in->data();
    G.w0 = reinterpret_cast<const u64 *>(dptr & ~(uintptr_t)15);
    G.base_bit = 8 * (dptr & 15);
    G.safe_words = (((dptr & 15) + in->total() + 15) / 16) * 2;
    G.off = reinterpret_cast<const u64 *>(in->d_off);
    G.info = d_info;

    uint64_t capc = in->total() / 2048 + n + 4096;  // candidate slots (grown on demand)
    uint64_t guess = 0;
    for (uint32_t s = 0; s < n; s++) guess += (uint64_t)D.info[s].nsamples * C;
    uint64_t scap = std::max<uint64_t>(guess + guess / 16 + 65536, ctx->tmp_buf3.cap > 256 ? (ctx->tmp_buf3.cap - 256) / sizeof(R) : 0);  // scratch elements (grown on demand)

    for (int attempt = 0;; attempt++) {
        if (attempt > 8) return fail(AUKIT_E_HIP, "internal: FLAC candidate tables keep overflowing");
        // ---- carve tmp_buf2
        uint64_t hs = 1; unsigned hbits = 0;
        while (hs < 2 * capc) { hs <<= 1; hbits++; }
        Carve cv;
        const size_t o_cnt = cv.take(sizeof(Counters)), o_chain = cv.take((size_t)n * sizeof(ChainOut)), o_cand = cv.take(capc * sizeof(Cand)),
                     o_ci = cv.take(capc * sizeof(CandInfo)), o_sd = cv.take(capc * C * sizeof(SubDesc)), o_keys = cv.take(hs * 8), o_vals = cv.take(hs * 4),
                     o_rowoff = cv.take((size_t)n * C * 8), o_fbase = cv.take((size_t)n * 8), o_kbase = cv.take(16 * 8);
        if ((rc = ctx->tmp_buf2.ensure(cv.at))) return rc;
        char *B = reinterpret_cast<char *>(ctx->tmp_buf2.p);
        Counters *d_cnt = reinterpret_cast<Counters *>(B + o_cnt);
        ChainOut *d_chain = reinterpret_cast<ChainOut *>(B + o_chain);
        Cand *d_cand = reinterpret_cast<Cand *>(B + o_cand);
        CandInfo *d_ci = reinterpret_cast<CandInfo *>(B + o_ci);
        SubDesc *d_sd = reinterpret_cast<SubDesc *>(B + o_sd);
        CandHash H{reinterpret_cast<u64 *>(B + o_keys), reinterpret_cast<unsigned *>(B + o_vals), 64 - hbits, hs - 1};
        u64 *d_rowoff = reinterpret_cast<u64 *>(B + o_rowoff), *d_fbase = reinterpret_cast<u64 *>(B + o_fbase), *d_kbase = reinterpret_cast<u64 *>(B + o_kbase);
        AUKIT_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(Counters), ctx->stream));
        AUKIT_HIP_CHECK(hipMemsetAsync(H.keys, 0xFF, hs * 8, ctx->stream));
        // ---- 2. sync candidates
        const uint64_t cand_room = capc - n - 64;  // the tail is kept for positions the chain asks for
        hipLaunchKernelGGL(k_flac_find, dim3((unsigned)std::min<uint64_t>((G.safe_words / 2 + 255) / 256, (uint64_t)ctx->num_cus * 64)), dim3(256), 0, ctx->stream, G,
                           (u64)in->total(), n, d_cand, cand_room, &d_cnt->ncand, H);
        AUKIT_HIP_CHECK(hipGetLastError());
        Counters hc;
        AUKIT_HIP_CHECK(hipMemcpyAsync(&hc, d_cnt, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (hc.ncand > cand_room) { capc = hc.ncand + hc.ncand / 8 + n + 4096; continue; }
        unsigned ncand = (unsigned)hc.ncand;

        // ---- 3. extract every candidate
        auto extract = [&](unsigned first, unsigned count, int limit_factor) -> int {
            if (!count) return AUKIT_OK;
            ExtractArgs<R> A;
            A.G = G; A.cands = d_cand; A.first = first; A.count = count; A.ci = d_ci; A.sd = d_sd; A.C = C;
            A.scratch = reinterpret_cast<R *>(ctx->tmp_buf3.p); A.scratch_cap = scap; A.scratch_cursor = &d_cnt->scratch_cursor; A.flags = &d_cnt->flags;
            A.limit_factor = limit_factor;
            A.ticket = &d_cnt->ticket;
            AUKIT_HIP_CHECK(hipMemsetAsync(&d_cnt->ticket, 0, 4, ctx->stream));
            // lanes pull candidates off the ticket counter: as many workgroups as the chip holds at once (17.6 KB of LDS each: 9 per CU)
            const unsigned grid = std::min<unsigned>((count + 63) / 64, (unsigned)ctx->num_cus * (unsigned)AUKIT_FLAC_EXTRACT_WGS);
            hipLaunchKernelGGL((k_flac_extract<R>), dim3(grid), dim3(64), 0, ctx->stream, A);
            AUKIT_HIP_CHECK(hipGetLastError());
            return AUKIT_OK;
        };
        std::vector<ChainOut> chain(n);
        bool restart = false;
        for (int pass = 0;; pass++) {
            if (pass == 0) {
                if ((rc = ctx->tmp_buf3.ensure(scap * sizeof(R) + 256))) return rc;
                AUKIT_HIP_CHECK(hipMemsetAsync(&d_cnt->scratch_cursor, 0, 8, ctx->stream));
                if ((rc = ctx_begin_kernel(ctx))) return rc;
                if ((rc = extract(0, ncand, 5))) return rc;
                if ((rc = ctx_end_kernel(ctx, "k_flac_extract", in->total() + guess * sizeof(R)))) return rc;
            }
            // ---- 4. follow the chains
            AUKIT_HIP_CHECK(hipMemsetAsync(d_cnt->kind_count, 0, sizeof hc.kind_count + sizeof hc.kind_fill, ctx->stream));
            hipLaunchKernelGGL(k_flac_chain, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, G, n, H, d_ci, d_sd, C, d_chain, d_cnt->kind_count);
            AUKIT_HIP_CHECK(hipGetLastError());
            AUKIT_HIP_CHECK(hipMemcpyAsync(&hc, d_cnt, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
            AUKIT_HIP_CHECK(hipMemcpyAsync(chain.data(), d_chain, (size_t)n * sizeof(ChainOut), hipMemcpyDeviceToHost, ctx->stream));
            AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (getenv("AUKIT_FLAC_STATS"))
                fprintf(stderr, "[flac stats] rounds %llu outer %llu turns %llu values|singles %llu lane_turns %llu live|single_turns %llu\n", hc.stats[0], hc.stats[1], hc.stats[2], hc.stats[3], hc.stats[4], hc.stats[5]);
            if (getenv("AUKIT_FLAC_DEBUG")) {
                std::vector<CandInfo> hci(ncand);
                (void)hipMemcpy(hci.data(), d_ci, ncand * sizeof(CandInfo), hipMemcpyDeviceToHost);
                fprintf(stderr, "[flac] pass %d ncand %u cursor %llu scap %llu flags %u\n", pass, ncand, (u64)hc.scratch_cursor, (u64)scap, hc.flags);
                for (unsigned k = 0; k < ncand && k < 16; k++)
                    fprintf(stderr, "  cand %u end %llu bs %d asgn %d status %d nsub %d used %u seq %u\n", k, hci[k].end_byte, hci[k].blocksize, hci[k].chan_asgn, hci[k].status, hci[k].nsub, hci[k].used, hci[k].seq);
                for (uint32_t s = 0; s < n && s < 4; s++)
                    fprintf(stderr, "  chain %u L %llu nfr %u status %d miss %d at %llu ci %u\n", s, chain[s].L, chain[s].nframes, chain[s].status, chain[s].miss_kind, chain[s].miss_at, chain[s].miss_ci);
            }
            if (hc.scratch_cursor > scap) {  // some frames had no room for their values: grow and extract everything again
                scap = hc.scratch_cursor + hc.scratch_cursor / 16 + 65536;
                pass = -1;
                continue;
            }
            // positions the serial decoder would visit that are not (fully) extracted yet
            std::vector<Cand> extra;
            std::vector<unsigned> redo;
            for (uint32_t s = 0; s < n; s++) {
                if (chain[s].miss_kind == 1) extra.push_back(Cand{s, 1, chain[s].miss_at});
                else if (chain[s].miss_kind == 2) redo.push_back(chain[s].miss_ci);
            }
            if (extra.empty() && redo.empty()) break;
            if (pass > 1000000) return fail(AUKIT_E_HIP, "internal: FLAC chain does not converge");
            if ((uint64_t)ncand + extra.size() > capc) { capc = (uint64_t)ncand + extra.size() * 2 + n + 4096; restart = true; break; }
            if (!extra.empty()) {
                AUKIT_HIP_CHECK(hipMemcpyAsync(d_cand + ncand, extra.data(), extra.size() * sizeof(Cand), hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(k_flac_hash_insert, dim3((unsigned)((extra.size() + 63) / 64)), dim3(64), 0, ctx->stream, d_cand, ncand, (unsigned)extra.size(), H);
                if ((rc = extract(ncand, (unsigned)extra.size(), 0))) return rc;
                AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // `extra` is read by the async copy
                ncand += (unsigned)extra.size();
            }
            for (unsigned k : redo) if ((rc = extract(k, 1, 0))) return rc;
        }
        if (restart) continue;

        // ---- rows, frame lists, job ranges
        D.status.assign(n, 0);
        D.row_off.assign((size_t)n * C, 0);
        D.row_len.assign((size_t)n * C, 0);
        std::vector<uint64_t> fbase(n);
        uint64_t tot = 0, nfr = 0;
        for (uint32_t s = 0; s < n; s++) {
            const uint64_t L = chain[s].L, stride = round_up(std::max<uint64_t>(L, 1), 4);
            for (int c = 0; c < C; c++) { D.row_off[(size_t)s * C + c] = tot + (uint64_t)c * stride; D.row_len[(size_t)s * C + c] = L; }
            tot += stride * C;
            fbase[s] = nfr;
            nfr += chain[s].nframes;
            D.status[s] = chain[s].status;
        }
        uint64_t kbase[17], njobs = 0;   // jobs sorted by order class, then by channel assignment (waves of one assignment decorrelate without divergence)
        for (int q = 0; q < 16; q++) { kbase[q] = njobs; njobs += hc.kind_count[q]; }
        kbase[16] = njobs;
        if ((rc = ctx->tmp_buf.ensure((size_t)tot * sizeof(R) + 256))) return rc;
        Carve cj;
        const size_t o_jobs = cj.take(njobs * sizeof(SubJob)), o_frames = cj.take(nfr * sizeof(FrameRec));
        if ((rc = ctx->seg_buf.ensure(cj.at + 256))) return rc;
        ctx->plan_key.clear();  // seg_buf no longer holds a cached resample plan
        SubJob *d_jobs = reinterpret_cast<SubJob *>(reinterpret_cast<char *>(ctx->seg_buf.p) + o_jobs);
        FrameRec *d_frames = reinterpret_cast<FrameRec *>(reinterpret_cast<char *>(ctx->seg_buf.p) + o_frames);
        if ((rc = h2d_table(ctx, d_rowoff, D.row_off.data(), (size_t)n * C * 8)) || (rc = h2d_table(ctx, d_fbase, fbase.data(), (size_t)n * 8)) ||
            (rc = h2d_table(ctx, d_kbase, kbase, 16 * 8))) return rc;
        R *rows = reinterpret_cast<R *>(ctx->tmp_buf.p);
        const R *scratch = reinterpret_cast<const R *>(ctx->tmp_buf3.p);
        if (njobs) {
            hipLaunchKernelGGL(k_flac_jobs, dim3((ncand + 255) / 256), dim3(256), 0, ctx->stream, d_cand, d_ci, d_sd, ncand, C, d_rowoff, d_fbase, d_kbase, d_cnt->kind_fill,
                               d_jobs, d_frames);
            AUKIT_HIP_CHECK(hipGetLastError());
            // ---- 6. prediction, grouped by order class so that the lanes of a wave run the same number of taps
            constexpr bool fused_tail = std::is_same<R, int>::value;  // int32 rows: decorrelate + wrap inside k_flac_restore
            if ((rc = ctx_begin_kernel(ctx))) return rc;
            // int32 rows: the lean 24-bit kernel first, then the general one on the waves it declined (AUKIT_FLAC_SLOW_RESTORE: on all of them)
            const bool slow_only = getenv("AUKIT_FLAC_SLOW_RESTORE") != nullptr;
            unsigned *d_redo = nullptr;
            if (fused_tail && !slow_only) {
                const size_t nwaves = (size_t)(njobs / 64 + 8);
                if ((rc = ctx->tile_buf.ensure(nwaves * 4))) return rc;
                ctx->plan_key.clear();
                d_redo = reinterpret_cast<unsigned *>(ctx->tile_buf.p);
                AUKIT_HIP_CHECK(hipMemsetAsync(d_redo, 0, nwaves * 4, ctx->stream));
            }
#define AUKIT_RESTORE(Q, MAXO)                                                                                                                     \
    if (kbase[4 * (Q) + 4] > kbase[4 * (Q)]) {                                                                                                     \
        const u64 nj = kbase[4 * (Q) + 4] - kbase[4 * (Q)];                                                                                        \
        const unsigned grid = (unsigned)((nj + 63) / 64);                                                                                          \
        unsigned *redo = d_redo && MAXO > 0 ? d_redo + kbase[4 * (Q)] / 64 + (Q) : nullptr;                                                        \
        if constexpr (fused_tail && MAXO > 0)                                                                                                      \
            if (redo) hipLaunchKernelGGL((k_flac_restore_fast<(MAXO > 0 ? MAXO : 1)>), dim3(grid), dim3(64), 0, ctx->stream, d_jobs + kbase[4 * (Q)], nj, d_sd, \
                                         reinterpret_cast<const int *>(scratch), reinterpret_cast<int *>(rows), redo, D.depth);                    \
        hipLaunchKernelGGL((k_flac_restore<R, MAXO>), dim3(grid), dim3(64), 0, ctx->stream, d_jobs + kbase[4 * (Q)], nj, d_sd, scratch, rows,      \
                           &d_cnt->flags, fused_tail ? D.depth : 0, redo);                                                                         \
    }
            AUKIT_RESTORE(0, 0);
            AUKIT_RESTORE(1, 4);
            AUKIT_RESTORE(2, 12);
            AUKIT_RESTORE(3, 32);
#undef AUKIT_RESTORE
            AUKIT_HIP_CHECK(hipGetLastError());
            if ((rc = ctx_end_kernel(ctx, "k_flac_restore", 2 * tot * sizeof(R)))) return rc;
            // ---- 7. decorrelate + wrap (double rows; int32 rows did it on the way out of the prediction)
            if (!fused_tail) {
                hipLaunchKernelGGL((k_flac_finish<R>), dim3((unsigned)nfr), dim3(256), 0, ctx->stream, d_frames, d_rowoff, C, rows, D.depth, &d_cnt->flags);
                AUKIT_HIP_CHECK(hipGetLastError());
            }
        }
        std::vector<FrameRec> hfr;
        if (want_frames && nfr) {
            hfr.resize(nfr);
            AUKIT_HIP_CHECK(hipMemcpyAsync(hfr.data(), d_frames, nfr * sizeof(FrameRec), hipMemcpyDeviceToHost, ctx->stream));
        }
        unsigned flags = 0;
        AUKIT_HIP_CHECK(hipMemcpyAsync(&flags, &d_cnt->flags, 4, hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (flags & FLAG_OVERFLOW) return 1;  // only the int32 instantiation raises it
        D.frames.assign(n, {});
        D.d_frames = d_frames; D.d_fbase = d_fbase; D.d_rowoff = d_rowoff; D.fbase = fbase; D.nfr = nfr;
        D.nframes.assign(n, 0);
        for (uint32_t s = 0; s < n; s++) D.nframes[s] = chain[s].nframes;
        if (want_frames)
            for (uint32_t s = 0; s < n; s++) {
                D.frames[s].reserve(chain[s].nframes);
                for (unsigned f = 0; f < chain[s].nframes; f++) D.frames[s].push_back({hfr[fbase[s] + f].sample_off, hfr[fbase[s] + f].bs});
            }
        D.wide = sizeof(R) == 8;
        return AUKIT_OK;
    }
}

__global__ __launch_bounds__(256) void k_flac_frames_brief(const FrameRec *frames, u64 nfr, uint2 *out) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < nfr) out[i] = make_uint2((unsigned)frames[i].bs, frames[i].end_rel);
}

// The fused decoder (flac_fused.hip): find → k_flac_decode (final integers into the scratch) → chain → k_flac_gather (scratch → rows).
// Returns AUKIT_OK, an error, or 2 = "the chain needs a frame k_flac_decode declined": the caller runs the first design (flac_run).
static int flac_run_fused(aukit_ctx *ctx, const aukit_batch *in, FlacDecoded &D, bool want_frames) {
    const uint32_t n = in->n;
    const int C = D.channels;
    int rc;
    const bool o16 = D.want16 && D.depth <= 16 && !getenv("AUKIT_FLAC_NO_I16");   // finals as int16 (flac_fused.hip, O16)
    const uintptr_t dptr = reinterpret_cast<uintptr_t>(in->data());
    FlacGlobals G;
    G.src = in->data();
    G.w0 = reinterpret_cast<const u64 *>(dptr & ~(uintptr_t)15);
    G.base_bit = 8 * (dptr & 15);
    G.safe_words = (((dptr & 15) + in->total() + 15) / 16) * 2;
    G.off = reinterpret_cast<const u64 *>(in->d_off);
    G.info = reinterpret_cast<const FlacStreamInfo *>(ctx->misc_buf.p);
    // the call's tables: its set of the two (flac_decode_rows; the stream infos lie at its front), or tmp_buf2 with everything on ctx->stream
    DevBuf &TB = D.set ? *D.set : ctx->tmp_buf2;
    hipStream_t pre = D.set ? D.pre : ctx->stream;
    const size_t info_bytes = D.set ? (((size_t)n * sizeof(FlacStreamInfo) + 255) & ~(size_t)255) : 0;
    uint64_t capc = in->total() / 2048 + n + 4096;
    uint64_t guess = 0;
    for (uint32_t s = 0; s < n; s++) guess += (uint64_t)D.info[s].nsamples * C;
    uint64_t scap = std::max<uint64_t>(guess + guess / 16 + 65536, ctx->tmp_buf3.cap > 256 ? (ctx->tmp_buf3.cap - 256) / 4 : 0);
    for (int attempt = 0;; attempt++) {
        if (attempt > 8) return fail(AUKIT_E_HIP, "internal: FLAC candidate tables keep overflowing");
        uint64_t hs = 1; unsigned hbits = 0;
        while (hs < 2 * capc) { hs <<= 1; hbits++; }
        Carve cv;
        cv.at = info_bytes;
        const size_t o_cnt = cv.take(sizeof(Counters)), o_chain = cv.take((size_t)n * sizeof(ChainOut)), o_cand = cv.take(capc * sizeof(Cand)),
                     o_ci = cv.take(capc * sizeof(CandInfo)), o_keys = cv.take(hs * 8), o_vals = cv.take(hs * 4),
                     o_rowoff = cv.take((size_t)n * C * 8), o_fbase = cv.take((size_t)n * 8), o_links = cv.take(capc * sizeof(ChainLink));
        if (TB.cap < cv.at) {   // (grows: the stream infos at the set's front move with it)
            std::vector<char> keep(info_bytes);
            if (info_bytes) { AUKIT_HIP_CHECK(hipStreamSynchronize(pre)); AUKIT_HIP_CHECK(hipMemcpy(keep.data(), TB.p, info_bytes, hipMemcpyDeviceToHost)); }
            if ((rc = TB.ensure(cv.at + cv.at / 4))) return rc;
            if (info_bytes) AUKIT_HIP_CHECK(hipMemcpy(TB.p, keep.data(), info_bytes, hipMemcpyHostToDevice));
        }
        char *B = reinterpret_cast<char *>(TB.p);
        if (D.set) G.info = reinterpret_cast<const FlacStreamInfo *>(B);
        Counters *d_cnt = reinterpret_cast<Counters *>(B + o_cnt);
        ChainLink *d_links = reinterpret_cast<ChainLink *>(B + o_links);
        ChainOut *d_chain = reinterpret_cast<ChainOut *>(B + o_chain);
        Cand *d_cand = reinterpret_cast<Cand *>(B + o_cand);
        CandInfo *d_ci = reinterpret_cast<CandInfo *>(B + o_ci);
        CandHash H{reinterpret_cast<u64 *>(B + o_keys), reinterpret_cast<unsigned *>(B + o_vals), 64 - hbits, hs - 1};
        u64 *d_rowoff = reinterpret_cast<u64 *>(B + o_rowoff), *d_fbase = reinterpret_cast<u64 *>(B + o_fbase);
        AUKIT_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(Counters), pre));
        AUKIT_HIP_CHECK(hipMemsetAsync(H.keys, 0xFF, hs * 8, pre));
        const uint64_t cand_room = capc - n - 64;
        hipLaunchKernelGGL(k_flac_find, dim3((unsigned)std::min<uint64_t>((G.safe_words / 2 + 255) / 256, (uint64_t)ctx->num_cus * 64)), dim3(256), 0, pre, G,
                           (u64)in->total(), n, d_cand, cand_room, &d_cnt->ncand, H);
        AUKIT_HIP_CHECK(hipGetLastError());
        Counters hc;
        AUKIT_HIP_CHECK(hipMemcpyAsync(&hc, d_cnt, sizeof hc, hipMemcpyDeviceToHost, pre));
        if (D.set) AUKIT_HIP_CHECK(hipEventRecord(ctx->pre_ev, pre));
        AUKIT_HIP_CHECK(hipStreamSynchronize(pre));
        if (hc.ncand > cand_room) { capc = hc.ncand + hc.ncand / 8 + n + 4096; continue; }
        // Round 6, late: the decoder and the chain walk stay on the look-ahead stream too (AUKIT_FLAC_DECODE_AHEAD=0: behind the search on ctx->stream as
        // before).  The call before's last passes — its normalize: HBM-bound, no LDS — then run BESIDE this call's decoder (VALU-bound, the CU's whole LDS)
        // instead of in front of it.  The frame scratch is the one buffer both streams touch: the decoder waits for its last reader (scratch_ev: the tile
        // chain of the call before), or for all of ctx->stream when nothing tracked the readers.  Every nested launch goes where ctx->stream points: it
        // points at the look-ahead stream until the chain has converged.
        const bool dahead = D.set && !(getenv("AUKIT_FLAC_DECODE_AHEAD") && atoi(getenv("AUKIT_FLAC_DECODE_AHEAD")) == 0);
        struct StreamSwap { aukit_ctx *c; hipStream_t saved; bool on; void back() { if (on) { c->stream = saved; on = false; } } ~StreamSwap() { back(); } } sw{ctx, ctx->stream, false};
        if (dahead) {
            if (ctx->scratch_dirty && D.entry) AUKIT_HIP_CHECK(hipStreamWaitEvent(pre, D.entry, 0));
            else if (ctx->scratch_ev_set) AUKIT_HIP_CHECK(hipStreamWaitEvent(pre, ctx->scratch_ev, 0));
            ctx->stream = pre; sw.on = true;
        } else if (D.set) AUKIT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->pre_ev, 0));   // the decoder (ctx->stream) behind the search
        unsigned ncand = (unsigned)hc.ncand;
        auto decode = [&](unsigned first, unsigned count, int limit_factor) -> int {
            FusedArgs A;
            A.G = G; A.cands = d_cand; A.first = first; A.count = count; A.ci = d_ci; A.C = C; A.depth = D.depth;
            A.scratch = reinterpret_cast<int *>(ctx->tmp_buf3.p); A.scratch_cap = scap; A.scratch_cursor = &d_cnt->scratch_cursor; A.flags = &d_cnt->flags;
            A.limit_factor = limit_factor; A.ticket = &d_cnt->ticket; A.stats = d_cnt->stats;
            A.out16 = o16 ? 1 : 0;
            A.dbg = getenv("AUKIT_FLAC_FUSED_DBG") ? atoi(getenv("AUKIT_FLAC_FUSED_DBG")) : 0;
            // k_flac_stream (a wave per 64 frames) when the batch fills the chip; k_flac_pq (a parser and a predictor wave per 64 frames: the lane's chain
            // cut in two) when it does not — fewer frames than three of its workgroups per CU hold at once.  AUKIT_FLAC_DECODER = stream | pq | fused for the A/B
            // (fused: k_flac_decode, flac_fused.hip, round 4)
            const char *which = getenv("AUKIT_FLAC_DECODER");
            if (which && !strcmp(which, "fused")) return flac_fused_launch(ctx, A);
            const bool pq = which ? !strcmp(which, "pq") : (uint64_t)A.count <= (uint64_t)ctx->num_cus * 3ull * 64ull;
            return pq ? flac_pq_launch(ctx, A) : flac_stream_launch(ctx, A);
        };
        std::vector<ChainOut> chain(n);
        bool restart = false;
        for (int pass = 0;; pass++) {
            if (pass == 0) {
                if ((rc = ctx->tmp_buf3.ensure(scap * 4 + 256))) return rc;
                AUKIT_HIP_CHECK(hipMemsetAsync(&d_cnt->scratch_cursor, 0, 8, ctx->stream));
                if ((rc = ctx_begin_kernel(ctx))) return rc;
                if ((rc = decode(0, ncand, 5))) return rc;
                if ((rc = ctx_end_kernel(ctx, "k_flac_decode", in->total() + guess * 4))) return rc;
            }
            if (getenv("AUKIT_FLAC_CHAIN_OLD")) hipLaunchKernelGGL(k_flac_chain, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, G, n, H, d_ci, (const SubDesc *)nullptr, C, d_chain, d_cnt->kind_count);
            else {
                hipLaunchKernelGGL(k_flac_links, dim3((ncand + 255) / 256), dim3(256), 0, ctx->stream, G, d_cand, d_ci, ncand, H, d_links);
                hipLaunchKernelGGL(k_flac_chain_links, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, G, n, H, d_ci, d_links, d_chain);
            }
            AUKIT_HIP_CHECK(hipGetLastError());
            AUKIT_HIP_CHECK(hipMemcpyAsync(&hc, d_cnt, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
            AUKIT_HIP_CHECK(hipMemcpyAsync(chain.data(), d_chain, (size_t)n * sizeof(ChainOut), hipMemcpyDeviceToHost, ctx->stream));
            AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (getenv("AUKIT_FLAC_STATS"))
                fprintf(stderr, "[flac stats] rounds %llu outer %llu turns %llu values|singles %llu lane_turns %llu live|single_turns %llu\n", hc.stats[0], hc.stats[1], hc.stats[2], hc.stats[3], hc.stats[4], hc.stats[5]);
            if (getenv("AUKIT_FLAC_DEBUG")) {
                std::vector<CandInfo> hci(ncand);
                (void)hipMemcpy(hci.data(), d_ci, ncand * sizeof(CandInfo), hipMemcpyDeviceToHost);
                fprintf(stderr, "[flac fused] pass %d ncand %u cursor %llu scap %llu\n", pass, ncand, (u64)hc.scratch_cursor, (u64)scap);
                for (unsigned k = 0; k < ncand && k < 16; k++)
                    fprintf(stderr, "  cand %u end %llu bs %d asgn %d status %d nsub %d used %u seq %u\n", k, hci[k].end_byte, hci[k].blocksize, hci[k].chan_asgn, hci[k].status, hci[k].nsub, hci[k].used, hci[k].seq);
                for (uint32_t s = 0; s < n && s < 4; s++)
                    fprintf(stderr, "  chain %u L %llu nfr %u status %d miss %d at %llu ci %u\n", s, chain[s].L, chain[s].nframes, chain[s].status, chain[s].miss_kind, chain[s].miss_at, chain[s].miss_ci);
            }
            if (hc.scratch_cursor > scap) { scap = hc.scratch_cursor + hc.scratch_cursor / 16 + 65536; pass = -1; continue; }
            std::vector<Cand> extra;
            std::vector<unsigned> redo;
            for (uint32_t s = 0; s < n; s++) {
                if (chain[s].miss_kind == 1) extra.push_back(Cand{s, 1, chain[s].miss_at});
                else if (chain[s].miss_kind == 2) redo.push_back(chain[s].miss_ci);
            }
            if (extra.empty() && redo.empty()) break;
            if (pass > 1000000) return fail(AUKIT_E_HIP, "internal: FLAC chain does not converge");
            if ((uint64_t)ncand + extra.size() > capc) { capc = (uint64_t)ncand + extra.size() * 2 + n + 4096; restart = true; break; }
            if (!extra.empty()) {
                AUKIT_HIP_CHECK(hipMemcpyAsync(d_cand + ncand, extra.data(), extra.size() * sizeof(Cand), hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(k_flac_hash_insert, dim3((unsigned)((extra.size() + 63) / 64)), dim3(64), 0, ctx->stream, d_cand, ncand, (unsigned)extra.size(), H);
                if ((rc = decode(ncand, (unsigned)extra.size(), 0))) return rc;
                AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                ncand += (unsigned)extra.size();
            }
            for (unsigned k : redo) if ((rc = decode(k, 1, 0))) return rc;   // hit their bit budget and are needed after all: once more, without one
        }
        if (dahead) {   // ctx->stream behind everything the look-ahead stream did for this call
            sw.back();
            AUKIT_HIP_CHECK(hipEventRecord(ctx->pre_ev, pre));
            AUKIT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->pre_ev, 0));
        }
        ctx->scratch_dirty = true;   // (until a deferred resample takes the buffer: lazy_resample_try)
        if (restart) continue;
        for (uint32_t s = 0; s < n; s++) if (chain[s].status == FE_DECLINE) return 2;
        if (o16 && (hc.flags & 0x100u)) return 3;   // a final value beyond int16 (not an ordinary stream): once more with int32 finals

        D.status.assign(n, 0);
        D.row_off.assign((size_t)n * C, 0);
        D.row_len.assign((size_t)n * C, 0);
        std::vector<uint64_t> fbase(n);
        uint64_t tot = 0, nfr = 0;
        for (uint32_t s = 0; s < n; s++) {
            const uint64_t L = chain[s].L, stride = round_up(std::max<uint64_t>(L, 1), 4);
            for (int c = 0; c < C; c++) { D.row_off[(size_t)s * C + c] = tot + (uint64_t)c * stride; D.row_len[(size_t)s * C + c] = L; }
            tot += stride * C;
            fbase[s] = nfr;
            nfr += chain[s].nframes;
            D.status[s] = chain[s].status;
        }
        if ((rc = ctx->seg_buf.ensure(nfr * sizeof(FrameRec) + 512))) return rc;
        ctx->plan_key.clear();
        FrameRec *d_frames = reinterpret_cast<FrameRec *>(ctx->seg_buf.p);
        if ((rc = h2d_table(ctx, d_rowoff, D.row_off.data(), (size_t)n * C * 8)) || (rc = h2d_table(ctx, d_fbase, fbase.data(), (size_t)n * 8))) return rc;
        if (nfr && (rc = flac_frames_launch(ctx, d_cand, d_ci, ncand, d_fbase, d_frames, G.off))) return rc;
        D.in_scratch = true;
        D.scratch16 = o16;
        D.tot_elems = tot;
        D.bs0.assign(n, 0);
        D.uniform = true;
        for (uint32_t s = 0; s < n; s++) { D.bs0[s] = chain[s].bs0; if (!chain[s].uniform) D.uniform = false; }
        // what the host needs of the frame records (stream.flac's chunk table): block size and end offset — 8 of a record's 32 bytes come down
        // (110 K records of a 1024-stream call: 3.5 MB and 0.15 ms of a call that is measured against 8 ms)
        std::vector<uint2> hfr;
        if (want_frames && nfr) {
            hfr.resize(nfr);
            if ((rc = ctx->tile_buf.ensure(nfr * sizeof(uint2) + 64))) return rc;
            ctx->plan_key.clear();
            hipLaunchKernelGGL(k_flac_frames_brief, dim3((unsigned)((nfr + 255) / 256)), dim3(256), 0, ctx->stream, d_frames, nfr, reinterpret_cast<uint2 *>(ctx->tile_buf.p));
            AUKIT_HIP_CHECK(hipGetLastError());
            void *stage = ctx_host_stage(ctx, nfr * sizeof(uint2));   // (pinned: a pageable destination costs the copy a staging pass of its own)
            AUKIT_HIP_CHECK(hipMemcpyAsync(stage ? stage : (void *)hfr.data(), ctx->tile_buf.p, nfr * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
            AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (stage) memcpy(hfr.data(), stage, nfr * sizeof(uint2));
        }
        D.frames.assign(n, {});
        D.d_frames = d_frames; D.d_fbase = d_fbase; D.d_rowoff = d_rowoff; D.fbase = fbase; D.nfr = nfr;
        D.nframes.assign(n, 0);
        for (uint32_t s = 0; s < n; s++) D.nframes[s] = chain[s].nframes;
        if (want_frames) {   // (the per-stream vectors are host work the device does not wait for: stream.flac makes them after its tail kernel is launched)
            D.brief = std::move(hfr);
            D.first_frame.assign(n, 0);
            for (uint32_t s = 0; s < n; s++) D.first_frame[s] = D.info[s].first_byte;
        }
        D.wide = false;
        return AUKIT_OK;
    }
}

// contiguous int32 rows in ctx->tmp_buf from the frames the fused decoder left in ctx->tmp_buf3 (the consumers that want rows)
static int flac_rows_materialize(aukit_ctx *ctx, FlacDecoded &D) {
    if (!D.in_scratch) return AUKIT_OK;
    int rc;
    if ((rc = ctx->tmp_buf.ensure((size_t)D.tot_elems * 4 + 256))) return rc;
    if (D.nfr) {
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        if ((rc = flac_gather_launch(ctx, D.d_frames, D.nfr, D.channels, D.d_rowoff, reinterpret_cast<const int *>(ctx->tmp_buf3.p), reinterpret_cast<int *>(ctx->tmp_buf.p), D.scratch16))) return rc;
        if ((rc = ctx_end_kernel(ctx, "k_flac_gather", 2 * D.tot_elems * 4))) return rc;
    }
    D.in_scratch = false;
    D.scratch16 = false;
    return AUKIT_OK;
}

static int flac_decode_rows(aukit_ctx *ctx, const aukit_batch *in, FlacDecoded &D, bool want_frames) {
    const uint32_t n = in->n;
    if (n == 0) return fail(AUKIT_E_ARG, "empty batch");
    int rc;
    // -- 1. stream headers.  Round 6: on the look-ahead stream (common.h: pre_stream) — this stage and the sync search read the input batch and
    // nothing else, and their host waits used to be waits for everything the call BEFORE had left on ctx->stream (config 5: its filter and
    // normalize passes, 4.5 ms): back-to-back calls now search while the one before still filters.  The tables of a call lie in one of two
    // alternating sets (the call before may still be read by its last kernels); ctx->stream sees the stream infos as a copy in misc_buf, in
    // stream order, for the first design's kernels.
    hipStream_t pre = ctx->stream;
    D.set = nullptr;
    // ... where the call before leaves the chip room: measured on config 5 with k_rsp behind the decoder (profiles/r06_flac_lookahead.txt) the search beside the
    // filter and normalize passes takes 0.13 - 0.25 ms off a step of 256 - 1024 ten-second streams and ADDS 0.22 ms to one of 2048 (what the search
    // takes from the kernels it runs beside is more than its own 0.6 ms there).  AUKIT_FLAC_LOOKAHEAD=0 / 1 decides it by hand
    // (with the decoder on the look-ahead stream as well — flac_run_fused — it pays at every size again: 2048 streams 11.0 ms against 11.35 with the search alone
    // there and 11.5 with everything on ctx->stream, one box)
    bool ahead = true;
    if (const char *e = getenv("AUKIT_FLAC_LOOKAHEAD")) ahead = atoi(e) != 0;
    if (getenv("AUKIT_FLAC_NO_LOOKAHEAD")) ahead = false;
    if (ahead) {
        if ((rc = ctx_pre_stream(ctx, &pre))) return rc;
        if (in->ready) AUKIT_HIP_CHECK(hipStreamWaitEvent(pre, in->ready, 0));
        // (call k writes the table set call k - 2 used: whatever read that one on ctx->stream was queued before call k - 1 began)
        const uint64_t k = ctx->flac_calls++;
        AUKIT_HIP_CHECK(hipEventRecord(ctx->entry_ev[k & 1], ctx->stream));
        if (k >= 1) AUKIT_HIP_CHECK(hipStreamWaitEvent(pre, ctx->entry_ev[(k - 1) & 1], 0));
        D.entry = ctx->entry_ev[k & 1];
        ctx->flac_par ^= 1;
        D.set = &ctx->flac_set[ctx->flac_par];
    }
    D.pre = pre;
    DevBuf &hb = ctx->misc_buf;
    if ((rc = hb.ensure((size_t)n * sizeof(FlacStreamInfo)))) return rc;
    FlacStreamInfo *d_info_pre = reinterpret_cast<FlacStreamInfo *>(hb.p);
    if (D.set) {
        if ((rc = D.set->ensure(((size_t)n * sizeof(FlacStreamInfo) + 255) & ~(size_t)255))) return rc;
        d_info_pre = reinterpret_cast<FlacStreamInfo *>(D.set->p);
    }
    hipLaunchKernelGGL(k_flac_header, dim3((n + 63) / 64), dim3(64), 0, pre, in->data(), reinterpret_cast<const u64 *>(in->d_off), n, d_info_pre);
    AUKIT_HIP_CHECK(hipGetLastError());
    D.info.resize(n);
    AUKIT_HIP_CHECK(hipMemcpyAsync(D.info.data(), d_info_pre, (size_t)n * sizeof(FlacStreamInfo), hipMemcpyDeviceToHost, pre));
    AUKIT_HIP_CHECK(hipStreamSynchronize(pre));
    if (D.set) AUKIT_HIP_CHECK(hipMemcpyAsync(hb.p, D.info.data(), (size_t)n * sizeof(FlacStreamInfo), hipMemcpyHostToDevice, ctx->stream));   // (pageable: the copy is staged before this returns)
    for (uint32_t s = 0; s < n; s++) {
        switch (D.info[s].status) {
        case 0: break;
        case 2: return fail(AUKIT_E_LUA, "Invalid magic string");
        case 3: return fail(AUKIT_E_LUA, "Stream info metadata block absent");
        case 4: return fail(AUKIT_E_LUA, "Sample depth not supported");
        default: return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value");
        }
        if (s == 0) { D.channels = D.info[0].channels; D.depth = D.info[0].depth; D.rate = D.info[0].rate; }
        else if (D.info[s].channels != D.channels || D.info[s].depth != D.depth || D.info[s].rate != D.rate)
            return fail(AUKIT_E_ARG, "all FLAC streams of a batch must share channel count, bit depth and sample rate");
    }
    const bool no_fused = getenv("AUKIT_FLAC_NO_FUSED") != nullptr;   // A/B and the tests: the first design for every batch
    if (D.depth <= 24 && !g_flac_force_wide() && !no_fused && !getenv("AUKIT_FLAC_SLOW_RESTORE")) {
        rc = flac_run_fused(ctx, in, D, want_frames);
        if (rc == 3) { D.want16 = false; rc = flac_run_fused(ctx, in, D, want_frames); }
        ctx->counters[AUKIT_COUNTER_FLAC_FUSED] = rc == AUKIT_OK ? 1 : 0;
        if (rc != 2) return rc;
        if (getenv("AUKIT_FLAC_DEBUG")) fprintf(stderr, "[flac] the fused decoder declined a frame of the chain: two-kernel decoder\n");
    } else ctx->counters[AUKIT_COUNTER_FLAC_FUSED] = 0;
    if (D.depth <= 24 && !g_flac_force_wide()) {
        rc = flac_run<int>(ctx, in, D, want_frames);
        if (rc != 1) return rc;
    }
    return flac_run<double>(ctx, in, D, want_frames);
}

static void frames_from_brief(FlacDecoded &D) {
    if (D.brief.empty()) return;
    const size_t n = D.nframes.size();
    D.frames.assign(n, {});
    D.frame_end.assign(n, {});
    for (size_t s = 0; s < n; s++) {
        D.frames[s].reserve(D.nframes[s]);
        D.frame_end[s].reserve(D.nframes[s]);
        uint64_t at = 0;   // (a frame's sample offset is the block sizes before it: the chain walk numbers them in order)
        for (unsigned f = 0; f < D.nframes[s]; f++) {
            const uint2 b = D.brief[D.fbase[s] + f];
            D.frames[s].push_back({at, (int)b.x});
            D.frame_end[s].push_back(b.y);
            at += b.x;
        }
    }
    D.brief.clear();
}

// aukit.flac(data)  aukit.lua:1657-1660
int decode_flac_audio(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, double new_rate, int interp, bool do_resample, int dtype,
                      aukit_audio **out) {
    FlacDecoded D;
    if (*out && ((*out)->lazy_rs || (*out)->lazy_rows.p)) lazy_drop(ctx, *out);   // the output's old rows (a deferred resample nobody asked for) return to the scratch the decoder is about to use
    D.want16 = true;   // every reader behind the loader takes int16 finals: the F32 resample path in place (k_rs_onepole<..., short>), the others through k_flac_gather
    int rc = flac_decode_rows(ctx, in, D, false);
    if (rc) return rc;
    for (uint32_t s = 0; s < in->n; s++)
        if (D.status[s]) return fail(AUKIT_E_LUA, "%s", flac_err_msg(D.status[s]));  // decodeFLAC raises: the whole load fails
    const double full = std::ldexp(1.0, D.depth);  // :505
    if (D.in_scratch && !do_resample && (dtype == AUKIT_F32 || dtype == AUKIT_F64)) {
        // the loader without a resample: the frames go from where they were decoded straight into the audio's rows, converted on the way
        std::vector<uint64_t> lens(in->n);
        for (uint32_t s = 0; s < in->n; s++) {
            lens[s] = D.row_len[(size_t)s * D.channels];
            if (lens[s] > 0x7FFFFFF0ull) return fail(AUKIT_E_UNSUPPORTED, "stream too long");
        }
        aukit_audio *a = *out;
        if ((rc = audio_prepare(ctx, &a, in->n, D.channels, D.rate, dtype, lens.data()))) return rc;
        *out = a;
        if (!D.nfr) return AUKIT_OK;
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        if ((rc = flac_gather_convert_launch(ctx, D.d_frames, D.nfr, D.channels, reinterpret_cast<const int *>(ctx->tmp_buf3.p), reinterpret_cast<const u64 *>(a->d_meta), in->n, a->dev, dtype, full, D.scratch16))) return rc;
        return ctx_end_kernel(ctx, "k_flac_gather<convert>", D.tot_elems * (D.scratch16 ? 2 : 4) + D.tot_elems * dtype_size(dtype));
    }
    if (!D.wide && do_resample && dtype == AUKIT_F32) {   // F32 pipelines: the resample is owed — a following effects.highpass / lowpass pays it in its own pass (flac_tail.hip)
        int lrc = AUKIT_OK;
        LazyFrames LF{D.d_frames, D.nfr, D.d_fbase, D.d_rowoff, &D.bs0, D.uniform, D.tot_elems, &D.nframes, D.scratch16};
        if (lazy_resample_try(ctx, D.row_off, D.row_len, in->n, D.channels, D.rate, new_rate, interp, full, out, &lrc, D.in_scratch ? &LF : nullptr)) return lrc;
        if (D.in_scratch) {   // frames too short (or of mixed sizes) to be followed in place: contiguous rows, and the resample is owed on those
            if ((rc = flac_rows_materialize(ctx, D))) return rc;
            if (lazy_resample_try(ctx, D.row_off, D.row_len, in->n, D.channels, D.rate, new_rate, interp, full, out, &lrc, nullptr)) return lrc;
        }
    }
    if ((rc = flac_rows_materialize(ctx, D))) return rc;
    if (D.wide) return audio_from_int_rows(ctx, SRC_AUDIO_F64, ctx->tmp_buf.p, D.row_off, D.row_len, in->n, D.channels, D.rate, new_rate, interp, do_resample, dtype, 1, 1, out);
    return audio_from_int_rows(ctx, SRC_I32, ctx->tmp_buf.p, D.row_off, D.row_len, in->n, D.channels, D.rate, new_rate, interp, do_resample, dtype, full, full, out);
}

// ================================================================= stream.flac  aukit.lua:3124-3191
struct FsJob {
    u64 src_off;   // element offset of this (frame, channel) block in the decoded rows
    u64 last_off;  // element offset of src[0] = last[2], the previous (frame, channel) block's last sample; ~0 → 0
    u64 m1_off;    // element offset of src[-1] = last[1]: the sample before it, or — after a one-sample block, whose src[#src-1] is its
                   // own src[0] (:3183) — the last sample of the block before that one; ~0 → 0
    u64 out_off;
    int blocksize, nout;
};
template <typename R> AUKIT_DEV double flac_row_value(const R *rows, u64 at, double full) {
    if constexpr (sizeof(R) == 4) return (double)rows[at] / full;
    else return rows[at];
}
template <int INTERP, typename OUT_T, typename R>
__global__ __launch_bounds__(64) void k_flac_stream(const FsJob *jobs, u64 njobs, const R *rows, double full, OUT_T *out, double ratio, double rcp, int exact,
                                                   double lp_alpha, int sinc_w) {
    const u64 j = (u64)blockIdx.x * 64 + threadIdx.x;
    if (j >= njobs) return;
    const FsJob job = jobs[j];
    double m1 = 0, z0 = 0;                   // src[-1], src[0] = last[1], last[2]  :3170-3171
    if (job.last_off != ~0ull) z0 = flac_row_value(rows, job.last_off, full);
    if (job.m1_off != ~0ull) m1 = flac_row_value(rows, job.m1_off, full);
    double ls = z0 / (z0 < 0 ? 128 : 127);   // :3172
    const int n = job.blocksize;
    auto tap = [&](int k) -> double { return k >= 1 ? flac_row_value(rows, job.src_off + (u64)(k - 1), full) : (k == 0 ? z0 : m1); };  // table index k
    for (int i = 0; i < job.nout; i++) {
        const double nn = (double)i;
        const double x = (exact ? div_rcp(nn, ratio, rcp) : nn / ratio) + 1.0;
        const double ffx = floor(x);
        const int k = (int)ffx;
        double s;
        if (x == ffx) s = tap(k);
        else {
            const double fx = x - ffx;
            if constexpr (INTERP == AUKIT_INTERP_NONE) s = tap(k);
            else if constexpr (INTERP == AUKIT_INTERP_LINEAR) { const double a = tap(k), b = (k + 1 <= n) ? tap(k + 1) : a; s = linear_exact(a, b, fx); }
            else if constexpr (INTERP == AUKIT_INTERP_SINC) s = sinc_at(tap, k, -1, n, fx, sinc_w);
            else {
                const double p1 = tap(k), p0 = (k - 1 >= -1) ? tap(k - 1) : p1, p2 = (k + 1 <= n) ? tap(k + 1) : p1, p3 = (k + 2 <= n) ? tap(k + 2) : p2;
                s = cubic_exact(p0, p1, p2, p3, fx);
            }
        }
        s = ls + lp_alpha * (s - ls);  // :3179 (recursive: ls = filtered s, Q14)
        ls = s;
        out[job.out_off + i] = (OUT_T)lua_clamp(s * (s < 0 ? 128 : 127), -128, 127);
    }
}

// The same in two passes (what stream.qoa does, qoa_stream.hip): with one lane per (frame, channel) job every output pays four dependent,
// scattered table loads and a scattered 4-byte store (1024 stereo streams of ten seconds: 36 ms).  Pass 1: the interpolated sample of
// every output, all outputs in parallel, into a scratch of doubles; pass 2: the recursive low-pass (:3179) and the scaling, serially per job
// over contiguous doubles, 32 per round with the next 32 in flight, 16-byte stores.  Same operations in the same order: same values.
template <int INTERP, typename R>
__global__ __launch_bounds__(256) void k_flac_stream_interp(const FsJob *jobs, const u64 *scr_off, const R *rows, double full, double *scr, double ratio, double rcp, int exact, int sinc_w) {
    const FsJob job = jobs[blockIdx.y];
    double m1 = 0, z0 = 0;
    if (job.last_off != ~0ull) z0 = flac_row_value(rows, job.last_off, full);
    if (job.m1_off != ~0ull) m1 = flac_row_value(rows, job.m1_off, full);
    const int n = job.blocksize;
    auto tap = [&](int k) -> double { return k >= 1 ? flac_row_value(rows, job.src_off + (u64)(k - 1), full) : (k == 0 ? z0 : m1); };
    double *o = scr + scr_off[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < job.nout; i += gridDim.x * 256) {
        const double nn = (double)i;
        const double x = (exact ? div_rcp(nn, ratio, rcp) : nn / ratio) + 1.0;
        const double ffx = floor(x);
        const int k = (int)ffx;
        double s;
        if (x == ffx) s = tap(k);
        else {
            const double fx = x - ffx;
            if constexpr (INTERP == AUKIT_INTERP_NONE) s = tap(k);
            else if constexpr (INTERP == AUKIT_INTERP_LINEAR) { const double a = tap(k), b = (k + 1 <= n) ? tap(k + 1) : a; s = linear_exact(a, b, fx); }
            else if constexpr (INTERP == AUKIT_INTERP_SINC) s = sinc_at(tap, k, -1, n, fx, sinc_w);
            else {
                const double p1 = tap(k), p0 = (k - 1 >= -1) ? tap(k - 1) : p1, p2 = (k + 1 <= n) ? tap(k + 1) : p1, p3 = (k + 2 <= n) ? tap(k + 2) : p2;
                s = cubic_exact(p0, p1, p2, p3, fx);
            }
        }
        o[i] = s;
    }
}
template <typename OUT_T, typename R>
__global__ __launch_bounds__(64) void k_flac_stream_iir(const FsJob *jobs, const u64 *scr_off, u64 njobs, const R *rows, double full, const double *scr, OUT_T *out, double lp_alpha) {
    const u64 j = (u64)blockIdx.x * 64 + threadIdx.x;
    if (j >= njobs) return;
    const FsJob job = jobs[j];
    double z0 = 0;
    if (job.last_off != ~0ull) z0 = flac_row_value(rows, job.last_off, full);
    double ls = z0 / (z0 < 0 ? 128 : 127);   // :3172
    const double *p = scr + scr_off[j];
    OUT_T *o = out + job.out_off;
    constexpr int RN = 32, PV = 16 / (int)sizeof(OUT_T);
    typedef OUT_T ovp __attribute__((ext_vector_type(PV), aligned(sizeof(OUT_T))));
    const int rounds = job.nout / RN;
    double cur[RN], nxt[RN];
    if (rounds) {
#pragma unroll
        for (int k = 0; k < RN; k++) cur[k] = p[k];
    }
    auto step = [&](double v) -> OUT_T {
        const double s = ls + lp_alpha * (v - ls);  // :3179 (recursive: ls = filtered s, Q14)
        ls = s;
        return (OUT_T)lua_clamp(s * (s < 0 ? 128 : 127), -128, 127);
    };
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int k = 0; k < RN; k++) nxt[k] = cur[k];
        if (r + 1 < rounds) {
#pragma unroll
            for (int k = 0; k < RN; k++) nxt[k] = p[(r + 1) * RN + k];
        }
        OUT_T res[RN];
#pragma unroll
        for (int k = 0; k < RN; k++) res[k] = step(cur[k]);
#pragma unroll
        for (int v = 0; v < RN / PV; v++) {
            ovp w;
#pragma unroll
            for (int e = 0; e < PV; e++) w[e] = res[v * PV + e];
            *reinterpret_cast<ovp *>(o + r * RN + v * PV) = w;
        }
#pragma unroll
        for (int k = 0; k < RN; k++) cur[k] = nxt[k];
    }
    for (int i = rounds * RN; i < job.nout; i++) o[i] = step(p[i]);
}

// the tail jobs of stream.flac (one per (frame, channel), in the order the reference walks them) from the frame records the decoder left on the
// device: one lane per stream — `last = {src[#src-1], src[#src]}` is shared across channels and frames (Q14), so a stream's jobs chain
__global__ __launch_bounds__(64) void k_flac_tail_jobs(const FrameRec *frames, const u64 *fbase, const unsigned *nframes, const u64 *rowoff, const u64 *a_meta, unsigned n, int C,
                                                      double ratio, TailJob *jobs, int in_scratch) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const u64 *a_row_off = a_meta + n, *a_row_stride = a_meta + 2 * (size_t)n;
    u64 op = 0, l1 = ~0ull, l2 = ~0ull;  // where last[1], last[2] live in the decoded rows (~0: the initial 0)
    TailJob *dst = jobs + fbase[s] * (u64)C;
    for (unsigned f = 0; f < nframes[s]; f++) {
        const FrameRec fr = frames[fbase[s] + f];
        const int nout = (int)floor((double)fr.bs * ratio);
        for (int c = 0; c < C; c++) {
            TailJob j;
            j.src_off = in_scratch ? fr.scratch + (u64)c * (u64)fr.bs : rowoff[(size_t)s * C + c] + fr.sample_off;   // (the fused decoder's frames are read where they lie)
            j.last_off = l2; j.m1_off = l1;
            j.out_off = a_row_off[s] + (u64)c * a_row_stride[s] + op;
            j.src_cstride = j.last_cstride = j.out_cstride = 0;
            j.n = fr.bs; j.nout = nout; j.pad = 0;
            *dst++ = j;
            // for a one-sample block src[#src-1] is src[0] = the old last[2]  (:3183)
            if (fr.bs >= 2) { l1 = j.src_off + (u64)fr.bs - 2; l2 = l1 + 1; }
            else if (fr.bs == 1) { l1 = l2; l2 = j.src_off; }
        }
        op += (u64)nout;
    }
}

int stream_flac(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, int interp, int, int dtype, aukit_audio **out, aukit_chunks **chunks_out) {
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "stream.flac: bad interpolation");   // (sinc: the reference-order kernels below — aukit.defaultInterpolation = "sinc" is legal at :3156)
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "stream.flac output must be AUKIT_F64 or AUKIT_F32");
    FlacDecoded D;
    static const bool TT = getenv("AUKIT_HOST_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) { if (TT) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[stream.flac host] %-12s %8.1f us\n", w, std::chrono::duration<double, std::micro>(t - T0).count()); T0 = t; } };
    int rc = flac_decode_rows(ctx, in, D, true);
    if (rc) return rc;
    lap("decode rows");
    const int C = D.channels;
    const double ratio = 48000 / D.rate;                                      // :3154
    const double lp_alpha = 1 - std::exp(-(D.rate / 96000) * 2 * M_PI);       // :3155
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0); ck->status.assign(in->n, 0); ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0);
    uint64_t max_nout = 0, sum_nout = 0, njobs = 0;
    const bool brief = !D.brief.empty();
    for (uint32_t s = 0; s < in->n; s++) {
        uint64_t l = 0;
        const size_t nf = brief ? D.nframes[s] : D.frames[s].size();
        for (size_t f = 0; f < nf; f++) {
            const int bsz = brief ? (int)D.brief[D.fbase[s] + f].x : D.frames[s][f].second;
            const uint64_t no = (uint64_t)std::floor((double)bsz * ratio);   // every frame ends up in some call's chunk
            l += no;
            max_nout = std::max(max_nout, no);
        }
        lens[s] = l;
        sum_nout += l * (uint64_t)C; njobs += (uint64_t)C * nf;
    }
    lap("lens");
    // the chunk table (which frames an iterator call returns) is host work nothing on the device waits for: it is built AFTER the tail kernel has been
    // launched (round 4: 0.3 ms of an 8.2 ms call on 1024 streams during which the GPU stood idle)
    auto build_chunks = [&]() {
        std::vector<uint32_t> clen;
        std::vector<double> cpos;
        std::vector<uint64_t> cend;
        std::vector<uint32_t> first(in->n + 1, 0);
        for (uint32_t s = 0; s < in->n; s++) {
            ck->length_seconds[s] = D.info[s].nsamples / D.rate;
            // iterator calls: frames accumulate while #chunk[1] < sampleRate; an error or the end of data kills the coroutine (errors are swallowed)
            size_t f = 0;
            bool dead = false;
            double pos = ctx->sb_pos;   // (the rest of a stream behind a bounded reader-function handle: the position the dropped calls had summed up)
            first[s] = (uint32_t)clen.size();
            while (!dead) {
                uint64_t got = 0;
                while ((double)got < D.rate) {
                    if (f >= D.frames[s].size()) { dead = true; break; }
                    got += (uint64_t)std::floor((double)D.frames[s][f].second * ratio);
                    f++;
                }
                pos = pos + (double)got / 48000;                                  // :3188
                clen.push_back((uint32_t)got);
                cpos.push_back(pos);
                cend.push_back(f > 0 && s < D.frame_end.size() && f - 1 < D.frame_end[s].size() ? (uint64_t)D.frame_end[s][f - 1] : 0ull);
            }
            ck->nchunks[s] = (uint32_t)clen.size() - first[s];
            ck->max_chunks = std::max(ck->max_chunks, ck->nchunks[s]);
        }
        first[in->n] = (uint32_t)clen.size();
        const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
        ck->lens.assign((size_t)ck->n * mc, 0);
        ck->pos.assign((size_t)ck->n * mc, 0);
        for (uint32_t s = 0; s < in->n; s++)
            for (uint32_t k = 0; k < ck->nchunks[s]; k++) { ck->lens[(size_t)s * mc + k] = clen[first[s] + k]; ck->pos[(size_t)s * mc + k] = cpos[first[s] + k]; }
        if (!D.frame_end.empty()) {   // where a bounded handle may cut (stream_handle.hip)
            ck->in_end.assign((size_t)ck->n * mc, 0);
            ck->in_first.assign(ck->n, 0);
            for (uint32_t s = 0; s < in->n; s++) {
                ck->in_first[s] = D.first_frame[s];
                for (uint32_t k = 0; k < ck->nchunks[s]; k++) ck->in_end[(size_t)s * mc + k] = cend[first[s] + k];
            }
        }
    };
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, C, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    lap("audio_prepare");
    {   // round 3: jobs written on the device from the decoder's frame records, one launch from the decoded rows (k_iir_tail, stream_tail.hip)
        const int rk = D.wide ? TAIL_ROWS_F64 : TAIL_ROWS_I32;
        const double fullv = std::ldexp(1.0, D.depth);
        if (njobs && D.d_frames && iir_tail_served(ctx, TAIL_FLAC, rk, 1, D.rate, fullv, interp, dtype, max_nout)) {
            if ((rc = ctx->misc_buf.ensure(njobs * sizeof(TailJob) + (size_t)in->n * 4 + 64))) { delete ck; return rc; }
            TailJob *dj = reinterpret_cast<TailJob *>(ctx->misc_buf.p);
            unsigned *dnf = reinterpret_cast<unsigned *>(dj + njobs);
            if ((rc = h2d_table(ctx, dnf, D.nframes.data(), (size_t)in->n * 4))) { delete ck; return rc; }
            hipLaunchKernelGGL(k_flac_tail_jobs, dim3((in->n + 63) / 64), dim3(64), 0, ctx->stream, D.d_frames, D.d_fbase, dnf, D.d_rowoff, reinterpret_cast<const u64 *>(a->d_meta), in->n, C,
                               ratio, dj, D.in_scratch ? 1 : 0);
            if (hipGetLastError() != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "k_flac_tail_jobs launch failed"); }
            lap("plan + jobs");
            int trc = AUKIT_OK;
            if (rk == TAIL_ROWS_I32 && dtype == AUKIT_F32 &&
                rs_onepole_jobs_try_dev(ctx, D.in_scratch ? ctx->tmp_buf3.p : ctx->tmp_buf.p, fullv, dj, njobs, D.rate, interp, lp_alpha, reinterpret_cast<float *>(a->dev),
                                        in->total() + sum_nout * dtype_size(dtype), "k_rs_onepole<flac>", &trc)) {   // the tile chain of flac_tail.hip (state carried from tile to tile, a workgroup takes frame after frame)
                if (trc) { delete ck; return trc; }
                lap("tail launch");
                if (brief) frames_from_brief(D);
                build_chunks();
                lap("chunk table");
                if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
                return AUKIT_OK;
            }
            if (iir_tail_try_dev(ctx, TAIL_FLAC, rk, D.in_scratch ? ctx->tmp_buf3.p : ctx->tmp_buf.p, fullv, dj, njobs, max_nout, sum_nout, 1, D.rate, interp, dtype, a->dev, in->total() + sum_nout * dtype_size(dtype),
                                 "k_iir_tail<flac>", &trc)) {
                if (trc) { delete ck; return trc; }
                lap("tail launch");
                if (brief) frames_from_brief(D);
                build_chunks();
                lap("chunk table");
                if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
                return AUKIT_OK;
            }
        }
    }
    if (brief) frames_from_brief(D);
    build_chunks();
    if ((rc = flac_rows_materialize(ctx, D))) { delete ck; return rc; }
    std::vector<FsJob> jobs;
    uint64_t nouts = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        uint64_t op = 0;
        u64 l1 = ~0ull, l2 = ~0ull;  // where last[1], last[2] live in the decoded rows (~0: the initial 0)
        for (auto &fr : D.frames[s]) {
            const int nout = (int)std::floor((double)fr.second * ratio);
            for (int c = 0; c < C; c++) {
                FsJob j;
                j.src_off = D.row_off[(size_t)s * C + c] + fr.first;
                j.last_off = l2;
                j.m1_off = l1;
                j.out_off = a->row_off[s] + (uint64_t)c * a->row_stride[s] + op;
                j.blocksize = fr.second;
                j.nout = nout;
                jobs.push_back(j);
                // last = {src[#src-1], src[#src]}, shared across channels (Q14); for a one-sample block src[#src-1] is src[0] = the old last[2]
                if (fr.second >= 2) { l1 = j.src_off + (uint64_t)fr.second - 2; l2 = l1 + 1; }
                else if (fr.second == 1) { l1 = l2; l2 = j.src_off; }
            }
            op += (uint64_t)nout;
            nouts += (uint64_t)nout * C;
        }
    }
    if (!jobs.empty()) {
        if ((rc = upload_table(ctx, ctx->seg_buf, jobs.data(), jobs.size() * sizeof(FsJob)))) { delete ck; return rc; }
        const uint64_t maxn = 1ull << 17;
        const int exact = exact_div_verified(ctx, ratio, maxn) ? 1 : 0;
        const unsigned grid = (unsigned)((jobs.size() + 63) / 64);
        const FsJob *dj = reinterpret_cast<const FsJob *>(ctx->seg_buf.p);
        const double full = std::ldexp(1.0, D.depth);
        if ((rc = ctx_begin_kernel(ctx))) { delete ck; return rc; }
        uint64_t scr_elems = 0, max_nout = 0;
        std::vector<uint64_t> scr_off(jobs.size());
        for (size_t k = 0; k < jobs.size(); k++) { scr_off[k] = scr_elems; scr_elems += ((uint64_t)jobs[k].nout + 1) & ~1ull; max_nout = std::max<uint64_t>(max_nout, (uint64_t)jobs[k].nout); }
        if (scr_elems * 8 <= (48ull << 30) && max_nout && !getenv("AUKIT_FLAC_STREAM_ONE_PASS")) {  // two passes
            if ((rc = ctx->tmp_buf3.ensure((size_t)scr_elems * 8 + 64))) { delete ck; return rc; }
            if ((rc = upload_table(ctx, ctx->misc_buf, scr_off.data(), scr_off.size() * 8))) { delete ck; return rc; }
            const u64 *dso = reinterpret_cast<const u64 *>(ctx->misc_buf.p);
            double *scr = reinterpret_cast<double *>(ctx->tmp_buf3.p);
            for (size_t first = 0; first < jobs.size(); first += 65535) {
                const dim3 g1((unsigned)std::min<uint64_t>((max_nout + 255) / 256, 1024), (unsigned)std::min<size_t>(65535, jobs.size() - first));
#define AUKIT_FI2(I, R) hipLaunchKernelGGL((k_flac_stream_interp<I, R>), g1, dim3(256), 0, ctx->stream, dj + first, dso + first, reinterpret_cast<const R *>(ctx->tmp_buf.p), full, scr, ratio, 1.0 / ratio, exact, ctx->sinc_w)
#define AUKIT_FI(I) do { if (D.wide) AUKIT_FI2(I, double); else AUKIT_FI2(I, int); } while (0)
                if (interp == 0) AUKIT_FI(0); else if (interp == 1) AUKIT_FI(1); else if (interp == 2) AUKIT_FI(2); else AUKIT_FI(3);
#undef AUKIT_FI
#undef AUKIT_FI2
            }
#define AUKIT_FR(T, R) hipLaunchKernelGGL((k_flac_stream_iir<T, R>), dim3(grid), dim3(64), 0, ctx->stream, dj, dso, (u64)jobs.size(), reinterpret_cast<const R *>(ctx->tmp_buf.p), full, scr, reinterpret_cast<T *>(a->dev), lp_alpha)
            if (dtype == AUKIT_F64) { if (D.wide) AUKIT_FR(double, double); else AUKIT_FR(double, int); }
            else { if (D.wide) AUKIT_FR(float, double); else AUKIT_FR(float, int); }
#undef AUKIT_FR
        } else {
#define AUKIT_FS2(I, T, R) hipLaunchKernelGGL((k_flac_stream<I, T, R>), dim3(grid), dim3(64), 0, ctx->stream, dj, (u64)jobs.size(), reinterpret_cast<const R *>(ctx->tmp_buf.p), full, reinterpret_cast<T *>(a->dev), ratio, 1.0 / ratio, exact, lp_alpha, ctx->sinc_w)
#define AUKIT_FS(I, T) do { if (D.wide) AUKIT_FS2(I, T, double); else AUKIT_FS2(I, T, int); } while (0)
        if (dtype == AUKIT_F64) { if (interp == 0) AUKIT_FS(0, double); else if (interp == 1) AUKIT_FS(1, double); else if (interp == 2) AUKIT_FS(2, double); else AUKIT_FS(3, double); }
        else { if (interp == 0) AUKIT_FS(0, float); else if (interp == 1) AUKIT_FS(1, float); else if (interp == 2) AUKIT_FS(2, float); else AUKIT_FS(3, float); }
#undef AUKIT_FS
#undef AUKIT_FS2
        }
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_flac_stream", in->total() + nouts * dtype_size(dtype)))) { delete ck; return rc; }
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

}  // namespace aukit
