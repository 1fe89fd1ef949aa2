// flac.hip — frame-parallel FLAC decode (aukit.lua:311-619: decodeFLAC, a port of Nayuki's simple decoder) on gfx950.
//
// FLAC frames carry no length, and the reference simply decodes them one after another (aukit.lua:615), ignoring
// the CRCs (:553, :557).  To decode frames in parallel without changing which bytes are treated as frames:
//   1. k_flac_header   one lane per stream walks the metadata blocks (:573-606) → STREAMINFO + first frame byte;
//   2. k_flac_find     every byte position that starts with the 14-bit sync code 0x3FFE (:518) is a candidate;
//   3. k_flac_parse    one lane per candidate parses the frame exactly like decodeFrame (Rice prefix by clz on a
//                      64-bit window) WITHOUT storing samples → end position, block size, subframe bit offsets, status;
//   4. host            follows end → start links from the first frame: precisely the frames the serial decoder visits;
//   5. k_flac_subframe one lane per (frame, subframe): residual decode + LPC restore on the fly (history in VGPRs);
//   6. k_flac_finish   stereo decorrelation (:482-497), wrap and / 2^depth (:501-507, Q14).
// Integer work throughout (exact); only the final division produces the reference's doubles.
#include <algorithm>
#include <map>
#include "resample.h"

namespace aukit {

int audio_from_int_rows(aukit_ctx *ctx, int src_kind, const void *rows_dev, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len,
                        uint32_t n, int channels, double rate, double new_rate, int interp, bool do_resample, int dtype, double norm_pos,
                        double norm_neg, aukit_audio **out);

enum FlacErr { FE_OK = 0, FE_EOF_START = 1 /* readByte() == nil at a frame start: clean end */, FE_NIL = 2, FE_SYNC = 3, FE_BLOCKSIZE = 4,
               FE_CHAN = 5, FE_SUBTYPE = 6, FE_RESMETHOD = 7, FE_PARTITION = 8,
               FE_LIMIT = 9 /* parse pass only: the candidate ran past its bit budget; re-parsed without a budget if the chain needs it */ };
static const char *flac_err_msg(int e) {
    switch (e) {
    case FE_NIL: return "attempt to perform arithmetic on a nil value";
    case FE_SYNC: return "Sync code expected";
    case FE_BLOCKSIZE: return "Reserved block size";
    case FE_CHAN: return "Reserved channel assignment";
    case FE_SUBTYPE: return "Reserved subframe type";
    case FE_RESMETHOD: return "Reserved residual coding method";
    case FE_PARTITION: return "Block size not divisible by number of Rice partitions";
    }
    return "FLAC decode error";
}

// MSB-first bit reader over [first_bit, end_bit) of the batch buffer.  Three big-endian 64-bit words are kept in
// registers (current, next, next-but-one): a Rice-coded sample consumes ≈15 bits, so one 8-byte load is issued every
// ≈4 samples and it is not needed until the following word crossing — the load latency stays off the decode chain
// (the first version re-loaded 16 bytes for every bit-field: 72 + 82 ms per 3.6 GB batch, see profiles/).
struct Bits {
    const unsigned long long *w0;   // 8-byte aligned base at or below the batch data
    unsigned long long first, end;  // bit offsets (relative to w0) of the BitInputStream start and of the end of the string
    unsigned long long pos;
    unsigned long long cur, nxt, nxt2, wi;
    unsigned long long limit;       // parse pass: give up (FE_LIMIT) beyond this bit — bounds the work of false sync candidates
    int kind;                       // parse pass: code-path class of the last subframe (0 const, 1 verbatim, 2/3/4 order <= 4/12/32, 5 wide)
    int eof;
};
AUKIT_DEV unsigned long long be64(unsigned long long v) { return __builtin_bswap64(v); }
AUKIT_DEV void bits_seek(Bits &b, unsigned long long pos) {
    b.pos = pos;
    b.wi = pos >> 6;
    b.cur = be64(b.w0[b.wi]);
    b.nxt = be64(b.w0[b.wi + 1]);
    b.nxt2 = be64(b.w0[b.wi + 2]);
}
AUKIT_DEV unsigned long long peek(const Bits &b) {
    const unsigned s = (unsigned)b.pos & 63;
    return s ? (b.cur << s) | (b.nxt >> (64 - s)) : b.cur;
}
AUKIT_DEV void skip(Bits &b, unsigned n) {  // n <= 64
    b.pos += n;
    if ((b.pos >> 6) != b.wi) {
        b.wi++;
        b.cur = b.nxt;
        b.nxt = b.nxt2;
        b.nxt2 = be64(b.w0[b.wi + 2]);
    }
}
AUKIT_DEV unsigned long long peek_at(const Bits &b, unsigned long long pos) {  // uncached, any position
    const unsigned long long wi = pos >> 6;
    const unsigned s = (unsigned)pos & 63;
    const unsigned long long hi = be64(b.w0[wi]);
    if (s == 0) return hi;
    return (hi << s) | (be64(b.w0[wi + 1]) >> (64 - s));
}
// BitInputStream.readUint(n)  :351-364, n <= 57.  n >= 32 keeps the stale high bits of the 44-bit buffer like the Lua does.
AUKIT_DEV long long read_uint(Bits &b, int n) {
    if (n == 0) return 0;
    if (b.pos + (unsigned long long)n > b.end) { b.eof = 1; return 0; }  // str_byte → nil
    if (n < 32) {
        const unsigned long long v = peek(b) >> (64 - n);
        skip(b, (unsigned)n);
        return (long long)v;
    }
    const unsigned long long after = b.pos + n;
    const int L = (int)((8 - ((after - b.first) & 7)) & 7);  // bits left in the buffer after this read
    int width = 44 - L;
    unsigned long long s = after - (unsigned long long)width;
    if (after < b.first + (unsigned long long)width) { width = (int)(after - b.first); s = b.first; }
    const unsigned long long v = peek_at(b, s) >> (64 - width);
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
        const unsigned long long w = peek(b);
        const unsigned long long avail = b.end - b.pos;
        int z = w ? __builtin_clzll(w) : 64;
        if ((unsigned long long)z >= avail) { b.eof = 1; return 0; }  // ran off the end inside the unary prefix
        if (z < 64) { val += z; skip(b, (unsigned)z + 1); break; }
        val += 64;
        skip(b, 64);
    }
    val = (val << param) + read_uint(b, param);
    return (val & 1) ? -(val >> 1) - 1 : (val >> 1);
}

struct FrameInfo {                 // result of parsing one candidate
    unsigned long long start_bit;  // relative to w0
    unsigned long long end_bit;    // first bit after the frame's CRC-16 (byte aligned)
    unsigned long long sub_bit[AUKIT_MAX_CHANNELS];
    int blocksize, chan_asgn, status, pad;
    unsigned char kind[AUKIT_MAX_CHANNELS];
};

struct FlacStreamInfo { unsigned long long first_byte; double rate, nsamples; int channels, depth, status, pad; };

// decodeFLAC header + metadata blocks  :569-606
__global__ __launch_bounds__(64) void k_flac_header(const unsigned char *src, const unsigned long long *off, unsigned n, FlacStreamInfo *out) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const unsigned char *p = src + off[s];
    const unsigned long long nb = off[s + 1] - off[s];
    FlacStreamInfo r{};
    r.status = 0;
    if (nb < 4) { r.status = 1; out[s] = r; return; }                                                    // nil arithmetic in intunpack
    if (!(p[0] == 0x66 && p[1] == 0x4C && p[2] == 0x61 && p[3] == 0x43)) { r.status = 2; out[s] = r; return; }  // "Invalid magic string"
    unsigned long long pos = 4;
    bool last = false, have = false;
    while (!last) {
        if (pos + 4 > nb) { r.status = 1; out[s] = r; return; }
        const int t = p[pos];
        last = (t & 0x80) != 0;
        const int type = t & 0x7F;
        const unsigned long long length = (unsigned long long)p[pos + 1] << 16 | (unsigned long long)p[pos + 2] << 8 | p[pos + 3];
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

// frame-header length as decodeFrame walks it (:518-553) and CRC-8 (poly 0x07) over it; false when the header does not fit
AUKIT_DEV bool flac_header_crc_ok(const unsigned char *src, unsigned long long p, unsigned long long end) {
    if (p + 6 > end) return false;
    const unsigned b2 = src[p + 2];
    const unsigned bsc = b2 >> 4, src_code = b2 & 15;
    unsigned long long idx = p + 4;
    const unsigned t = src[idx];
    int lead = 0;
    for (int i = 7; i >= 0; i--) { if (!(t & (1u << i))) break; lead++; }
    idx += 1 + (lead > 1 ? lead - 1 : 0);
    if (bsc == 6) idx += 1; else if (bsc == 7) idx += 2;
    if (src_code == 12) idx += 1; else if (src_code == 13 || src_code == 14) idx += 2;
    if (idx >= end) return false;
    unsigned crc = 0;
    for (unsigned long long q = p; q < idx; q++) {
        crc ^= src[q];
        for (int k = 0; k < 8; k++) crc = (crc & 0x80) ? ((crc << 1) ^ 0x07) & 0xFF : (crc << 1) & 0xFF;
    }
    return crc == src[idx];
}

struct Cand { unsigned stream; unsigned pad; unsigned long long byte; };  // byte: absolute in the batch

__global__ __launch_bounds__(256) void k_flac_find(const unsigned char *src, const unsigned long long *off, const FlacStreamInfo *info, unsigned n,
                                                  Cand *cands, unsigned long long cap, unsigned long long *count) {
    const unsigned s = blockIdx.y;
    if (info[s].status) return;
    const unsigned long long b0 = off[s] + info[s].first_byte, b1 = off[s + 1];
    for (unsigned long long p = b0 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; p + 1 < b1; p += (unsigned long long)gridDim.x * 256) {
        if (src[p] == 0xFF && (src[p + 1] & 0xFC) == 0xF8) {  // temp * 64 + readUint(6) == 0x3FFE  :518
            // Speed-only filter: keep candidates whose header CRC-8 checks out.  The reference ignores the CRC (:553), so a
            // real frame with a damaged CRC is still decoded: the host parses any chain position that is not in this list on demand.
            if (!flac_header_crc_ok(src, p, b1)) continue;
            const unsigned long long k = atomicAdd(count, 1ull);
            if (k < cap) cands[k] = Cand{s, 0, p};
        }
    }
}

// decodeResiduals (:380-409) + restoreLinearPrediction (:411-419) fused: every residual is turned into a sample at once,
// with the last MAXO samples in registers (hist[j] = result[i-1-j]).  STORE = false only advances the bit reader.
template <bool STORE, int MAXO, typename H>
AUKIT_DEV int flac_residuals(Bits &b, int order, int blocksize, int lshift, H *hist, const H *coef, long long mul, double *out) {
    const int method = (int)read_uint(b, 2);
    if (b.eof) return FE_NIL;
    if (method >= 2) return FE_RESMETHOD;
    const int param_bits = method == 0 ? 4 : 5, escape = method == 0 ? 0xF : 0x1F;
    const int porder = (int)read_uint(b, 4);
    const int nparts = 1 << porder;
    if (blocksize % nparts != 0) return FE_PARTITION;
    const int psize = blocksize / nparts;
    for (int pi = 0; pi < nparts; pi++) {
        const int start = pi * psize + (pi == 0 ? order : 0), endd = (pi + 1) * psize;
        const int param = (int)read_uint(b, param_bits);
        int nbits = 0;
        const bool esc = param >= escape;
        if (esc) nbits = (int)read_uint(b, 5);
        if (b.eof) return FE_NIL;
        if (!STORE && b.pos > b.limit) return FE_LIMIT;
        for (int j = start; j < endd; j++) {
            const long long r = esc ? read_sint(b, nbits) : read_rice(b, param);
            if constexpr (STORE) {
                long long sum = 0;
#pragma unroll
                for (int q = 0; q < MAXO; q++) sum += (long long)hist[q] * (long long)coef[q];
                const long long pred = lshift >= 0 ? (sum >> lshift) : (sum << (-lshift));  // floor(sum / 2^shift)
                const long long v = r + pred;
                out[j] = (double)(v * mul);
#pragma unroll
                for (int q = MAXO - 1; q > 0; q--) hist[q] = hist[q - 1];
                hist[0] = (H)v;
            }
        }
        if (b.eof) return FE_NIL;
    }
    return FE_OK;
}

template <int MAXO, typename H>
AUKIT_DEV int flac_predict(Bits &b, int type, int order, int depth, int blocksize, long long mul, double *out) {
    H hist[MAXO], coef[MAXO];
#pragma unroll
    for (int j = 0; j < MAXO; j++) { hist[j] = 0; coef[j] = 0; }
    for (int i = 0; i < order; i++) {  // warm-up samples :422-424 / :430-432
        const long long v = read_sint(b, depth);
        if (i < blocksize) out[i] = (double)(v * mul);
#pragma unroll
        for (int j = MAXO - 1; j > 0; j--) hist[j] = hist[j - 1];
        hist[0] = (H)v;
    }
    int lshift = 0;
    if (type >= 32) {
        const int precision = (int)read_uint(b, 4) + 1;
        lshift = (int)read_sint(b, 5);
        for (int i = 0; i < order; i++) {
            const long long c = read_sint(b, precision);
#pragma unroll
            for (int j = 0; j < MAXO; j++) if (j == i) coef[j] = (H)c;
        }
    } else {  // FIXED_PREDICTION_COEFFICIENTS  :334-340
        const int fc[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < MAXO) coef[j] = (H)fc[order][j];
    }
    if (b.eof) return FE_NIL;
    return flac_residuals<true, MAXO, H>(b, order, blocksize, lshift, hist, coef, mul, out);
}

// decodeSubframe (:443-470) on a bit reader.  STORE = false: parse only; STORE = true: residual + prediction restored on the fly.
template <bool STORE>
AUKIT_DEV int flac_subframe(Bits &b, int depth, int blocksize, double *out) {
    read_uint(b, 1);
    const int type = (int)read_uint(b, 6);
    int shift = (int)read_uint(b, 1);
    if (shift == 1) {  // unary wasted-bits count  :447-449
        for (;;) {
            const long long bit = read_uint(b, 1);
            if (b.eof) return FE_NIL;
            if (bit) break;
            shift++;
        }
    }
    if (b.eof) return FE_NIL;
    depth -= shift;
    if (depth < 0 || depth > 57) return FE_NIL;  // 2^(n-1) with a negative n misbehaves in the Lua too; treat as malformed
    const long long mul = 1ll << shift;
    if (type == 0) {
        b.kind = 0;
        const long long c = read_sint(b, depth);
        if (b.eof) return FE_NIL;
        if (STORE) for (int i = 0; i < blocksize; i++) out[i] = (double)(c * mul);
        return FE_OK;
    }
    if (type == 1) {
        b.kind = 1;
        for (int i = 0; i < blocksize; i++) {
            if (!STORE && (i & 255) == 0 && b.pos > b.limit) return FE_LIMIT;
            const long long v = read_sint(b, depth);
            if (STORE) out[i] = (double)(v * mul);
        }
        return b.eof ? FE_NIL : FE_OK;
    }
    int order, lshift = 0;
    if (type >= 8 && type <= 12) order = type - 8;
    else if (type >= 32 && type <= 63) order = type - 31;
    else return FE_SUBTYPE;
    if (!STORE) {  // parse only: consume warm-up, coefficients and residuals
        b.kind = depth > 31 ? 5 : (order <= 4 ? 2 : (order <= 12 ? 3 : 4));
        for (int i = 0; i < order; i++) read_sint(b, depth);
        if (type >= 32) {
            const int precision = (int)read_uint(b, 4) + 1;
            read_sint(b, 5);
            for (int i = 0; i < order; i++) read_sint(b, precision);
        }
        if (b.eof) return FE_NIL;
        return flac_residuals<false, 1, int>(b, order, blocksize, 0, nullptr, nullptr, 1, nullptr);
    }
    // 8/16/24-bit audio: history and coefficients fit int32 (one v_mad_i64_i32 per tap); 32-bit audio takes the int64 path
    if (depth <= 31) {
        if (order <= 4) return flac_predict<4, int>(b, type, order, depth, blocksize, mul, out);
        if (order <= 12) return flac_predict<12, int>(b, type, order, depth, blocksize, mul, out);
        return flac_predict<32, int>(b, type, order, depth, blocksize, mul, out);
    }
    return flac_predict<32, long long>(b, type, order, depth, blocksize, mul, out);
}

struct FlacGlobals {
    const unsigned char *src;           // batch data
    const unsigned long long *w0;       // aligned base
    unsigned long long base_bit;        // bit offset of src relative to w0
    const unsigned long long *off;
    const FlacStreamInfo *info;
};

// decodeFrame header + parse-only subframes  :510-557
__global__ __launch_bounds__(64) void k_flac_parse(const FlacGlobals G, const Cand *cands, unsigned long long ncand, FrameInfo *out, int limit_factor) {
    const unsigned long long ci = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (ci >= ncand) return;
    const Cand c = cands[ci];
    const FlacStreamInfo si = G.info[c.stream];
    Bits b;
    b.w0 = G.w0;
    b.first = G.base_bit + 8 * (G.off[c.stream] + si.first_byte);
    b.end = G.base_bit + 8 * G.off[c.stream + 1];
    b.eof = 0;
    b.limit = ~0ull;
    b.kind = 0;
    bits_seek(b, G.base_bit + 8 * c.byte);
    FrameInfo f{};
    f.start_bit = b.pos;
    f.status = FE_OK;
    const long long t0 = read_uint(b, 8);
    if (b.eof) { f.status = FE_EOF_START; out[ci] = f; return; }
    const long long sync = t0 * 64 + read_uint(b, 6);
    if (b.eof) { f.status = FE_NIL; out[ci] = f; return; }
    if (sync != 0x3FFE) { f.status = FE_SYNC; out[ci] = f; return; }
    read_uint(b, 2);
    const int bsc = (int)read_uint(b, 4), src_code = (int)read_uint(b, 4);
    f.chan_asgn = (int)read_uint(b, 4);
    read_uint(b, 4);
    const int t = (int)read_uint(b, 8);
    if (b.eof) { f.status = FE_NIL; out[ci] = f; return; }
    int t2 = -1;
    for (int i = 7; i >= 0; i--) { if (!(t & (1 << i))) break; t2++; }
    for (int i = 1; i <= t2; i++) read_uint(b, 8);
    int bs;
    if (bsc == 1) bs = 192;
    else if (bsc >= 2 && bsc <= 5) bs = 576 << (bsc - 2);
    else if (bsc == 6) bs = (int)read_uint(b, 8) + 1;
    else if (bsc == 7) bs = (int)read_uint(b, 16) + 1;
    else if (bsc >= 8) bs = 256 << (bsc - 8);
    else { f.status = FE_BLOCKSIZE; out[ci] = f; return; }
    if (src_code == 12) read_uint(b, 8);
    else if (src_code == 13 || src_code == 14) read_uint(b, 16);
    read_uint(b, 8);  // CRC-8, ignored :553
    if (b.eof) { f.status = FE_NIL; out[ci] = f; return; }
    f.blocksize = bs;
    // bit budget: `limit_factor` times the size of an all-VERBATIM frame (no encoder emits a bigger one; the chain re-parses without it if needed)
    b.limit = limit_factor > 0 ? b.pos + (unsigned long long)limit_factor * (unsigned long long)bs * (unsigned long long)si.channels * (unsigned long long)(si.depth + 2) + 4096 : ~0ull;
    const int nch = si.channels, depth = si.depth;
    int st = FE_OK;
    if (f.chan_asgn <= 7) {
        for (int ch = 0; ch < nch && st == FE_OK; ch++) { f.sub_bit[ch] = b.pos; st = flac_subframe<false>(b, depth, bs, nullptr); f.kind[ch] = (unsigned char)b.kind; }
    } else if (f.chan_asgn <= 10) {
        if (nch < 2) st = FE_NIL;
        else {
            f.sub_bit[0] = b.pos;
            st = flac_subframe<false>(b, depth + (f.chan_asgn == 9 ? 1 : 0), bs, nullptr);
            f.kind[0] = (unsigned char)b.kind;
            if (st == FE_OK) { f.sub_bit[1] = b.pos; st = flac_subframe<false>(b, depth + (f.chan_asgn == 9 ? 0 : 1), bs, nullptr); f.kind[1] = (unsigned char)b.kind; }
        }
    } else st = FE_CHAN;
    if (st != FE_OK) { f.status = st; out[ci] = f; return; }
    b.pos = b.first + (((b.pos - b.first) + 7) & ~7ull);  // alignToByte
    // readUint(16): a nil here is discarded, the NEXT readByte then returns nil  :557
    b.pos = (b.pos + 16 <= b.end) ? b.pos + 16 : b.end;
    f.end_bit = b.pos;
    out[ci] = f;
}

struct SubJob { unsigned long long bit; unsigned long long out_off; unsigned stream; int depth, blocksize, pad; };

__global__ __launch_bounds__(64) void k_flac_subframe(const FlacGlobals G, const SubJob *jobs, unsigned long long njobs, double *out, int *err) {
    const unsigned long long j = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= njobs) return;
    const SubJob job = jobs[j];
    const FlacStreamInfo si = G.info[job.stream];
    Bits b;
    b.w0 = G.w0;
    b.first = G.base_bit + 8 * (G.off[job.stream] + si.first_byte);
    b.end = G.base_bit + 8 * G.off[job.stream + 1];
    b.eof = 0;
    b.limit = ~0ull;
    b.kind = 0;
    bits_seek(b, job.bit);
    const int st = flac_subframe<true>(b, job.depth, job.blocksize, out + job.out_off);
    if (st != FE_OK) atomicCAS(err, 0, st);
}

struct FinJob { unsigned long long off0, off1; int blocksize, chan_asgn; };  // off0/off1: the frame's first two channel blocks
// decodeSubframes tail  :482-507: decorrelate, wrap, / 2^depth.  rows[] lists every (frame, channel) block for the plain wrap pass.
__global__ __launch_bounds__(256) void k_flac_finish(const FinJob *jobs, unsigned long long njobs, double *data, int depth) {
    const FinJob job = jobs[blockIdx.x];
    const double half = ldexp(1.0, depth - 1), full = ldexp(1.0, depth);
    for (int i = threadIdx.x; i < job.blocksize; i += 256) {
        if (job.chan_asgn >= 8) {
            double a = data[job.off0 + i], s = data[job.off1 + i];
            if (job.chan_asgn == 8) s = a - s;                                   // left/side
            else if (job.chan_asgn == 9) a = a + s;                              // side/right
            else { const double side = s; const double right = a - floor(side / 2); s = right; a = right + side; }  // mid/side
            if (a >= half) a -= full;
            if (s >= half) s -= full;
            data[job.off0 + i] = a / full;
            data[job.off1 + i] = s / full;
        } else {
            double a = data[job.off0 + i];
            if (a >= half) a -= full;
            data[job.off0 + i] = a / full;
        }
    }
}

struct FlacDecoded {
    std::vector<FlacStreamInfo> info;
    std::vector<std::vector<std::pair<uint64_t, int>>> frames;  // per stream: (sample offset, blocksize) of every decoded frame, in order
    std::vector<int> status;                                     // per stream: 0 = clean end, else FlacErr of the frame that failed
    std::vector<uint64_t> row_off, row_len;                      // (stream, channel) rows of doubles in ctx->tmp_buf
    int channels = 0, depth = 0;
    double rate = 0;
};

static int flac_decode_rows(aukit_ctx *ctx, const aukit_batch *in, FlacDecoded &D) {
    const uint32_t n = in->n;
    if (n == 0) return fail(AUKIT_E_ARG, "empty batch");
    int rc;
    // -- 1. stream headers
    DevBuf &hb = ctx->misc_buf;
    if ((rc = hb.ensure((size_t)n * sizeof(FlacStreamInfo)))) return rc;
    hipLaunchKernelGGL(k_flac_header, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off), n,
                       reinterpret_cast<FlacStreamInfo *>(hb.p));
    AUKIT_HIP_CHECK(hipGetLastError());
    D.info.resize(n);
    AUKIT_HIP_CHECK(hipMemcpyAsync(D.info.data(), hb.p, (size_t)n * sizeof(FlacStreamInfo), hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
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
    // -- 2. sync candidates
    const uint64_t cap = in->total() / 32 + 4096;
    DevBuf &cb = ctx->tmp_buf2;
    if ((rc = cb.ensure(cap * sizeof(Cand) + 64))) return rc;
    unsigned long long *d_count = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(cb.p) + cap * sizeof(Cand));
    AUKIT_HIP_CHECK(hipMemsetAsync(d_count, 0, 8, ctx->stream));
    uint64_t maxlen = 1;
    for (uint32_t s = 0; s < n; s++) maxlen = std::max<uint64_t>(maxlen, in->off[s + 1] - in->off[s]);
    hipLaunchKernelGGL(k_flac_find, dim3((unsigned)std::min<uint64_t>((maxlen + 255) / 256, 2048), n), dim3(256), 0, ctx->stream, in->data(),
                       reinterpret_cast<const unsigned long long *>(in->d_off), reinterpret_cast<const FlacStreamInfo *>(hb.p), n, reinterpret_cast<Cand *>(cb.p), cap, d_count);
    AUKIT_HIP_CHECK(hipGetLastError());
    unsigned long long ncand = 0;
    AUKIT_HIP_CHECK(hipMemcpyAsync(&ncand, d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ncand > cap) return fail(AUKIT_E_UNSUPPORTED, "too many FLAC sync candidates (%llu)", ncand);
    std::vector<Cand> cands(ncand);
    if (ncand) AUKIT_HIP_CHECK(hipMemcpy(cands.data(), cb.p, ncand * sizeof(Cand), hipMemcpyDeviceToHost));
    // the first frame position of every stream is a chain root even if it does not look like a sync code
    for (uint32_t s = 0; s < n; s++) cands.push_back(Cand{s, 0, in->off[s] + D.info[s].first_byte});
    std::sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.stream != b.stream ? a.stream < b.stream : a.byte < b.byte; });
    cands.erase(std::unique(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.stream == b.stream && a.byte == b.byte; }), cands.end());
    ncand = cands.size();
    // -- 3. parse every candidate
    const uintptr_t dptr = reinterpret_cast<uintptr_t>(in->data());
    FlacGlobals G;
    G.src = in->data();
    G.w0 = reinterpret_cast<const unsigned long long *>(dptr & ~(uintptr_t)7);
    G.base_bit = 8 * (dptr & 7);
    G.off = reinterpret_cast<const unsigned long long *>(in->d_off);
    G.info = reinterpret_cast<const FlacStreamInfo *>(hb.p);
    if ((rc = cb.ensure((ncand + 1) * (sizeof(Cand) + sizeof(FrameInfo)) + 64))) return rc;  // +1: scratch slot for on-demand parses
    AUKIT_HIP_CHECK(hipMemcpyAsync(cb.p, cands.data(), ncand * sizeof(Cand), hipMemcpyHostToDevice, ctx->stream));
    FrameInfo *d_fi = reinterpret_cast<FrameInfo *>(reinterpret_cast<char *>(cb.p) + (ncand + 1) * sizeof(Cand));
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    hipLaunchKernelGGL(k_flac_parse, dim3((unsigned)((ncand + 63) / 64)), dim3(64), 0, ctx->stream, G, reinterpret_cast<const Cand *>(cb.p), ncand, d_fi, 4);
    AUKIT_HIP_CHECK(hipGetLastError());
    if ((rc = ctx_end_kernel(ctx, "k_flac_parse", in->total()))) return rc;
    std::vector<FrameInfo> fi(ncand);
    AUKIT_HIP_CHECK(hipMemcpyAsync(fi.data(), d_fi, ncand * sizeof(FrameInfo), hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // -- 4. follow the chain per stream
    D.frames.assign(n, {});
    D.status.assign(n, 0);
    D.row_off.assign((size_t)n * D.channels, 0);
    D.row_len.assign((size_t)n * D.channels, 0);
    std::vector<SubJob> sub;
    std::vector<FinJob> fin;
    uint64_t tot = 0;
    size_t ci = 0;
    for (uint32_t s = 0; s < n; s++) {
        size_t lo = ci;
        while (ci < ncand && cands[ci].stream == s) ci++;
        const size_t hi = ci;
        // first pass over the chain to size the rows
        std::vector<size_t> chain;
        uint64_t at = in->off[s] + D.info[s].first_byte;
        const uint64_t endb = in->off[s + 1];
        for (;;) {
            if (at >= endb) break;  // readByte() → nil → decodeFrame returns false
            size_t a = lo, b = hi;
            while (a < b) { size_t m = (a + b) / 2; if (cands[m].byte < at) a = m + 1; else b = m; }
            if (a >= hi || cands[a].byte != at) {
                // not in the pre-parsed list (failed the CRC filter, or no sync pattern here): parse this one position now —
                // k_flac_parse reports "Sync code expected" / nil arithmetic exactly like decodeFrame would
                const Cand one{s, 0, at};
                AUKIT_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<Cand *>(cb.p) + ncand, &one, sizeof(Cand), hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(k_flac_parse, dim3(1), dim3(64), 0, ctx->stream, G, reinterpret_cast<const Cand *>(cb.p) + ncand, 1ull, d_fi + ncand, 0);
                FrameInfo extra;
                AUKIT_HIP_CHECK(hipMemcpyAsync(&extra, d_fi + ncand, sizeof(FrameInfo), hipMemcpyDeviceToHost, ctx->stream));
                AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                fi.push_back(extra);
                a = fi.size() - 1;
            }
            if (fi[a].status == FE_LIMIT) {  // a real frame larger than the budget: parse it again without one
                hipLaunchKernelGGL(k_flac_parse, dim3(1), dim3(64), 0, ctx->stream, G, reinterpret_cast<const Cand *>(cb.p) + a, 1ull, d_fi + a, 0);
                AUKIT_HIP_CHECK(hipMemcpyAsync(&fi[a], d_fi + a, sizeof(FrameInfo), hipMemcpyDeviceToHost, ctx->stream));
                AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            }
            const FrameInfo &f = fi[a];
            if (f.status == FE_EOF_START) break;
            if (f.status != FE_OK) { D.status[s] = f.status; break; }
            chain.push_back(a);
            at = (f.end_bit - G.base_bit) / 8;
        }
        uint64_t L = 0;
        for (size_t a : chain) L += (uint64_t)fi[a].blocksize;
        const uint64_t stride = round_up(std::max<uint64_t>(L, 1), 2);
        for (int c = 0; c < D.channels; c++) { D.row_off[(size_t)s * D.channels + c] = tot + (uint64_t)c * stride; D.row_len[(size_t)s * D.channels + c] = L; }
        uint64_t sp = 0;
        for (size_t a : chain) {
            const FrameInfo &f = fi[a];
            D.frames[s].push_back({sp, f.blocksize});
            for (int c = 0; c < D.channels; c++) {
                int dep = D.depth;
                if (f.chan_asgn >= 8) {
                    if (c >= 2) break;  // only subframes[1], [2] exist in the decorrelated modes; extra channels would be nil
                    if (c == 0) dep += (f.chan_asgn == 9 ? 1 : 0); else dep += (f.chan_asgn == 9 ? 0 : 1);
                }
                sub.push_back(SubJob{f.sub_bit[c], tot + (uint64_t)c * stride + sp, s, dep, f.blocksize, (int)f.kind[c]});
            }
            if (f.chan_asgn >= 8) {
                if (D.channels != 2) { D.status[s] = FE_NIL; }
                fin.push_back(FinJob{tot + sp, tot + stride + sp, f.blocksize, f.chan_asgn});
            } else
                for (int c = 0; c < D.channels; c++) fin.push_back(FinJob{tot + (uint64_t)c * stride + sp, 0, f.blocksize, 0});
            sp += (uint64_t)f.blocksize;
        }
        tot += stride * D.channels;
    }
    // -- 5/6. decode the chained frames
    if ((rc = ctx->tmp_buf.ensure((size_t)tot * 8 + 64))) return rc;
    if (!sub.empty()) {
        // lanes of a wave run in lock-step: group subframes that take the same code path (constant / verbatim / order class)
        std::stable_sort(sub.begin(), sub.end(), [](const SubJob &x, const SubJob &y) { return x.pad < y.pad; });
        DevBuf &jb = ctx->seg_buf;
        const size_t sb = sub.size() * sizeof(SubJob), fb = fin.size() * sizeof(FinJob);
        if ((rc = jb.ensure(sb + fb + 64))) return rc;
        AUKIT_HIP_CHECK(hipMemcpyAsync(jb.p, sub.data(), sb, hipMemcpyHostToDevice, ctx->stream));
        AUKIT_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(jb.p) + sb, fin.data(), fb, hipMemcpyHostToDevice, ctx->stream));
        int *err = reinterpret_cast<int *>(reinterpret_cast<char *>(jb.p) + sb + fb);
        AUKIT_HIP_CHECK(hipMemsetAsync(err, 0, 8, ctx->stream));
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        hipLaunchKernelGGL(k_flac_subframe, dim3((unsigned)((sub.size() + 63) / 64)), dim3(64), 0, ctx->stream, G, reinterpret_cast<const SubJob *>(jb.p),
                           (unsigned long long)sub.size(), reinterpret_cast<double *>(ctx->tmp_buf.p), err);
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_flac_subframe", in->total() + tot * 8))) return rc;
        hipLaunchKernelGGL(k_flac_finish, dim3((unsigned)fin.size()), dim3(256), 0, ctx->stream, reinterpret_cast<const FinJob *>(reinterpret_cast<char *>(jb.p) + sb),
                           (unsigned long long)fin.size(), reinterpret_cast<double *>(ctx->tmp_buf.p), D.depth);
        AUKIT_HIP_CHECK(hipGetLastError());
        int herr = 0;
        AUKIT_HIP_CHECK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (herr) return fail(AUKIT_E_HIP, "internal: FLAC subframe decode disagreed with the parse pass (%d)", herr);
    }
    return AUKIT_OK;
}

// aukit.flac(data)  aukit.lua:1657-1660
int decode_flac_audio(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, double new_rate, int interp, bool do_resample, int dtype,
                      aukit_audio **out) {
    FlacDecoded D;
    int rc = flac_decode_rows(ctx, in, D);
    if (rc) return rc;
    for (uint32_t s = 0; s < in->n; s++)
        if (D.status[s]) return fail(AUKIT_E_LUA, "%s", flac_err_msg(D.status[s]));  // decodeFLAC raises: the whole load fails
    return audio_from_int_rows(ctx, SRC_AUDIO_F64, ctx->tmp_buf.p, D.row_off, D.row_len, in->n, D.channels, D.rate, new_rate, interp, do_resample, dtype, 1, 1, out);
}

// ================================================================= stream.flac  aukit.lua:3124-3191
struct FsJob {
    unsigned long long src_off;   // element offset of this (frame, channel) block in the decoded rows
    unsigned long long last_off;  // element offset of the previous (frame, channel) block's LAST sample (src[0]); ~0 → {0, 0}
    unsigned long long out_off;
    int blocksize, nout;
};
template <int INTERP, typename OUT_T>
__global__ __launch_bounds__(64) void k_flac_stream(const FsJob *jobs, unsigned long long njobs, const double *rows, OUT_T *out, double ratio, double rcp, int exact,
                                                   double lp_alpha) {
    const unsigned long long j = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= njobs) return;
    const FsJob job = jobs[j];
    const double *src = rows + job.src_off;  // src[k-1] = table index k
    double m1 = 0, z0 = 0;                   // src[-1], src[0] = last[1], last[2]  :3170-3171
    if (job.last_off != ~0ull) { z0 = rows[job.last_off]; m1 = rows[job.last_off - 1]; }
    double ls = z0 / (z0 < 0 ? 128 : 127);   // :3172
    const int n = job.blocksize;
    auto tap = [&](int k) -> double { return k >= 1 ? src[k - 1] : (k == 0 ? z0 : m1); };
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

int stream_flac(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, int interp, int, int dtype, aukit_audio **out, aukit_chunks **chunks_out) {
    if (interp < 0 || interp > 2) return fail(interp == AUKIT_INTERP_SINC ? AUKIT_E_UNSUPPORTED : AUKIT_E_ARG, "stream.flac: interpolation must be none, linear or cubic");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "stream.flac output must be AUKIT_F64 or AUKIT_F32");
    FlacDecoded D;
    int rc = flac_decode_rows(ctx, in, D);
    if (rc) return rc;
    const int C = D.channels;
    const double ratio = 48000 / D.rate;                                      // :3154
    const double lp_alpha = 1 - std::exp(-(D.rate / 96000) * 2 * M_PI);       // :3155
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0); ck->status.assign(in->n, 0); ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0);
    std::vector<std::vector<uint32_t>> clen(in->n);
    std::vector<std::vector<double>> cpos(in->n);
    for (uint32_t s = 0; s < in->n; s++) {
        ck->length_seconds[s] = D.info[s].nsamples / D.rate;
        // iterator calls: frames accumulate while #chunk[1] < sampleRate; an error or the end of data kills the coroutine (errors are swallowed)
        size_t f = 0;
        bool dead = false;
        double pos = 0;
        while (!dead) {
            uint64_t got = 0;
            while ((double)got < D.rate) {
                if (f >= D.frames[s].size()) { dead = true; break; }
                got += (uint64_t)std::floor((double)D.frames[s][f].second * ratio);
                f++;
            }
            pos = pos + (double)got / 48000;                                  // :3188
            clen[s].push_back((uint32_t)got);
            cpos[s].push_back(pos);
            lens[s] += got;
        }
        ck->nchunks[s] = (uint32_t)clen[s].size();
        ck->max_chunks = std::max(ck->max_chunks, ck->nchunks[s]);
    }
    const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
    ck->lens.assign((size_t)ck->n * mc, 0);
    ck->pos.assign((size_t)ck->n * mc, 0);
    for (uint32_t s = 0; s < in->n; s++)
        for (uint32_t k = 0; k < ck->nchunks[s]; k++) { ck->lens[(size_t)s * mc + k] = clen[s][k]; ck->pos[(size_t)s * mc + k] = cpos[s][k]; }
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, C, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    std::vector<FsJob> jobs;
    uint64_t nouts = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        uint64_t op = 0;
        unsigned long long prev_last = ~0ull;
        for (auto &fr : D.frames[s]) {
            const int nout = (int)std::floor((double)fr.second * ratio);
            for (int c = 0; c < C; c++) {
                FsJob j;
                j.src_off = D.row_off[(size_t)s * C + c] + fr.first;
                j.last_off = prev_last;
                j.out_off = a->row_off[s] + (uint64_t)c * a->row_stride[s] + op;
                j.blocksize = fr.second;
                j.nout = nout;
                jobs.push_back(j);
                prev_last = (fr.second >= 2) ? j.src_off + (uint64_t)fr.second - 1 : prev_last;  // last = {src[#src-1], src[#src]}, shared across channels (Q14)
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
        const double *rows = reinterpret_cast<const double *>(ctx->tmp_buf.p);
        if ((rc = ctx_begin_kernel(ctx))) { delete ck; return rc; }
#define AUKIT_FS(I, T) hipLaunchKernelGGL((k_flac_stream<I, T>), dim3(grid), dim3(64), 0, ctx->stream, dj, (unsigned long long)jobs.size(), rows, reinterpret_cast<T *>(a->dev), ratio, 1.0 / ratio, exact, lp_alpha)
        if (dtype == AUKIT_F64) { if (interp == 0) AUKIT_FS(0, double); else if (interp == 1) AUKIT_FS(1, double); else AUKIT_FS(2, double); }
        else { if (interp == 0) AUKIT_FS(0, float); else if (interp == 1) AUKIT_FS(1, float); else AUKIT_FS(2, float); }
#undef AUKIT_FS
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_flac_stream", in->total() + nouts * dtype_size(dtype)))) { delete ck; return rc; }
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

}  // namespace aukit
