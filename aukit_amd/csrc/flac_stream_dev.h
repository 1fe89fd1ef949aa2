// flac_stream_dev.h — what the register-resident FLAC decoders share (flac_stream.hip: one wave per 64 frames; flac_pq.hip: a parser wave and a
// predictor wave per 64 frames, small batches): the lane's ring window and its MSB-first reader (BitInputStream, aukit.lua:342-378), the stereo
// combinations (:482-497), the finals' stores.
#pragma once
#include <algorithm>
#include <type_traits>
#include "flac_dev.h"

namespace aukit {
namespace {

constexpr int SWD = 32;             // dwords of bit-stream window per lane (a ring)
constexpr int SWS = SWD + 4;        // row stride in dwords (16-byte aligned rows: a granule enters with one ds_write_b128)
constexpr int SPF = 5;              // 16-byte granules a lane requests per round
constexpr int SMAXO = 12;           // predictor orders served
constexpr unsigned SRING = SWD - 1;
constexpr int SOS = 36;              // dwords between two lanes' output rows (128 bytes + 16: 16-byte aligned, conflict-free 8-byte writes)
constexpr int SNEED = 7;            // dwords of window behind the last one read that a group of four values may touch
constexpr int SLOOK = 8;            // dwords of window a generic step wants in front of it (every header of an ordinary stream)

enum { S_FRAME = 0, S_SUB, S_RUN, S_CONST, S_COEF, S_PART, S_SUBEND, S_FRAMEEND, S_DONE };

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// MSB-first reader of the generic code, on the lane's ring window in LDS only: lw[k & 31] = big-endian dword k of the batch (counted from
// G.w0) for k in [wlo, whi).  A read beyond it — a unary run of hundreds of zero bits — raises `oow` and the frame is declined.
struct SRd {
    const unsigned *lw;
    u64 wlo, whi, pos, end;
    int eof, oow;
};
AUKIT_DEV unsigned srd_dword(SRd &r, u64 k) {
    if (k - r.wlo >= r.whi - r.wlo) r.oow = 1;
    return r.lw[(unsigned)k & SRING];
}
AUKIT_DEV unsigned srd_peek(SRd &r) {   // the next 32 bits
    const u64 d = r.pos >> 5;
    const unsigned u = (unsigned)r.pos & 31u;
    const unsigned a = srd_dword(r, d);
    if (u == 0) return a;
    return (a << u) | (srd_dword(r, d + 1) >> (32 - u));
}
AUKIT_DEV unsigned srd_get(SRd &r, int n) {   // BitInputStream.readUint(n), 0 <= n <= 31  (:351-364)
    if (n == 0) return 0;
    if (r.pos + (u64)n > r.end) { r.eof = 1; return 0; }
    const unsigned v = srd_peek(r) >> (32 - n);
    r.pos += (u64)n;
    return v;
}
AUKIT_DEV int srd_sget(SRd &r, int n) {       // readSignedInt(n)  (:365-369)
    const unsigned v = srd_get(r, n);
    return n > 0 ? ((int)(v << (32 - n)) >> (32 - n)) : 0;
}
// readRiceSignedInt (:370-376) with every check: codes longer than 32 bits, the end of the data.  Returns FE_OK (value in v), FE_NIL, FE_DECLINE.
AUKIT_DEV int srd_rice(SRd &b, int k, int &v) {
    u64 zeros = 0;
    bool gz = true, dec = false;
    while (gz) {   // (single-exit: see k_flac_chain)
        if (b.pos >= b.end) { b.eof = 1; gz = false; }
        else {
            const unsigned hi = srd_peek(b);
            const u64 avail = b.end - b.pos;
            const int z = hi ? __builtin_clz(hi) : 32;
            if ((u64)z >= avail) { b.eof = 1; gz = false; }   // ran off the end inside the unary prefix
            else if (z < 32) { zeros += (u64)z; b.pos += (u64)z + 1; gz = false; }
            else { zeros += 32; b.pos += 32; if (zeros > (1u << 26) || b.oow) { dec = true; gz = false; } }
        }
    }
    const unsigned lowb = srd_get(b, k);
    const u64 u = (zeros << k) + lowb;
    if (b.eof) return FE_NIL;
    if (dec || (u >> 31)) return FE_DECLINE;   // beyond int32: the first design's overflow path knows what to do
    v = (int)(unsigned)(u >> 1) ^ -(int)(unsigned)(u & 1);
    return FE_OK;
}

// the finals' stores.  Non-temporal by default as in k_flac_decode; -DAUKIT_FS_PLAIN: ordinary (cached, write-back) stores
template <typename V>
__device__ __forceinline__ void sstore(V v, V *p) {
#ifdef AUKIT_FS_PLAIN
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ int smad24(int a, int b, int c) {   // v_mad_i32_i24 named outright (cf. flac.hip)
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// the three combinations of :482-497 (first subframe a, second subframe b) in one form: X - ((Y >> sh) & m), that plus Y, with (X, Y) = (a, b) for
// LEFT/SIDE (sh 0) and MID/SIDE (sh 1: floor(side / 2)), (b, a) and m = 0 for SIDE/RIGHT
AUKIT_DEV void sdecor(bool swap, int sh, int m, int a, int b, int &c0, int &c1) {
    const int X = swap ? b : a, Y = swap ? a : b;
    c1 = X - ((Y >> sh) & m);
    c0 = c1 + Y;
}

}  // namespace
}  // namespace aukit
