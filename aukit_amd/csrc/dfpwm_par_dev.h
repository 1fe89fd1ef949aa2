// dfpwm_par_dev.h — device-side pieces shared by the chunk-parallel DFPWM decoder (dfpwm_par.hip) and the chunk-speculative
// transcoder / encoder (dfpwm_spec.hip): the feed description, the clamp-add maps of the strength scan, the chunk decoder.
#pragma once
#include "common.h"
#include "dfpwm_dev.h"

namespace aukit {

typedef unsigned long long u64;

struct SatMap { int a, lo, hi; };
AUKIT_DEV int sm_clamp(int v, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi)); return r; }   // lo <= hi everywhere: one v_med3_i32 (the ternaries became compares and selects)
AUKIT_DEV SatMap sm_then(const SatMap &f, const SatMap &g) { return SatMap{f.a + g.a, sm_clamp(f.lo + g.a, g.lo, g.hi), sm_clamp(f.hi + g.a, g.lo, g.hi)}; }
AUKIT_DEV int sm_apply(const SatMap &f, int x) { return sm_clamp(x + f.a, f.lo, f.hi); }

// The byte sequence a decoder is fed is described by (run, stride): fed byte f is source byte (f / run) * stride + f % run of the
// (pseudo-)stream, and `fed` bytes are fed in all:
//   aukit.dfpwm   slices of 6001 bytes advanced by 6000 (Q10: every 6001st byte is fed twice)   run 6001, stride 6000
//   stream.dfpwm  slices of 6000 C + 1 bytes advanced by 6000 C (:2449-2451)                     run 6000 C + 1, stride 6000 C
//   MDFPWM        decoderL / decoderR see alternate 6000-byte blocks (:1432-1436)                run 6000, stride 12000
struct Feed { u64 run, stride; };
AUKIT_DEV u64 dfp_src_index(u64 f, const Feed &fd) { const u64 k = f / fd.run; return k * fd.stride + (f - k * fd.run); }

// Calls fn(byte) for the fed bytes [f0, f1) in order.  Inside a run the source is contiguous: aligned 16-byte loads (the next one
// in flight while the current one is consumed), single bytes up to the first aligned address and around the end of a run.
// fn4(word) takes four fed bytes at once (the dwords of the aligned vectors: little-endian, first byte lowest).
// DEEP: 64 bytes per turn, the next 64 requested before these are looked at — a lone lane (the serial passes) waits a microsecond or two for
// every load it has not asked for early; and a lane that comes back to its 128-byte line for the next 16 bytes every few thousand instructions
// finds it gone from the L2 (131 072 lanes' lines: k_dfx_chunks fetched 3.8 GB for 2.1 of input + warm-up)
template <bool DEEP = false, typename F, typename F4>
AUKIT_DEV void fed_for_each(const unsigned char *p, u64 f0, u64 f1, const Feed &fd, F &&fn, F4 &&fn4) {
    if (f0 >= f1) return;
    const u64 k0 = f0 / fd.run, o0 = f0 - k0 * fd.run;
    const unsigned char *a = p + k0 * fd.stride + o0;  // next source byte
    const unsigned run = (unsigned)fd.run;
    const long long skip = (long long)fd.stride - (long long)fd.run;  // added at the end of a run (-1 for the overlapping slices)
    unsigned left = (unsigned)(fd.run - o0);                          // fed bytes left in the current run
    u64 rem = f1 - f0;
    uint4 pre = make_uint4(0, 0, 0, 0);
    bool have = false;
    if constexpr (DEEP) {
        // (rolled: one copy of fn4's body — the vectors rotate through c0 instead of being indexed)
        uint4 c0 = pre, c1 = pre, c2 = pre, c3 = pre, n0 = pre, n1 = pre, n2 = pre, n3 = pre;
        bool have4 = false;
        while (rem) {
            if (((uintptr_t)a & 15) == 0 && left >= 64 && rem >= 64) {
                const uint4 *v = reinterpret_cast<const uint4 *>(a);
                if (!have4) { c0 = v[0]; c1 = v[1]; c2 = v[2]; c3 = v[3]; }
                have4 = left >= 128 && rem >= 128;
                if (have4) { n0 = v[4]; n1 = v[5]; n2 = v[6]; n3 = v[7]; }
#pragma unroll 1
                for (int q = 0; q < 4; q++) {
                    const unsigned w4[4] = {c0.x, c0.y, c0.z, c0.w};
#pragma unroll 1
                    for (int w = 0; w < 4; w++) fn4(w4[w]);
                    c0 = c1; c1 = c2; c2 = c3;
                }
                if (have4) { c0 = n0; c1 = n1; c2 = n2; c3 = n3; }
                a += 64; left -= 64; rem -= 64;
            } else {
                fn((unsigned)*a);
                a++; left--; rem--;
                have4 = false;
            }
            if (left == 0) { a += skip; left = run; have4 = false; }
        }
        return;
    }
    while (rem) {
        if (((uintptr_t)a & 15) == 0 && left >= 16 && rem >= 16) {
            const uint4 q = have ? pre : *reinterpret_cast<const uint4 *>(a);
            have = left >= 32 && rem >= 32;
            if (have) pre = *reinterpret_cast<const uint4 *>(a + 16);
            const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll 1
            for (int w = 0; w < 4; w++) fn4(w4[w]);
            a += 16; left -= 16; rem -= 16;
        } else {
            fn((unsigned)*a);
            a++; left--; rem--;
        }
        if (left == 0) { a += skip; left = run; have = false; }
    }
}
template <typename F>
AUKIT_DEV void fed_for_each(const unsigned char *p, u64 f0, u64 f1, const Feed &fd, F &&fn) {
    fed_for_each<false>(p, f0, f1, fd, fn, [&](unsigned word) {
#pragma unroll
        for (int j = 0; j < 4; j++) fn((word >> (8 * j)) & 0xFF);
    });
}

struct DfParParams {
    const unsigned char *src;
    const u64 *off;     // [n] first source byte of each (pseudo-)stream
    const u64 *fed;     // [n] bytes fed to its decoder
    Feed feed;
    unsigned n;
    unsigned nblk;      // map blocks per stream (max over the batch) = chunks per stream: block b covers the fed bytes [B(b), B(b + 1)),
                        // B(0) = 0, B(b) = b * bpc * W - W — the positions where the chunk lanes start their warm-up
    unsigned bpc;       // blocks per chunk
    unsigned nchunk;    // chunks per stream
    u64 W;              // block size in fed bytes
    int init_n;         // states at `init`: stream s starts from state s % init_n (stream.mdfpwm: decoderL / decoderR alternate)
    const int *init;    // [init_n][6]: the state a stream's decoder starts from (a bounded reader-function handle's rest of a stream); null: the reset state
    u64 lead;           // rows mode, C == 1: elements to skip at the start of each row (stream.dfpwm's leading 0)
    SatMap *maps;       // [n][nblk][msub]
    int *s_start;       // [n][nblk + 1] strength at block starts
    int *st_start, *st_end;  // [n][nchunk][6] decoder state after warm-up / at chunk end (n, strength, pb, lpf, pn, -: dfpwm_dev.h); strength -1 = no such chunk
    // output
    int mode;           // 0: rows (C channels), 1: stereo → mono mix
    int C;
    signed char *out;
    const u64 *out_off, *out_stride;  // rows: element offset of channel 0 / channel stride per stream; mix: element offset per stream
    unsigned *stats;
    unsigned skip_last;   // k_df_blockmaps: the last block's map is not wanted (nothing starts behind it): identity, its bytes are not read
    const unsigned *lead_on;  // k_df_blockmaps (may be null): [n] the fed unit at which the silence a stream starts with ends (0: none) ...
    unsigned char *inner;     // ... and (may be null) [n] set to 1 where a stream holds digital silence BEHIND that: a whole turn of the scan (64 or 128
                              // source bytes) that is one byte repeated — dfpwm_spec.hip declines large batches of such streams
    unsigned msub;        // k_df_blockmaps / k_df_blockscan: lanes per map block (0 = 1): `maps` is [n][nblk][msub], a block's bytes cut into msub pieces
    unsigned c_lo, c_hi;  // k_df_chunks / k_df_verify: the chunk indices [c_lo, c_hi) of every stream (a time slice of the batch)
};

// first fed byte of chunk c
AUKIT_DEV u64 dfp_chunk_start(const DfParParams &P, unsigned c) { return (u64)c * P.bpc * P.W; }

struct DfOut {  // where decoded samples go
    int mode, C;
    signed char *base;   // rows: channel 0 of the stream; mix: the stream's mono samples
    u64 stride, lead;
    Feed feed;
    const signed char *lut;
};

// Fast-forward through repeated source dwords (FF, the serial redo of k_df_verify): the decoder is a function of (state, bits), so when a dword
// left the state where it found it, the same dword again leaves it there again and makes the same 32 samples — digital silence (0x55 / 0xAA
// bytes at the strength floor, the charge stepping +-1 around wherever it stood when the silence began: the one thing a chunk lane's warm-up
// never finds out, every chunk of the passage is redone) and rails reach such a cycle within a few dwords.  A compare and the stores instead
// of 512 instructions: one second of silence inside 64 streams cost the loader 3.5 of its 4.2 ms.
typedef unsigned dfp_u32x4a __attribute__((ext_vector_type(4), aligned(4)));
struct DfRepeat {
    unsigned pw = 0;
    bool fixed = false;
    int s[5] = {0, 0, 0, 0, 0};
    dfp_u32x4a va, vb;
    AUKIT_DEV bool hit(unsigned word) const { return fixed && word == pw; }
    AUKIT_DEV void before(const DfDec &d) { s[0] = d.p.n; s[1] = d.p.strength; s[2] = d.p.pb; s[3] = d.lpf; s[4] = d.pn; }
    AUKIT_DEV void after(unsigned word, const DfDec &d) {
        fixed = s[0] == d.p.n && s[1] == d.p.strength && s[2] == d.p.pb && s[3] == d.lpf && s[4] == d.pn;
        pw = word;
    }
};

// decode fed bytes [f0, f1) of one stream; EMIT = false: state only
template <bool EMIT, bool FF = false>
AUKIT_DEV void dfp_run(const unsigned char *p, u64 f0, u64 f1, DfDec &d, const DfOut &O) {
    u64 i = 8 * f0;  // index of the next decoded sample in the fed order
    [[maybe_unused]] DfRepeat rp;
    if (EMIT && O.mode == 1) {
        // stereo frames → one mono int8 each: 4 per fed byte (one dword store), 16 per aligned source dword (one 16-byte store — a
        // quarter of the store instructions, each of which visits 64 cache lines for the 64 streams of a wave)
        const signed char *lutc = O.lut + 128 * 257;  // indexed by signed (l, r)
        auto four = [&](unsigned byte) -> unsigned {
            const unsigned nb = ~byte;
            unsigned packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int l = df_decode_b(d, df_pm1(nb, 2 * k)), r = df_decode_b(d, df_pm1(nb, 2 * k + 1));
                packed |= ((unsigned)(unsigned char)lutc[l * 256 + r]) << (8 * k);
            }
            return packed;
        };
        typedef unsigned u32x4a __attribute__((ext_vector_type(4), aligned(4)));
        fed_for_each<FF>(p, f0, f1, O.feed,
                     [&](unsigned byte) { *reinterpret_cast<unsigned *>(O.base + (i >> 1)) = four(byte); i += 8; if (FF) rp.fixed = false; },
                     [&](unsigned word) {
                         u32x4a v;
                         if (FF && rp.hit(word)) v = rp.va;
                         else {
                             if (FF) rp.before(d);
                             v.x = four(word & 0xFF); v.y = four((word >> 8) & 0xFF); v.z = four((word >> 16) & 0xFF); v.w = four(word >> 24);
                             if (FF) { rp.after(word, d); rp.va = v; }
                         }
                         *reinterpret_cast<u32x4a *>(O.base + (i >> 1)) = v;
                         i += 32;
                     });
        return;
    }
    if (EMIT && O.mode == 0 && (O.C == 2 || (O.C == 1 && !O.lead))) {
        // de-interleaved rows of one or two channels, the loaders' usual cases: 16-byte stores for the dwords of the aligned source vectors
        typedef unsigned u32x4a __attribute__((ext_vector_type(4), aligned(4)));
        if (O.C == 2) {
            auto pair = [&](unsigned byte, unsigned &c0, unsigned &c1) {  // four stereo frames of one fed byte
                const unsigned nb = ~byte;
                c0 = 0; c1 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    c0 |= ((unsigned)(unsigned char)df_decode_b(d, df_pm1(nb, 2 * k))) << (8 * k);
                    c1 |= ((unsigned)(unsigned char)df_decode_b(d, df_pm1(nb, 2 * k + 1))) << (8 * k);
                }
            };
            fed_for_each<FF>(p, f0, f1, O.feed,
                         [&](unsigned byte) {
                             unsigned c0, c1;
                             pair(byte, c0, c1);
                             *reinterpret_cast<unsigned *>(O.base + (i >> 1)) = c0;
                             *reinterpret_cast<unsigned *>(O.base + O.stride + (i >> 1)) = c1;
                             i += 8;
                             if (FF) rp.fixed = false;
                         },
                         [&](unsigned word) {
                             u32x4a va, vb;
                             if (FF && rp.hit(word)) { va = rp.va; vb = rp.vb; }
                             else {
                                 if (FF) rp.before(d);
                                 unsigned a[4], b[4];
                                 pair(word & 0xFF, a[0], b[0]); pair((word >> 8) & 0xFF, a[1], b[1]); pair((word >> 16) & 0xFF, a[2], b[2]); pair(word >> 24, a[3], b[3]);
                                 va.x = a[0]; va.y = a[1]; va.z = a[2]; va.w = a[3]; vb.x = b[0]; vb.y = b[1]; vb.z = b[2]; vb.w = b[3];
                                 if (FF) { rp.after(word, d); rp.va = va; rp.vb = vb; }
                             }
                             *reinterpret_cast<u32x4a *>(O.base + (i >> 1)) = va;
                             *reinterpret_cast<u32x4a *>(O.base + O.stride + (i >> 1)) = vb;
                             i += 32;
                         });
        } else {
            auto eight = [&](unsigned byte, unsigned &lo, unsigned &hi) {  // the eight samples of one fed byte
                const unsigned nb = ~byte;
                lo = 0; hi = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) lo |= ((unsigned)(unsigned char)df_decode_b(d, df_pm1(nb, k))) << (8 * k);
#pragma unroll
                for (int k = 0; k < 4; k++) hi |= ((unsigned)(unsigned char)df_decode_b(d, df_pm1(nb, 4 + k))) << (8 * k);
            };
            fed_for_each<FF>(p, f0, f1, O.feed,
                         [&](unsigned byte) {
                             unsigned lo, hi;
                             eight(byte, lo, hi);
                             *reinterpret_cast<uint2 *>(O.base + i) = make_uint2(lo, hi);
                             i += 8;
                             if (FF) rp.fixed = false;
                         },
                         [&](unsigned word) {
                             u32x4a va, vb;
                             if (FF && rp.hit(word)) { va = rp.va; vb = rp.vb; }
                             else {
                                 if (FF) rp.before(d);
                                 unsigned a[8];
                                 eight(word & 0xFF, a[0], a[1]); eight((word >> 8) & 0xFF, a[2], a[3]); eight((word >> 16) & 0xFF, a[4], a[5]); eight(word >> 24, a[6], a[7]);
                                 va.x = a[0]; va.y = a[1]; va.z = a[2]; va.w = a[3]; vb.x = a[4]; vb.y = a[5]; vb.z = a[6]; vb.w = a[7];
                                 if (FF) { rp.after(word, d); rp.va = va; rp.vb = vb; }
                             }
                             *reinterpret_cast<u32x4a *>(O.base + i) = va;
                             *reinterpret_cast<u32x4a *>(O.base + i + 16) = vb;
                             i += 32;
                         });
        }
        return;
    }
    fed_for_each(p, f0, f1, O.feed, [&](unsigned byte) {
        const unsigned nb = ~byte;
        if (!EMIT) {
#pragma unroll
            for (int k = 0; k < 8; k++) df_decode_b(d, df_pm1(nb, k));
        } else if (O.C == 1 && O.lead) {  // row shifted by `lead` elements: byte stores
#pragma unroll
            for (int k = 0; k < 8; k++) O.base[O.lead + i + k] = (signed char)df_decode_b(d, df_pm1(nb, k));
            i += 8;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int v = df_decode_bit(d, byte & 1);
                byte >>= 1;
                const u64 fr = i / (u64)O.C;
                O.base[(i - fr * (u64)O.C) * O.stride + fr] = (signed char)v;
                i++;
            }
        }
    });
}

AUKIT_DEV void dfp_pack(const DfDec &d, int *o) { o[0] = d.p.n; o[1] = d.p.strength; o[2] = d.p.pb; o[3] = d.lpf; o[4] = d.pn; o[5] = 0; }
AUKIT_DEV void dfp_unpack(const int *o, DfDec &d) { d.p.n = o[0]; d.p.strength = o[1]; d.p.pb = o[2]; d.lpf = o[3]; d.pn = o[4]; }

AUKIT_DEV int dfp_mix(int l, int r) {  // aukit.pcm table input :1082, Audio:mono :685-686, encodePCM :874 + the encoder's floor
    double acc = 0;
    acc = acc + (double)l / (l < 0 ? 128 : 127);
    acc = acc + (double)r / (r < 0 ? 128 : 127);
    const double m = acc / 2;
    return (int)floor(m * (m < 0 ? 128 : 127));
}

AUKIT_DEV DfOut dfp_out(const DfParParams &P, unsigned s, const signed char *lut) {
    DfOut O;
    O.mode = P.mode; O.C = P.C; O.lut = lut;
    O.base = P.out + P.out_off[s];
    O.stride = (P.mode == 0 && P.out_stride) ? P.out_stride[s] : 0;
    O.feed = P.feed; O.lead = P.lead;
    return O;
}

}  // namespace aukit
