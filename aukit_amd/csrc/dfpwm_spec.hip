// dfpwm_spec.hip — the chunk-speculative DFPWM transcoder: aukit.dfpwm(d, 2, rate):mono():dfpwm() (aukit.lua:1392-1414, :677-689,
// :1005-1018; BASELINE config 4) with ONE lane per (stream, time chunk) that decodes its chunk's stereo samples, mixes them down and
// ENCODES them on the spot — the mono samples never leave the lane's registers, and a batch of any size fills the chip
// (the lane-per-stream encoder of dfpwm_par.hip has a serial floor of 27 ns per sample whatever the batch: 13 ms for ten seconds).
//
// The decoder side is dfpwm_par.hip's: exact strength at the warm-up start from the clamp-add scan, charge and filter from zero,
// verified afterwards.  The ENCODER's bits depend on its own state, so nothing about it is scannable; what makes it speculable:
//   * its charge is a contraction towards the samples, its strength a ±1 counter driven by its own bits — warmed up over a few
//     thousand samples two encoders end in the same state IF they sit in the same class of the invariant
//         I = (strength - 2 [previous bit = 0] - t) mod 4      (t = index of the next sample)
//     which every step preserves unless the strength is clamped at 8 or 1023 (strength' = strength + b b', and
//     b b' = 1 - 2 (q xor q') ≡ 1 + 2 q + 2 q' mod 4 with q = [bit = 0]).  Measured on the config-4 signal (tools/experiments/,
//     DESIGN.md §3.10): started from all 2032 (strength, previous bit) pairs, 2048 samples leave 4 distinct states — one per class;
//     with the true class a single guess is right in 99.4 % (2048 samples) / 99.8 % (2560) / 99.99 % (4096) of the chunks;
//   * the true encoder only ever clamps in its first samples on such input (strength starts at the floor), so the class it is in after
//     512 samples (k_dfx_prologue, a lane per stream) is its class for good — or until a passage at the floor (silence), which
//     the verify pass notices: it re-speculates the rest of that stream with the class it now knows (a "round").
// Exactness never rests on any of this: k_dfx_verify walks every stream's chunks in order, compares each chunk's recorded start
// state (decoder and encoder) with the true end state of the chunk before it and, where they differ, runs the chunk again from the
// true state — block by block, against the states the chunk lane left at every block boundary, until the two runs have merged.
#include <algorithm>
#include "common.h"
#include "dfpwm_dev.h"
#include "dfpwm_par_dev.h"

namespace aukit {

int dfpwm_strength_scan(aukit_ctx *ctx, const DfParParams &P);  // k_df_blockmaps + k_df_blockscan (dfpwm_par.hip)
}
extern "C" int aukit_dfpwm_transcode_mono(aukit_ctx *ctx, const aukit_batch *in, int channels, aukit_batch **out);
extern "C" int aukit_batch_wrap_device(aukit_ctx *ctx, aukit_batch **out, const void *dev_bytes, const uint64_t *offsets, uint32_t n);
extern "C" void aukit_batch_free(aukit_batch *b);
namespace aukit {

AUKIT_DEV unsigned dfs_class(const DfEnc &e, u64 t) { return ((unsigned)e.strength - (e.pb < 0 ? 2u : 0u) - (unsigned)t) & 3u; }
AUKIT_DEV int dfs_pack(const DfEnc &e) { return (int)((unsigned)e.cu | (unsigned)e.strength << 8 | (e.pb > 0 ? 1u << 18 : 0u)); }
AUKIT_DEV DfEnc dfs_unpack(int v) { DfEnc e; e.cu = v & 255; e.strength = (v >> 8) & 1023; e.pb = (v >> 18) & 1 ? 1 : -1; return e; }

constexpr unsigned DFX_PROBE_FROM = 256, DFX_PROBE_END = 768;  // fed bytes: the probe's guess warms up over 512 of them (2048 mono samples: on the config-4
                                                                 // signal 0.6 % of such guesses miss; the batch is declined at 6 %)
#ifndef AUKIT_DFX_WG
#define AUKIT_DFX_WG 256   // threads of a chunk-lane workgroup (one 64 KiB mix table each: two workgroups per CU)
#endif
#ifndef AUKIT_DFX_STRIKES
#define AUKIT_DFX_STRIKES 3
#endif
constexpr int DFX_STRIKES = AUKIT_DFX_STRIKES;   // rounds in a row in which a stream got less than a sixteenth of itself further: hard
constexpr unsigned DFX_LEAD_MAX = 60000;   // fed units (5 s of a 48 kHz stream) of leading silence the prologue walks through for the second reference
constexpr unsigned DFX_X0 = 128;  // fed bytes (512 mono samples) the prologue runs from the reset state to learn the encoder's class

struct DfxParams {
    DfParParams P;          // src, off, fed, feed, n, nblk = nchunk, bpc, W (= the block: Wd + We), s_start
    unsigned Wd;            // fed bytes at the start of a warm-up block that only the decoder runs (its charge and filter settle)
    unsigned npad;          // n rounded up to 64
    unsigned round, rounds; // this launch's round; rounds in all (the last verify redoes whatever is left)
    int *st;                // [nchunk][12][npad] start state (decoder n, strength, pb, lpf, pn; encoder packed), end state (same six)
    unsigned probe;         // k_dfx_prologue also tries this many guesses per stream, window behind window (flags[14]: guesses that missed, [15]: that met in silence); 0: no probe
    unsigned fix_iv;        // intervals k_dfx_fix runs a chunk again before it gives up (rounds before the last)
    unsigned G, nck;        // checkpoints: the state every G fed bytes inside a chunk (G divides W), nck = bpc W / G - 1 of them
    int *ck;                // [nchunk][nck][6][npad]
    int *fx;                // [nchunk][13][npad] k_dfx_fix's record per chunk: 0 none, else intervals it wrote << 1 | merged with the chunk lane's run (0: ran to the chunk's end),
                            //   the state it started from (6), the end state it reached (6, status 2)
    int *ctl;               // [9][npad]: first chunk not final yet (nchunk: done, nchunk + 1: given up — a "hard" stream); the reference encoder state the
                            //   chunk lanes model their guess on (packed); the true state (5 + 1) where that chunk starts; strikes
    int *pst;               // [8][npad] k_dfx_prologue, `phase` 1 -> 2: the true state where the references end (the decoder's five words, the encoder's one, the fed position)
    unsigned *onset;        // [npad] k_dfx_onset: the fed unit at which the silence a stream starts with ends (0: none)
    int *ref2;              // [npad] the SECOND reference of round 0: the true encoder behind the silence a stream starts with (dfs_pack; 0: none)
    unsigned phase;         // k_dfx_prologue: 0 reference and probe in one launch; 1 the reference only; 2 the probe, from `pst`
    unsigned *hard;         // [npad] the hard streams (flags[13] of them): left to the lane-per-stream encoder (host side)
    unsigned *flags;        // [r] = round r's verify left something to re-speculate;  [8] chunks, [9] checkpoint intervals run again, [10] streams re-speculated, [11] chunks run again by k_dfx_fix, [13] hard streams
    const unsigned char *lut;  // [65536] mono + 128 for (l + 128) * 256 + (r + 128): the reference's fp64 mix (dfp_mix), in global memory
    const u64 *count;       // rows kind (Audio:dfpwm on int8 samples): samples per stream — P.fed counts whole UNITS of four of them, the rest is the tail's
    unsigned char *enc_out; // the packed result
    const u64 *ooff;        // [n + 1]
};

// ---- output bits: 4 per fed byte, 16 per source dword; whole 8-byte rounds leave as one store
struct DfsAcc {
    u64 bits = 0;            // the round being filled
    u64 h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0;   // whole rounds held back, h0 the newest
    unsigned pos = 0, nh = 0;
    unsigned char *o = nullptr;
};
typedef unsigned dfs_u32x2u __attribute__((ext_vector_type(2), aligned(1)));  // (unaligned global stores are single instructions on gfx950)
typedef unsigned dfs_u32x4u __attribute__((ext_vector_type(4), aligned(1)));
AUKIT_DEV void dfs_store16(unsigned char *o, u64 a, u64 b) {
    dfs_u32x4u w; w.x = (unsigned)a; w.y = (unsigned)(a >> 32); w.z = (unsigned)b; w.w = (unsigned)(b >> 32);
    *reinterpret_cast<dfs_u32x4u *>(o) = w;
}
AUKIT_DEV void dfs_store8(unsigned char *o, u64 a) {
    dfs_u32x2u w; w.x = (unsigned)a; w.y = (unsigned)(a >> 32);
    *reinterpret_cast<dfs_u32x2u *>(o) = w;
}
// a full round of 64 bits: every eighth one leaves, with the seven before it, as FOUR 16-byte stores in a row — 64 bytes of one or two lines that
// the L2 sees together.  (8-byte stores from 131 072 lanes at their own addresses left the L2 as partial sectors, 6.5 GB written for 1 GB of
// result; 16-byte stores 3.6 GB: the lane's next one came thousands of instructions later, its line long evicted — profiles/r05_dfpwm_pmc_write.csv)
AUKIT_DEV void dfs_round(DfsAcc &a, u64 r) {
#ifdef AUKIT_DFS_PAIRS   // (A/B: 16-byte stores, every second round)
    if (a.nh == 1) { dfs_store16(a.o, a.h0, r); a.o += 16; a.nh = 0; return; }
#endif
    if (a.nh == 7) {
        dfs_store16(a.o, a.h6, a.h5); dfs_store16(a.o + 16, a.h4, a.h3); dfs_store16(a.o + 32, a.h2, a.h1); dfs_store16(a.o + 48, a.h0, r);
        a.o += 64;
        a.nh = 0;
        return;
    }
    a.h6 = a.h5; a.h5 = a.h4; a.h4 = a.h3; a.h3 = a.h2; a.h2 = a.h1; a.h1 = a.h0; a.h0 = r;
    a.nh++;
}
// whatever whole rounds are held back go out, oldest first (chunk ends, checkpoints a run may stop at)
AUKIT_DEV void dfs_flush(DfsAcc &a) {
    const u64 h[7] = {a.h0, a.h1, a.h2, a.h3, a.h4, a.h5, a.h6};
#pragma unroll
    for (int k = 6; k >= 0; k--)
        if ((unsigned)k < a.nh) { dfs_store8(a.o, h[k]); a.o += 8; }
    a.nh = 0;
}
AUKIT_DEV void dfs_put(DfsAcc &a, unsigned v, unsigned nbits) {
    a.bits |= (u64)v << a.pos;
    a.pos += nbits;
    if (a.pos >= 64) {
        dfs_round(a, a.bits);
        a.pos -= 64;
        a.bits = a.pos ? (u64)(v >> (nbits - a.pos)) : 0;
    }
}

// The chunk lanes' tables in LDS: the mix (indexed by signed (l, r)) and, per source byte, its eight bits as ± 1 in eight nibbles — a decoder step
// wants its bit as + 1 / - 1, two instructions from the byte (bit-field extract, or 1), ONE from this dword (a signed 4-bit field): a look-up and
// eight extracts per byte instead of sixteen instructions, 1.75 of the 47 per mono sample.
struct DfxLds {
    const unsigned char *mix;
    const unsigned *bits;
    const int *lp;   // 128 - 140 q for q = -128 .. 128, addressed from its middle: the low-pass step's first multiply-add as a look-up (-DAUKIT_DFX_LP_TABLE, an
                     // A/B that lost: 12.3 -> 12.8 ms at 16 384 streams, profiles/r05_dfx_lp_table_ab.txt — the LDS pipe has no room for eight more per byte)
    AUKIT_DEV unsigned char operator[](int i) const { return mix[i]; }
};
// df_decode_b (dfpwm_dev.h) with the low-pass's `128 - 140 q` from a table: q = (n + x + 3) >> 2 is only ever multiplied by -140 and biased, so
// the lane keeps 4 q — the same sum with its two low bits cleared, one AND for the shift — as the table's byte offset, and one multiply-add is left
AUKIT_DEV int dfx_decode_bt(DfDec &d, int b, const int *lp) {
    const bool same = b == d.p.pb;
    const int pn = d.pn;
    const int n = df_predict(d.p, b);
    const int q4 = (n + (same ? n : pn) + 3) & ~3;
    d.pn = n;
    const int t = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(lp) + q4);
    int u;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(u) : "v"(d.lpf), "s"(116), "v"(t));
    d.lpf = u >> 8;
    return d.lpf;
}
AUKIT_DEV void dfx_bits_to_lds(unsigned *bt, unsigned nthreads) {
    for (unsigned i = threadIdx.x; i < 256; i += nthreads) {
        unsigned t = 0;
        for (int k = 0; k < 8; k++) t |= ((i >> k) & 1u ? 0x1u : 0xFu) << (4 * k);
        bt[i] = t;
    }
}
// one fed byte: eight decoder steps, four mixes, four encoder steps → four bits
template <typename LUT>
AUKIT_DEV unsigned dfx_byte(DfDec &d, DfEnc &e, unsigned byte, LUT lutc) {
    unsigned out = 0;
    if constexpr (std::is_same<LUT, DfxLds>::value) {
        const unsigned tb = lutc.bits[byte];
#pragma unroll
        for (int k = 0; k < 4; k++) {
#ifdef AUKIT_DFX_LP_TABLE
            const int l = dfx_decode_bt(d, (int)(tb << (28 - 8 * k)) >> 28, lutc.lp), r = dfx_decode_bt(d, (int)(tb << (24 - 8 * k)) >> 28, lutc.lp);
#else
            const int l = df_decode_b(d, (int)(tb << (28 - 8 * k)) >> 28), r = df_decode_b(d, (int)(tb << (24 - 8 * k)) >> 28);
#endif
            out |= df_encode_u(e, (unsigned)lutc[l * 256 + r]) & (1u << k);
        }
    } else {
        const unsigned nb = ~byte;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int l = df_decode_b(d, df_pm1(nb, 2 * k)), r = df_decode_b(d, df_pm1(nb, 2 * k + 1));
            out |= df_encode_u(e, (unsigned)lutc[l * 256 + r]) & (1u << k);
        }
    }
    return out;
}

// fed bytes [f0, f1) of one stream through decoder, mix and encoder; EMIT: the bits go to `acc`
template <bool EMIT, typename LUT>
AUKIT_DEV void dfx_span(const unsigned char *p, u64 f0, u64 f1, const Feed &fd, DfDec &d, DfEnc &e, LUT lutc, DfsAcc &acc) {
    // (64-byte turns of the source — fed_for_each<true> — would fetch 2.5 GB instead of 3.8 at 16 384 streams and cost 0.4 ms of 12.9: the kernel is
    // bound by its instructions, not by its bytes; profiles/r05_dfx_burst_ab.txt)
#ifndef AUKIT_DFX_LOADS_DEEP
#define AUKIT_DFX_LOADS_DEEP false
#endif
    fed_for_each<AUKIT_DFX_LOADS_DEEP>(p, f0, f1, fd,
                 [&](unsigned byte) {
                     const unsigned b4 = dfx_byte(d, e, byte, lutc);
                     if (EMIT) dfs_put(acc, b4, 4);
                 },
                 [&](unsigned word) {
                     unsigned b16 = 0;
#pragma unroll
                     for (int j = 0; j < 4; j++) b16 |= dfx_byte(d, e, (word >> (8 * j)) & 0xFF, lutc) << (4 * j);
                     if (EMIT) dfs_put(acc, b16, 16);
                 });
}

// ---- the same engine on int8 SAMPLES (Audio:dfpwm, aukit.lua:1005-1018, behind k_dfpwm_quantize): no decoder, no table — `p` is the stream's row
// of samples and the index unit is four samples (what a fed byte of the stereo transcode makes: every index computation stays as it is).
// Kernels and helpers take the row kind as their "table" argument.
struct DfeRows {};
template <bool EMIT>
AUKIT_DEV void dfx_span(const unsigned char *p, u64 f0, u64 f1, const Feed &, DfDec &, DfEnc &e, DfeRows, DfsAcc &acc) {
    auto unit = [&](unsigned w) -> unsigned {   // four samples, first lowest
        w ^= 0x80808080u;                         // u = v + 128
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) out |= df_encode_u(e, (w >> (8 * k)) & 0xFF) & (1u << k);
        return out;
    };
    u64 f = f0;
    const unsigned char *a = p + 4 * f0;   // (rows are 16-byte aligned, chunk and checkpoint starts multiples of 16 units)
    while (f < f1 && ((uintptr_t)a & 15)) { const unsigned b4 = unit(*reinterpret_cast<const unsigned *>(a)); if (EMIT) dfs_put(acc, b4, 4); a += 4; f++; }
    if (f + 4 <= f1) {
        uint4 cur = *reinterpret_cast<const uint4 *>(a);
        while (f + 4 <= f1) {
            uint4 nxt = cur;
            if (f + 8 <= f1) nxt = *reinterpret_cast<const uint4 *>(a + 16);   // in flight while `cur` is encoded
            const unsigned b16 = unit(cur.x) | unit(cur.y) << 4 | unit(cur.z) << 8 | unit(cur.w) << 12;
            if (EMIT) dfs_put(acc, b16, 16);
            cur = nxt; a += 16; f += 4;
        }
    }
    while (f < f1) { const unsigned b4 = unit(*reinterpret_cast<const unsigned *>(a)); if (EMIT) dfs_put(acc, b4, 4); a += 4; f++; }
}

// the stream's last bits: the last byte is padded with samples of value 0 (aukit.lua:1011-1016 hands the encoder whole bytes)
AUKIT_DEV void dfx_tail(DfsAcc &acc, DfEnc &e, const unsigned char *rest = nullptr, unsigned nrest = 0) {
    dfs_flush(acc);
    // (rows kind: the `nrest` < 4 samples behind the last whole unit first)
    for (unsigned k = 0; k < nrest; k++) { acc.bits |= (u64)(df_encode_u(e, (unsigned)rest[k] ^ 0x80u) & 1u) << acc.pos; acc.pos++; }
    while (acc.pos & 7) { acc.bits |= (u64)(df_encode_u(e, 128u) & 1u) << acc.pos; acc.pos++; }
    for (unsigned k = 0; k < acc.pos; k += 8) *acc.o++ = (unsigned char)(acc.bits >> k);
    acc.pos = 0; acc.bits = 0;
}
// the tail of stream s where the stream ends at unit `fed` (transcode: nothing behind the last fed byte; rows: count % 4 samples)
template <int KIND>
AUKIT_DEV void dfx_stream_tail(const DfxParams &X, unsigned s, const unsigned char *p, u64 fed, DfsAcc &acc, DfEnc &e) {
    if constexpr (KIND == 0) dfx_tail(acc, e);
    else dfx_tail(acc, e, p + 4 * fed, (unsigned)(X.count[s] - 4 * fed));
}

AUKIT_DEV void dfx_store6(int *st, unsigned npad, const DfDec &d, const DfEnc &e) {
    int v[6];
    dfp_pack(d, v);
#pragma unroll
    for (int i = 0; i < 5; i++) st[(size_t)i * npad] = v[i];
    st[(size_t)5 * npad] = dfs_pack(e);
}
AUKIT_DEV void dfx_load6(const int *st, unsigned npad, int *v) {
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = st[(size_t)i * npad];
}
AUKIT_DEV bool dfx_same6(const int *a, const int *b) {
    bool same = true;
#pragma unroll
    for (int i = 0; i < 6; i++) same = same && a[i] == b[i];
    return same;
}

// The encoder a chunk lane starts its warm-up with, modelled on a true state of the stream (the reference: after the stream's first samples,
// or where the verify pass last found the guesses failing): the reference's strength and previous bit — the reference and every warm-up
// start sit at sample indices that are multiples of 4, so the copy is in the reference's class of the invariant without further ado — and
// the charge on the first sample `u0`.  A reference AT the strength floor (silence: the encoder runs a 2-cycle there, clamped at every
// step, and which of its two phases it is in is not a matter of class) is copied whole where the first sample looks like the same silence.
// `flat`: the lane's first source dword is one byte four times (digital silence: 0x55 / 0xAA in DFPWM, one sample value in the rows) — a
// signal that merely crosses the reference's level where the lane starts is no silence (one chunk in forty did, and took its stream to a
// second round)
AUKIT_DEV DfEnc dfx_guess(const DfEnc &ref, unsigned u0, int ref2 = 0, bool flat = true) {
    DfEnc e;
    const int du = ref.cu - (int)u0;
    if (ref.strength > 9) { e.strength = ref.strength; e.pb = ref.pb; e.cu = (int)u0; }
    else if (du >= -2 && du <= 2 && flat) e = ref;
    else if (ref2) {
        // the stream STARTS in silence (the reference sits at the floor) and this lane does not: the prologue walked on to where the signal
        // sets in and left the true state there — its class is the one the signal keeps until the next clamp
        const DfEnc r2 = dfs_unpack(ref2);
        e.strength = r2.strength; e.pb = r2.pb; e.cu = (int)u0;
    } else {
        // a floor reference says nothing about a passage with signal: any strength away from the floor, in the reference's class all the
        // same — the lanes of the stream then agree with each other, which is what tells a change of class from noise
        e.cu = (int)u0; e.pb = ref.pb; e.strength = 40 + ((ref.strength - 40) & 3);
    }
    return e;
}

// the 64 KiB mix table, once per context (k_dfx_chunks copies it into LDS: 256 bytes per thread instead of 256 fp64 mixes)
__global__ __launch_bounds__(256) void k_dfx_lut(unsigned char *lut) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    lut[i] = (unsigned char)(dfp_mix((int)(i >> 8) - 128, (int)(i & 255) - 128) + 128);
}

// A round after the first runs only if the round before left streams to speculate again — and not at all once most of the batch has been
// given up on: the input is noise-like then, the lucky rest would fail a few chunks further on, and every round costs the time of a whole
// chunk lane however few streams it is for.  (Streams left unfinished this way are hard streams to the host.)
AUKIT_DEV bool dfx_round_off(const DfxParams &X) {
    if (__hip_atomic_load(&X.flags[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;   // the probe declined the batch
    if (X.round == 0) return false;
    // ... nor once more streams are hard than the exact parallel encoder takes (16): the lane-per-stream schedule will run then, and it takes
    // its 12 ms for one stream or for all that are left — every further round is time on top (64 noise-like streams that got past the probe:
    // six rounds and a fallback for the half of them given up on, 28 ms; 15 with this)
    const unsigned hard = __hip_atomic_load(&X.flags[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return !__hip_atomic_load(&X.flags[X.round - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || 2 * hard > X.P.n || hard > 16;
}

// the probe's verdict, on the device (the host hears of it at the end of the call: nothing in between waits for the host)
// (`inner`, large batches: streams with digital silence behind the one they start with — every such passage ends the speculation of its
// stream for this round, and a batch cut into few chunks has no second one: more than eight of them and the lane-per-stream schedule, which
// then has to run for them, may as well run for all)
__global__ __launch_bounds__(256) void k_dfx_decide(unsigned *flags, unsigned n, unsigned silence_counts, const unsigned char *inner, unsigned windows) {
    __shared__ unsigned cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    if (inner && silence_counts) {
        unsigned c = 0;
        for (unsigned i = threadIdx.x; i < n; i += 256) c += inner[i];
        if (c) atomicAdd(&cnt, c);
    }
    __syncthreads();
    if (threadIdx.x) return;
    if (inner) flags[12] = cnt;
    if ((unsigned long long)(flags[14] + (silence_counts ? flags[15] : 0u)) * 16 > (unsigned long long)n * windows || cnt > 8) flags[6] = 1;   // (one guess in sixteen)
}

// copies the mix table into LDS (64 KiB: 4096 16-byte vectors)
AUKIT_DEV void dfx_lut_to_lds(const unsigned char *g, unsigned char *l, unsigned nthreads) {
    const uint4 *gv = reinterpret_cast<const uint4 *>(g);
    uint4 *lv = reinterpret_cast<uint4 *>(l);
    for (unsigned i = threadIdx.x; i < 4096; i += nthreads) lv[i] = gv[i];
}

// Where the silence a stream STARTS with ends: a wave per stream compares the source bytes behind unit DFX_X0 with the byte there, 1 KiB a turn
// (X.onset[s] = the first fed unit that holds another byte; 0: the stream does not go on with its byte at DFX_X0 for at least 64 units — no
// silence worth the name — or all of the DFX_LEAD_MAX units looked at are silent).  Digital silence is ONE byte repeated: 0x55 or 0xAA in
// DFPWM, one sample value in the rows.
template <int KIND>
__global__ __launch_bounds__(64) void k_dfx_onset(const DfxParams X) {
    const DfParParams &P = X.P;
    const unsigned s = blockIdx.x, lane = threadIdx.x;
    const u64 fed = P.fed[s];
    unsigned res = 0;
    if (fed > 3 * (u64)DFX_X0) {
        const unsigned char *p = P.src + P.off[s];
        // source bytes [i0, i1): fed units [DFX_X0, lim) — rows: four bytes a unit; transcode: the feed's runs overlap by one byte (6001 / 6000),
        // a source byte index i is fed unit i + i / stride
        const u64 lim = fed - DFX_X0 < (u64)DFX_X0 + DFX_LEAD_MAX ? fed - DFX_X0 : (u64)DFX_X0 + DFX_LEAD_MAX;
        const u64 i0 = KIND == 1 ? 4ull * DFX_X0 : dfp_src_index(DFX_X0, P.feed), i1 = KIND == 1 ? 4 * lim : dfp_src_index(lim, P.feed);
        const unsigned v = p[i0];
        u64 first = i1;
        for (u64 i = i0; i < i1 && first == i1; i += 1024) {
            const u64 at = i + 16ull * lane;
            unsigned bad = 16;
            for (unsigned k = 0; k < 16; k++) if (at + k < i1 && p[at + k] != v && bad == 16) bad = k;
            const u64 m = __ballot(bad < 16);
            if (m) first = i + 16ull * (unsigned)__builtin_ctzll(m) + (unsigned)__shfl((int)bad, __builtin_ctzll(m));
        }
        if (first < i1) {
            const u64 f = KIND == 1 ? first / 4 : (first / P.feed.stride) * P.feed.run + first % P.feed.stride;
            if (f >= (u64)DFX_X0 + 64) res = (unsigned)f;
        }
    }
    if (lane == 0) X.onset[s] = res;
}

// a lane per stream: the true encoder's state after the first DFX_X0 fed bytes (the reference of round 0); the control block
template <int KIND>
__global__ __launch_bounds__(256) void k_dfx_prologue(const DfxParams X) {
    extern __shared__ unsigned char lutu[];
    const DfParParams &P = X.P;
    // (the mix table in LDS here too: a lone lane per stream waits for every look-up, and one in global memory is an L2 round trip —
    // 0.91 ms of prologue and probe at 16 384 streams, 0.41 with this)
    if constexpr (KIND == 0) {   // (four waves to a workgroup: they share the copy)
        dfx_lut_to_lds(X.lut, lutu, 256);
        __syncthreads();
    }
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= P.n) return;
    const unsigned char *p = P.src + P.off[s];
    const u64 fed = P.fed[s], f1 = fed < DFX_X0 ? fed : (u64)DFX_X0;
    DfDec d{};
    DfEnc e{};
    DfsAcc acc;
    auto lutc = [&]() { if constexpr (KIND == 0) return (const unsigned char *)lutu + 128 * 257; else return DfeRows{}; }();
    u64 pos = f1;        // where the references end and the probe sets out
    int ref2 = 0;
    if (X.phase != 2) {
        dfx_span<false>(p, 0, f1, P.feed, d, e, lutc, acc);
        X.ctl[s] = 0;
        X.ctl[(size_t)X.npad + s] = dfs_pack(e);   // (at sample 4 f1: a multiple of 4, like every warm-up start — see k_dfx_chunks)
        int v[6];
        dfp_pack(d, v);
#pragma unroll
        for (int i = 0; i < 5; i++) X.ctl[(size_t)(2 + i) * X.npad + s] = v[i];   // the decoder's side of the reference
        X.ctl[(size_t)8 * X.npad + s] = 0;
        // A stream that STARTS in digital silence (every other CD rip): the reference sits at the strength floor and says nothing about the
        // class the encoder will be in once the signal sets in — every lane behind the onset would guess one class in four, all of them the
        // same, and the whole batch would pay a second round.  The lane walks on through the silence (k_dfx_onset found where it ends),
        // DFX_X0 units into the signal, and leaves the state there as the lanes' second model.
        const u64 on = X.onset[s];
        if (e.strength <= 9 && on) {
            // (1) a few dwords until the pair of states repeats: decoder and encoder are functions of (state, bits), so a dword that left them
            // where it found them does so again — every further dword of the silence can be skipped; (2) on from the last whole dword before the
            // onset, through it, until the strength leaves the floor; (3) DFX_X0 units into the signal: the second reference
            u64 f = f1;
            bool fixed = false;
            for (int k = 0; k < 16 && !fixed && f + 4 < on; k++) {
                const int s0[8] = {d.p.n, d.p.strength, d.p.pb, d.lpf, d.pn, e.cu, e.strength, e.pb};
                dfx_span<false>(p, f, f + 4, P.feed, d, e, lutc, acc);
                f += 4;
                fixed = s0[0] == d.p.n && s0[1] == d.p.strength && s0[2] == d.p.pb && s0[3] == d.lpf && s0[4] == d.pn && s0[5] == e.cu && s0[6] == e.strength && s0[7] == e.pb;
            }
            if (fixed || f + 4 >= on) {
                if (fixed && on > f + 8) f += (on - 8 - f) / 4 * 4;   // (8 short: a feed run's doubled byte maps to the later of its two units)
                u64 g = f;
                while (g < on + 256 && g + DFX_X0 < fed && e.strength <= 9) { dfx_span<false>(p, g, g + 4, P.feed, d, e, lutc, acc); g += 4; }
                if (e.strength > 9 && g + DFX_X0 <= fed) {
                    dfx_span<false>(p, g, g + DFX_X0, P.feed, d, e, lutc, acc);
                    g += DFX_X0;
                    if (e.strength > 9) ref2 = dfs_pack(e);
                }
                pos = g;
            } else pos = f;
        }
        X.ref2[s] = ref2;
        if (X.phase == 1) {   // (the control block is the verify pass's from here on: the probe continues from a copy)
            dfp_pack(d, v);
#pragma unroll
            for (int i = 0; i < 5; i++) X.pst[(size_t)i * X.npad + s] = v[i];
            X.pst[(size_t)5 * X.npad + s] = dfs_pack(e);
            X.pst[(size_t)6 * X.npad + s] = (int)pos;
            X.pst[(size_t)7 * X.npad + s] = X.ctl[(size_t)X.npad + s];   // the reference as THIS pass left it: round 0's verify may rewrite ctl while the probe runs beside it
            return;
        }
    } else {
        int v[6];
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] = X.pst[(size_t)i * X.npad + s];
        dfp_unpack(v, d);
        e = dfs_unpack(v[5]);
        pos = (u64)X.pst[(size_t)6 * X.npad + s];
        ref2 = X.ref2[s];
    }
    // The probe (large batches: a failed speculation costs them a whole step): the true encoder runs on for DFX_PROBE_END - DFX_X0 units, and
    // from DFX_PROBE_FROM - DFX_X0 on a guess modelled on the references runs beside it, the way the chunk lanes' guesses will — do they meet?
    // A small batch gets several such windows per stream, one behind the other (X.probe = their number; round 6): ONE guess per stream said "go"
    // to sixteen streams of noise one time in twenty (a noise stream's guess meets by chance five times in six) and the batch then paid six
    // rounds and a fallback — 15.6 ms against the older schedule's 6.8 (profiles/r06_dfx_grid_before.txt)
    const DfEnc ref = dfs_unpack(X.phase == 2 ? X.pst[(size_t)7 * X.npad + s] : X.ctl[(size_t)X.npad + s]);
    unsigned missed = 0, silent = 0;
    for (unsigned w = 0; w < X.probe; w++) {
        const u64 pf = pos + (DFX_PROBE_FROM - DFX_X0), pe = pos + (DFX_PROBE_END - DFX_X0);
        if (fed < pe) break;
        dfx_span<false>(p, pos, pf, P.feed, d, e, lutc, acc);
        DfEnc g;
        bool first = true, flat;
        if constexpr (KIND == 0) {
            const unsigned b0 = p[dfp_src_index(pf, P.feed)], b1 = p[dfp_src_index(pf + 1, P.feed)], b2 = p[dfp_src_index(pf + 2, P.feed)], b3 = p[dfp_src_index(pf + 3, P.feed)];
            flat = b0 == b1 && b1 == b2 && b2 == b3;
        } else {
            const unsigned w0 = *reinterpret_cast<const unsigned *>(p + 4 * pf);
            flat = w0 == (w0 & 0xFFu) * 0x01010101u;
        }
        if constexpr (KIND == 0) {
            fed_for_each(p, pf, pe, P.feed, [&](unsigned byte) {
                const unsigned nb = ~byte;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int l = df_decode_b(d, df_pm1(nb, 2 * k)), r = df_decode_b(d, df_pm1(nb, 2 * k + 1));
                    const unsigned u = (unsigned)lutc[l * 256 + r];
                    if (first) { g = dfx_guess(ref, u, ref2, flat); first = false; }
                    df_encode_u(e, u);
                    df_encode_u(g, u);
                }
            });
        } else {
            for (u64 i = 4ull * pf; i < 4ull * pe; i++) {
                const unsigned u = (unsigned)p[i] ^ 0x80u;
                if (first) { g = dfx_guess(ref, u, ref2, flat); first = false; }
                df_encode_u(e, u);
                df_encode_u(g, u);
            }
        }
        if (dfs_pack(g) != dfs_pack(e)) missed++;
        else if (e.strength <= 9) silent++;   // (met, but in silence: what follows the silence will be in another class)
        pos = pe;
    }
    if (missed) atomicAdd(&X.flags[14], missed);
    if (silent) atomicAdd(&X.flags[15], silent);
}

AUKIT_DEV int *dfx_ck(const DfxParams &X, unsigned c, unsigned j, unsigned s) { return X.ck + (((size_t)c * X.nck + j) * 6) * X.npad + s; }

// Runs chunk c of stream s again from (d, e), checkpoint interval by interval, writing its bytes, until its state equals the one the chunk
// lane left at a checkpoint — from there on the chunk lane's bytes and end state are the true ones (returns true) — or to the chunk's end
// (returns false; (d, e) = the end state, the stream's tail written when the stream ends here).
// `min_iv` / `may_merge`: k_dfx_fix may have run this chunk before from a state that was not the true one — its bytes lie over the
// chunk lane's in the first `min_iv` intervals (no merge counts before those are rewritten), or in all of them (`may_merge` false).
template <int KIND, typename LUT>
AUKIT_DEV bool dfx_rerun(const DfxParams &X, unsigned s, unsigned c, const unsigned char *p, u64 fed, DfDec &d, DfEnc &e, LUT lutc, unsigned &intervals,
                         unsigned min_iv = 0, bool may_merge = true, unsigned max_iv = 0xFFFFFFFFu, bool *gave_up = nullptr) {
    const DfParParams &P = X.P;
    const u64 f0 = dfp_chunk_start(P, c), e1 = dfp_chunk_start(P, c + 1), f1 = e1 < fed ? e1 : fed;
    DfsAcc acc;
    acc.o = X.enc_out + X.ooff[s] + f0 / 2;
    u64 b0 = f0;
    for (unsigned j = 0; j < X.nck; j++) {
        const u64 b1 = b0 + X.G;
        if (b1 >= f1) break;
        dfx_span<true>(p, b0, b1, P.feed, d, e, lutc, acc);
        intervals++;
        b0 = b1;
        int t[6], w[6];
        dfp_pack(d, t);
        t[5] = dfs_pack(e);
        dfx_load6(dfx_ck(X, c, j, s), X.npad, w);
        if (may_merge && j + 1 >= min_iv && dfx_same6(t, w)) { dfs_flush(acc); return true; }
        if (j + 1 >= max_iv) { dfs_flush(acc); *gave_up = true; return false; }  // (k_dfx_fix before the last round: two runs that have not merged by now are not going to)
    }
    dfx_span<true>(p, b0, f1, P.feed, d, e, lutc, acc);
    intervals++;
    if (f1 == fed) dfx_stream_tail<KIND>(X, s, p, fed, acc, e);
    dfs_flush(acc);
    return false;
}

// a lane per (stream, chunk)
template <int KIND>
__global__ __launch_bounds__(AUKIT_DFX_WG) void k_dfx_chunks(const DfxParams X) {
    extern __shared__ unsigned char lutu[];
    [[maybe_unused]] __shared__ unsigned bits_lds[KIND == 0 ? 256 : 1];
#ifdef AUKIT_DFX_LP_TABLE
    [[maybe_unused]] __shared__ int lp_lds[KIND == 0 ? 260 : 1];
#else
    int *const lp_lds = nullptr;
#endif
    const DfParParams &P = X.P;
    if (dfx_round_off(X)) return;
    if constexpr (KIND == 0) {
        dfx_lut_to_lds(X.lut, lutu, AUKIT_DFX_WG);
        dfx_bits_to_lds(bits_lds, AUKIT_DFX_WG);
#ifdef AUKIT_DFX_LP_TABLE
        for (int i = threadIdx.x; i < 257; i += AUKIT_DFX_WG) lp_lds[i] = 128 - 140 * (i - 128);
#endif
        __syncthreads();
    }
#ifdef AUKIT_DFX_NO_BITS_TABLE   // (A/B)
    auto lutc = [&]() { if constexpr (KIND == 0) return (const unsigned char *)(lutu + 128 * 257); /* indexed by signed (l, r) */ else return DfeRows{}; }();
#else
    auto lutc = [&]() { if constexpr (KIND == 0) return DfxLds{(const unsigned char *)(lutu + 128 * 257), bits_lds, lp_lds + 128}; else return DfeRows{}; }();
#endif
    const u64 gid = (u64)blockIdx.x * AUKIT_DFX_WG + threadIdx.x;
    const unsigned c = (unsigned)(gid / P.n), s = (unsigned)(gid - (u64)c * P.n);  // a wave = one chunk index of 64 streams
    if (c >= P.nchunk) return;
    const unsigned c_from = X.round ? (unsigned)X.ctl[s] : 0u;
    if (c < c_from) return;  // final already (a stream that is done has c_from = nchunk)
    const unsigned char *p = P.src + P.off[s];
    const u64 fed = P.fed[s];
    const u64 f0 = dfp_chunk_start(P, c), e1 = dfp_chunk_start(P, c + 1), f1 = e1 < fed ? e1 : fed;
    int *st = X.st + (size_t)c * 12 * X.npad + s;
    if (f0 >= fed && c > 0) { st[(size_t)X.npad] = -1; return; }  // strength -1: no such chunk
    DfDec d{};
    DfEnc e{};
    DfsAcc acc;
    if (c == c_from) {
        if (c > 0) {  // the true state, left by the verify pass of the round before
            int v[6];
            dfx_load6(X.ctl + (size_t)2 * X.npad + s, X.npad, v);
            dfp_unpack(v, d);
            e = dfs_unpack(v[5]);
        }
    } else {
        if constexpr (KIND == 1) {
            // warm-up over the block before the chunk: a guess modelled on the reference (dfx_guess), charge on the block's first sample
            const u64 fw = f0 - P.W;
            const unsigned w0 = *reinterpret_cast<const unsigned *>(p + 4 * fw);
            e = dfx_guess(dfs_unpack(X.ctl[(size_t)X.npad + s]), (w0 & 0xFFu) ^ 0x80u, X.round ? 0 : X.ref2[s], w0 == (w0 & 0xFFu) * 0x01010101u);
            dfx_span<false>(p, fw, f0, P.feed, d, e, lutc, acc);
        } else {
            // warm-up over the block before the chunk: the decoder with its exact strength and previous bit, charge and filter from zero;
            // after Wd bytes the encoder joins in
            const u64 fw = f0 - P.W, fe = fw + X.Wd;
            d.p.strength = P.s_start[(size_t)s * (P.nblk + 1) + c];
            d.p.pb = fw ? (int)((p[dfp_src_index(fw - 1, P.feed)] >> 6) & 2) - 1 : -1;
            {   // At the strength floor the decoder's charge step is ±1 whatever the charge: an integrator of the bits, which forgets nothing
                // — a warm-up from zero never arrives (digital silence: the bits alternate, the charge is off by its phase for good).  If the
                // reference decoder sat at the floor too, its charge, filter and previous charge are the better start: the same 2-cycle.
                const int rs = X.ctl[(size_t)3 * X.npad + s];
                if (d.p.strength <= 9 && rs >= 0 && rs <= 9) {
                    d.p.n = X.ctl[(size_t)2 * X.npad + s];
                    d.lpf = X.ctl[(size_t)5 * X.npad + s];
                    d.pn = X.ctl[(size_t)6 * X.npad + s];
                }
            }
            DfOut O{};
            O.feed = P.feed;
            dfp_run<false>(p, fw, fe, d, O);
            {   // the first byte of the encoder's warm-up by hand: its first mono sample is where the charge starts
                const unsigned nb = ~(unsigned)p[dfp_src_index(fe, P.feed)];
                unsigned u[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int l = df_decode_b(d, df_pm1(nb, 2 * k)), r = df_decode_b(d, df_pm1(nb, 2 * k + 1));
                    u[k] = (unsigned)lutc[l * 256 + r];
                }
                // The guess is modelled on a true state of this stream (the reference: after the first samples, or where the verify pass
                // last found the guesses failing): its strength and previous bit — the reference and every warm-up start sit at sample
                // indices that are multiples of 4, so the copy is in the reference's class of the invariant without further ado — and the
                // charge on the first sample.  A reference AT the strength floor (silence: the encoder runs a 2-cycle there, clamped at
                // every step, and which of its two phases it is in is not a matter of class) is copied whole, charge included.
                const unsigned b0 = p[dfp_src_index(fe, P.feed)], b1 = p[dfp_src_index(fe + 1, P.feed)], b2 = p[dfp_src_index(fe + 2, P.feed)], b3 = p[dfp_src_index(fe + 3, P.feed)];
                e = dfx_guess(dfs_unpack(X.ctl[(size_t)X.npad + s]), u[0], X.round ? 0 : X.ref2[s], b0 == b1 && b1 == b2 && b2 == b3);
#pragma unroll
                for (int k = 0; k < 4; k++) df_encode_u(e, u[k]);
            }
            dfx_span<false>(p, fe + 1, f0, P.feed, d, e, lutc, acc);
        }
    }
    dfx_store6(st, X.npad, d, e);
    acc.o = X.enc_out + X.ooff[s] + f0 / 2;  // four mono samples per fed byte, eight per output byte
    u64 b0 = f0;
    for (unsigned j = 0; j < X.nck; j++) {
        const u64 b1 = b0 + X.G;
        if (b1 >= f1) break;
        dfx_span<true>(p, b0, b1, P.feed, d, e, lutc, acc);
        dfx_store6(dfx_ck(X, c, j, s), X.npad, d, e);
        b0 = b1;
        if (__hip_atomic_load(&X.flags[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the probe, running beside this pass, declined the batch
    }
    dfx_span<true>(p, b0, f1, P.feed, d, e, lutc, acc);
    if (f1 == fed) dfx_stream_tail<KIND>(X, s, p, fed, acc, e);
    dfs_flush(acc);
    dfx_store6(st + (size_t)6 * X.npad, X.npad, d, e);
}

// a lane per (stream, chunk) again: a chunk whose recorded start state is not the recorded end state of the chunk before it runs again
// from that end state until the two runs merge — every mismatch of the batch at once, a few checkpoint intervals each, instead of one
// after the other in the stream's verify lane.  (The end state of the chunk before is the true one unless that chunk failed itself and
// did not merge: k_dfx_verify checks what this lane started from.)
template <int KIND>
__global__ __launch_bounds__(256) void k_dfx_fix(const DfxParams X) {
    extern __shared__ unsigned char lutu[];
    const DfParParams &P = X.P;
    if (dfx_round_off(X)) return;
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const unsigned c = (unsigned)(gid / P.n), s = (unsigned)(gid - (u64)c * P.n);
    bool need = false;
    int from[6] = {0, 0, 0, 0, 0, 0};
    if (c >= 1 && c < P.nchunk && c > (X.round ? (unsigned)X.ctl[s] : 0u)) {
        int v[6];
        dfx_load6(X.st + (size_t)c * 12 * X.npad + s, X.npad, v);
        if (v[1] >= 0) {
            dfx_load6(X.st + ((size_t)(c - 1) * 12 + 6) * X.npad + s, X.npad, from);
            need = !dfx_same6(v, from);
        }
        if (!need) X.fx[(size_t)c * 13 * X.npad + s] = 0;
    }
    if (!__syncthreads_or(need ? 1 : 0)) return;
    if constexpr (KIND == 0) {
        dfx_lut_to_lds(X.lut, lutu, 256);
        __syncthreads();
    }
    if (!need) return;
    auto lutc = [&]() { if constexpr (KIND == 0) return (const unsigned char *)(lutu + 128 * 257); else return DfeRows{}; }();
    DfDec d;
    dfp_unpack(from, d);
    DfEnc e = dfs_unpack(from[5]);
    // A run that has not merged after a few intervals is given up: the true encoder is not where the chunk lane guessed it for good
    // (another class: it never merges), and the verify lane has the rest of the stream speculated again.
    unsigned iv = 0;
    bool gave_up = false;
    const bool merged = dfx_rerun<KIND>(X, s, c, P.src + P.off[s], P.fed[s], d, e, lutc, iv, 0, true, X.fix_iv, &gave_up);
    int *fx = X.fx + (size_t)c * 13 * X.npad + s;
    fx[0] = gave_up ? -(int)iv : (int)(iv << 1 | (merged ? 1u : 0u));  // (iv >= 1)
#pragma unroll
    for (int i = 0; i < 6; i++) fx[(size_t)(1 + i) * X.npad] = from[i];
    if (!merged) dfx_store6(fx + (size_t)7 * X.npad, X.npad, d, e);
    atomicAdd(&X.flags[9], iv);
    atomicAdd(&X.flags[11], 1u);
}

// a lane per stream: the chain of true states.  A chunk whose recorded start state is the true end state of the chunk before it is final;
// so is one that k_dfx_fix ran again FROM that true state until it merged (or to its end).  Anything else ends the walk: the stream's
// later chunks are speculated again in the next round, modelled on the true state here — unless the rounds are used up or the stream keeps
// failing every few chunks (noise: its encoder touches the strength floor and changes class all the time), in which case it is handed
// to the host as "hard".
__global__ __launch_bounds__(64) void k_dfx_verify(const DfxParams X) {
    const DfParParams &P = X.P;
    if (dfx_round_off(X)) return;
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= P.n) return;
    const unsigned c_from = X.round ? (unsigned)X.ctl[s] : 0u;
    if (c_from >= P.nchunk) return;
    unsigned c = c_from;
    int truth[6];
    dfx_load6(X.st + ((size_t)c * 12 + 6) * X.npad + s, X.npad, truth);  // chunk c ran from the true state
    unsigned chunks = 1;
    for (c++; c < P.nchunk; c++) {
        // (the probe's verdict may arrive while this lane walks its stream: a batch of a few hundred streams is through its chunk pass in 0.1 ms, the
        // probe runs 0.25 ms beside it — a declined batch then paid this walk in full, 0.5 ms of the 1.5 ms a declined call cost: profiles/r06_dfx_grid.txt)
        if ((c & 7u) == 0u && __hip_atomic_load(&X.flags[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const int *st = X.st + (size_t)c * 12 * X.npad + s;
        int v[6];
        dfx_load6(st, X.npad, v);
        if (v[1] < 0) break;  // the stream ended before this chunk
        const int *fx = X.fx + (size_t)c * 13 * X.npad + s;
        const int fst = fx[0];   // 0: k_dfx_fix did not run; < 0: it gave up after -fst intervals; else intervals << 1 | merged
        // (a chunk that started from the true state but that k_dfx_fix ran all the same — the chunk before it was run again to its end without
        // merging, so its RECORDED end state, where k_dfx_fix started this one from, was not the true one — has the bytes of that run over
        // its first intervals: not final)
        if (dfx_same6(v, truth) && fst == 0) { chunks++; dfx_load6(st + (size_t)6 * X.npad, X.npad, truth); continue; }
        if (fst > 0 && !dfx_same6(v, truth)) {
            int w[6];
            dfx_load6(fx + (size_t)X.npad, X.npad, w);
            if (dfx_same6(w, truth)) { chunks++; dfx_load6(fst & 1 ? st + (size_t)6 * X.npad : fx + (size_t)7 * X.npad, X.npad, truth); continue; }
        }
        // the guesses do not lead to the true state from here on
        const unsigned progress = c - c_from, few = P.nchunk / 16 > 2 ? P.nchunk / 16 : 2;
        const int strikes = progress < few ? X.ctl[(size_t)8 * X.npad + s] + 1 : 0;
        atomicAdd(&X.flags[8], chunks);
        // how the later chunks look: after a change of class the guesses still agree with each other (they converge to the same wrong run),
        // on noise they do not — every boundary is a mismatch, and another round would fail at the next one
        unsigned later = 0, bad = 0, flips = 0;
        bool was = true;   // (chunk c itself missed)
        for (unsigned k = c + 1; k < P.nchunk; k++) {
            if (X.st[((size_t)k * 12 + 1) * X.npad + s] < 0) break;
            later++;
            const bool b = X.fx[(size_t)k * 13 * X.npad + s] != 0;
            bad += b ? 1u : 0u;
            flips += b != was ? 1u : 0u;
            was = b;
        }
        // Round 6: noise is told from a passage of silence by how the missed boundaries LIE, not by how many there are.  Noise misses one boundary in
        // six, anywhere: hits and misses alternate all along the stream (flips ≈ 0.28 per chunk).  Silence inside a stream leaves one boundary in five
        // to two re-run as well, but in ONE run (two flips a passage).  (The count alone — one in ten — sent 4096 gated streams to the fallback,
        // 8.8 -> 20 ms; one in two, round 5's threshold, noise never reached: sixteen streams of it went through six rounds and THEN to the fallback,
        // 15.6 ms for the older schedule's 6.8: profiles/r06_dfx_grid_before.txt.)
        if (X.round + 1 >= X.rounds || strikes >= DFX_STRIKES || (later >= 12 && bad * 2 > later) || (later >= 12 && flips >= 6 && flips * 8 > later)) {   // (six flips: more than two passages of silence make — a batch cut into few chunks per stream has few boundaries to count)
            X.ctl[s] = (int)P.nchunk + 1;
            X.hard[atomicAdd(&X.flags[13], 1u)] = s;
            return;
        }
        X.ctl[s] = (int)c;
        X.ctl[(size_t)X.npad + s] = truth[5];   // (at sample 4 x chunk start: a multiple of 4)
#pragma unroll
        for (int i = 0; i < 6; i++) X.ctl[(size_t)(2 + i) * X.npad + s] = truth[i];
        X.ctl[(size_t)8 * X.npad + s] = strikes;
        __hip_atomic_store(&X.flags[X.round], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&X.flags[10], 1u);
        return;
    }
    X.ctl[s] = (int)P.nchunk;  // done
    atomicAdd(&X.flags[8], chunks);
}

// bytes [src_off[i], + len[i]) of `src` to dst + dst_off[i]: the hard streams to and from the lane-per-stream schedule
__global__ __launch_bounds__(256) void k_dfx_move(const unsigned char *src, unsigned char *dst, const u64 *tab, unsigned m) {
    const unsigned i = blockIdx.y;
    const u64 so = tab[i], d0 = tab[m + i], len = tab[2 * (size_t)m + i];
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < len; k += (u64)gridDim.x * 256) dst[d0 + k] = src[so + k];
}

// what the two front ends hand the planner
struct DfxJob {
    int kind;                          // 0: aukit.dfpwm(d, 2, rate):mono():dfpwm() on DFPWM bytes; 1: Audio:dfpwm on int8 samples
    const unsigned char *src;          // the batch's bytes / the rows' buffer
    std::vector<uint64_t> off, fed;    // [n] first byte of every stream; fed bytes / whole units of four samples
    std::vector<uint64_t> count;       // kind 1: samples per stream
    const aukit_batch *in = nullptr;   // kind 0: the batch (hard streams are gathered from it)
    const u64 *d_in_off = nullptr, *d_count = nullptr;   // kind 1: the device tables the lane-per-stream encoder takes
};
int dfpwm_encode_i8(aukit_ctx *ctx, const signed char *in, const u64 *d_in_off, const u64 *d_count, uint32_t n, unsigned char *out, const u64 *d_ooff);  // dfpwm_par.hip
bool dfpwm_encode_i8_small(aukit_ctx *ctx, const signed char *in, const uint64_t *h_in_off, const uint64_t *h_count, uint32_t n, unsigned char *out, const uint64_t *h_ooff, int *rc);  // dfpwm_par.hip: k_dfe_*, up to 16 streams

// host side; *taken = false (nothing launched, or the probe declined: nothing written) when the batch is left to the older schedules
static int dfx_run(aukit_ctx *ctx, const DfxJob &J, unsigned char *out, const u64 *d_ooff, const uint64_t *h_ooff, bool *taken) {
    *taken = false;
    const uint32_t n = (uint32_t)J.off.size();
    if (ctx->dfx_disable || ctx->dfx_off) return AUKIT_OK;
    // (the environment switches of the older schedules — tests and A/B runs — name those schedules: not this one)
    if (n == 0 || getenv("AUKIT_DFPWM_SERIAL") || getenv("AUKIT_DFPWM_NOSPEC") || getenv("AUKIT_DFPWM_FUSED") || getenv("AUKIT_DFPWM_SLICES") || getenv("AUKIT_DFPWM_BLOCK") ||
        getenv("AUKIT_DFPWM_CHUNKS") || getenv("AUKIT_DFPWM_ENC_SERIAL"))
        return AUKIT_OK;
    const std::vector<uint64_t> &h_off = J.off, &h_fed = J.fed;
    uint64_t fed_max = 0, fed_sum = 0;
    for (uint32_t s = 0; s < n; s++) { fed_max = std::max(fed_max, h_fed[s]); fed_sum += h_fed[s]; }
    // A FEW SHORT streams (the reference's own use: one song) stay with the exact parallel encoder k_dfe_* (dfpwm_par.hip): it takes the same
    // 0.7 ms (Audio:dfpwm) / 1.3 ms (transcode) for ten seconds of anything, where a round of this schedule costs a lone lane's time — 0.4 / 1.3 ms
    // for clean signal, but one more round for silence in front, four for two passages of it, six and a fallback for noise
    // (profiles/r05_small_batches.txt: one stream 0.35 - 1.3 / 2.2 - 10 ms; four 0.85 - 2.0 / 1.2 - 12.9).  From about forty stream-seconds on
    // the candidate search's 500-fold work costs more than the rounds do.
    uint64_t few = 500000;   // fed bytes of the batch (stereo transcode; units of four samples for Audio:dfpwm): four ten-second streams
    if (const char *e = getenv("AUKIT_DFX_FEW")) few = strtoull(e, nullptr, 10);
    if (n <= 16 && fed_sum <= few) return AUKIT_OK;
    unsigned We = 640, Wd = 64;  // fed bytes: 2560 mono samples of encoder warm-up behind 64 bytes of decoder-only warm-up
    if (const char *e = getenv("AUKIT_DFX_WE")) We = (unsigned)std::max(64, atoi(e)) & ~63u;
    if (const char *e = getenv("AUKIT_DFX_WD")) Wd = (unsigned)std::max(64, atoi(e)) & ~63u;
    // chunks per stream: enough lanes for `wps` waves on every SIMD, at least three blocks each (the warm-up block is then 1/4 of a lane's work)
    unsigned wps = 2;
    if (const char *e = getenv("AUKIT_DFX_WPS")) wps = (unsigned)std::max(1, atoi(e));
    const uint64_t lanes = (uint64_t)ctx->num_cus * 4 * 64 * wps;
    unsigned want = (unsigned)std::max<uint64_t>(1, (lanes + n - 1) / n);
    if (const char *e = getenv("AUKIT_DFX_CHUNKS")) want = (unsigned)std::max(1, atoi(e));
    unsigned min_bpc = 0;   // 0: chosen below
    if (const char *e = getenv("AUKIT_DFX_MIN_BPC")) min_bpc = (unsigned)std::max(1, atoi(e));
    // long chunks (a large batch) can afford a longer encoder warm-up: 3840 mono samples, one guess in ~5000 misses instead of one in 500 —
    // and every miss costs the step the time of a checkpoint interval in k_dfx_fix
    if (!getenv("AUKIT_DFX_WE") && (fed_max + 703) / 704 >= 16 * (uint64_t)want) We = 960;
    const uint64_t W = (uint64_t)We + Wd;  // a multiple of 64: checkpoint intervals of W / 4 are whole 8-byte output rounds
    const unsigned nblk_all = (unsigned)((fed_max + W - 1) / W);
    // blocks per chunk: a lane walks bpc + 1 blocks, one of them warm-up.  A batch that fills the chip wants the warm-up small (three blocks: a
    // quarter of a lane's work); a small one is better off with short chunks on SIMDs that would idle — the step takes the time of one lane, and
    // so does every further round.  Measured (profiles/r05_small_batches.txt): one block per chunk while the lanes stay under 3/4 of a wave per
    // SIMD, two likewise, else three.
    const unsigned b0 = std::max<unsigned>(nblk_all ? (nblk_all + want - 1) / want : 1, 1);
    unsigned bpc = std::max(b0, min_bpc ? min_bpc : 3u);
    if (!min_bpc)
        for (unsigned b = b0; b < 3; b++)
            if ((uint64_t)n * ((nblk_all + b - 1) / b) * 4 <= (uint64_t)ctx->num_cus * 4 * 64 * 3) { bpc = b; break; }
    bool probe_aside = true;
    if (const char *e = getenv("AUKIT_DFX_PROBE_ASIDE")) probe_aside = atoi(e) != 0;
    const unsigned nchunk = nblk_all ? (nblk_all + bpc - 1) / bpc : 0;
    if (nchunk < 2) return AUKIT_OK;
    // rounds: a re-speculation costs the time of one chunk lane however few streams need it — a fraction of the step when the batch is cut
    // into many chunks per stream, all of it again when it is cut into few (a large batch: none there).  Every passage of silence can cost
    // two (into it, out of it); streams that need more, or fail every few chunks, are "hard" and go to the lane-per-stream encoder.
    unsigned rounds = nchunk >= 24 ? 6 : (nchunk >= 16 ? 2 : 1);
    if (const char *e = getenv("AUKIT_DFX_ROUNDS")) rounds = (unsigned)std::max(1, std::min(atoi(e), 6));   // (flags[round] for rounds 0 .. 5; flags[6] is the probe's "declined": ADVICE r05)
    static_assert(true, "flags: [0, 6) per round, 6 declined, 8 .. 15 counters");
    const unsigned npad = (unsigned)round_up(n, 64);
    // checkpoints: the finer, the less a mismatching chunk runs again before it merges; 24 bytes each, at most ~320 MB of them
    unsigned G = (unsigned)W / 4;
    while (G < W && (uint64_t)nchunk * (bpc * (W / G) - 1) * 24 * npad > (320ull << 20)) G *= 2;
    if (const char *e = getenv("AUKIT_DFX_G")) { const unsigned k = (unsigned)std::max(1, atoi(e)); if (W % k == 0 && (W / k) % 16 == 0) G = (unsigned)W / k; }   // (whole 8-byte output rounds)
    const unsigned nck = bpc * (unsigned)(W / G) - 1;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    // the strength scan's lanes: a map block (the bytes between two chunks' warm-up starts) cut into pieces until four waves per SIMD walk them
    unsigned msub = 1;
    while (msub < 16 && (uint64_t)n * nchunk * msub < (uint64_t)ctx->num_cus * 4 * 64 * 4 && (uint64_t)bpc * W / (2 * msub) >= 1024) msub *= 2;
    if (const char *e = getenv("AUKIT_DFX_MSUB")) msub = (unsigned)std::max(1, std::min(atoi(e), 64));
    const size_t o_tab = take((size_t)n * 24), o_maps = take((size_t)n * nchunk * msub * sizeof(SatMap)), o_ss = take((size_t)n * (nchunk + 1) * 4),
                 o_st = take((size_t)nchunk * 12 * npad * 4), o_ck = take((size_t)nchunk * nck * 6 * npad * 4 + 4), o_fx = take((size_t)nchunk * 13 * npad * 4),
                 o_ctl = take((size_t)9 * npad * 4), o_pst = take((size_t)8 * npad * 4), o_on = take((size_t)2 * npad * 4), o_hard = take((size_t)npad * 4), o_fl = take(64 + (size_t)npad);
    int rc = ctx->tmp_buf2.ensure(o + 256);
    if (rc) return rc;
    char *B = reinterpret_cast<char *>(ctx->tmp_buf2.p);
    if ((rc = h2d_table(ctx, B + o_tab, h_off.data(), (size_t)n * 8)) || (rc = h2d_table(ctx, B + o_tab + (size_t)n * 8, h_fed.data(), (size_t)n * 8))) return rc;
    if (J.kind == 1 && (rc = h2d_table(ctx, B + o_tab + (size_t)n * 16, J.count.data(), (size_t)n * 8))) return rc;
    if (J.kind == 0 && !ctx->dfx_lut.p) {
        if ((rc = ctx->dfx_lut.ensure(65536))) return rc;
        hipLaunchKernelGGL(k_dfx_lut, dim3(256), dim3(256), 0, ctx->stream, reinterpret_cast<unsigned char *>(ctx->dfx_lut.p));
        AUKIT_HIP_CHECK(hipGetLastError());
    }
    DfxParams X{};
    DfParParams &P = X.P;
    P.src = J.src; P.off = reinterpret_cast<const u64 *>(B + o_tab); P.fed = P.off + n; P.feed = Feed{6001, 6000};
    X.count = P.off + 2 * (size_t)n;
    P.n = n; P.nblk = nchunk; P.bpc = bpc; P.nchunk = nchunk; P.W = W;
    P.maps = reinterpret_cast<SatMap *>(B + o_maps); P.s_start = reinterpret_cast<int *>(B + o_ss);
    P.init = nullptr; P.mode = 1; P.C = 2;
    P.msub = msub;
    P.skip_last = 1;   // (the strength behind the last chunk's start is nobody's warm-up: an eighth of the scan's bytes for a batch cut into eight chunks)
    X.Wd = Wd; X.npad = npad; X.rounds = rounds; X.G = G; X.nck = nck;
    X.fix_iv = std::max<unsigned>(2, (unsigned)(W / G));  // a warm-up length
    X.st = reinterpret_cast<int *>(B + o_st); X.ck = reinterpret_cast<int *>(B + o_ck); X.fx = reinterpret_cast<int *>(B + o_fx); X.ctl = reinterpret_cast<int *>(B + o_ctl); X.pst = reinterpret_cast<int *>(B + o_pst); X.onset = reinterpret_cast<unsigned *>(B + o_on); X.ref2 = reinterpret_cast<int *>(B + o_on) + npad; X.hard = reinterpret_cast<unsigned *>(B + o_hard);
    X.flags = reinterpret_cast<unsigned *>(B + o_fl);
    X.lut = reinterpret_cast<const unsigned char *>(ctx->dfx_lut.p);   // (kind 1: never read)
    X.enc_out = out; X.ooff = d_ooff;
    if (hipMemsetAsync(X.flags, 0, 64 + (size_t)npad, ctx->stream) != hipSuccess) return fail(AUKIT_E_HIP, "hipMemsetAsync failed");
    P.lead_on = X.onset; P.inner = reinterpret_cast<unsigned char *>(X.flags) + 64;
    // The probe: one guess per stream is tried first, at the stream's start (0.3 ms of a lane per stream, one look at the outcome).  Where
    // more than one in sixteen misses, or the streams start in silence (whatever follows it will be in another class), the batch is declined
    // and runs the older schedule: a failed speculation costs rounds of a whole chunk lane's time each (and a large batch, cut into few chunks
    // per stream, a whole step for a gain of a quarter).  What the probe cannot see — silence or noise later in the streams — costs rounds.
    // windows of the probe per stream: one.  (More of them — 64 guesses per batch whatever its size — were tried for small batches, round 6: eight
    // windows are 1.5 ms of a lone lane in front of the host's first look, 8 streams of signal 1.0 -> 2.6 ms; what tells noise from signal at
    // any batch size is round 0 itself: k_dfx_verify below counts the chunk boundaries where the guesses missed.  AUKIT_DFX_PROBE_WINDOWS for the A/B)
    X.probe = getenv("AUKIT_DFX_NOPROBE") ? 0u : 1u;
    if (const char *e = getenv("AUKIT_DFX_PROBE_WINDOWS")) X.probe = (unsigned)std::max(1, std::min(atoi(e), 16));
    // (the prologue's 256 lone waves run on the side stream, beside the strength scan: the probe hides behind k_df_blockmaps)
    if (!ctx->dfx_attr_set) {
        AUKIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dfx_chunks<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        AUKIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dfx_fix<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        AUKIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dfx_prologue<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        ctx->dfx_attr_set = true;
    }
    // (where each stream's leading silence ends: the prologue's second reference and the strength scan's look-out for silence behind it)
    if (J.kind == 0) hipLaunchKernelGGL(k_dfx_onset<0>, dim3(n), dim3(64), 0, ctx->stream, X);
    else hipLaunchKernelGGL(k_dfx_onset<1>, dim3(n), dim3(64), 0, ctx->stream, X);
    hipStream_t side = nullptr;
    if ((rc = ctx_side_fork(ctx, &side))) return rc;
    // The chunk lanes do not wait for the probe: they need the reference only (512 samples of a lone lane, not 3072) — the probe runs beside
    // them, and when it declines the batch they hear of it at their next checkpoint (k_dfx_chunks looks at the flag there): a declined batch
    // has lost the 0.3 - 0.4 ms the probe ran on, every other batch starts that much earlier (2048 streams: 0.43 -> 0.19 ms into the step).
    const bool aside = probe_aside && X.probe;
    auto prologue = [&](unsigned phase) {
        X.phase = phase;
        if (J.kind == 0) hipLaunchKernelGGL(k_dfx_prologue<0>, dim3((n + 255) / 256), dim3(256), 65536, side, X);
        else hipLaunchKernelGGL(k_dfx_prologue<1>, dim3((n + 255) / 256), dim3(256), 0, side, X);
    };
    prologue(aside ? 1u : 0u);
    // (streams that START in silence cost one round of re-speculation where the signal sets in — worth it where rounds are cheap, i.e. the batch
    // is cut into many chunks per stream; a batch with few chunks per stream declines them)
    if (J.kind == 0 && (rc = dfpwm_strength_scan(ctx, P))) { (void)ctx_side_join(ctx); return rc; }   // (never leave the side stream forked)
    if ((rc = ctx_side_join(ctx))) return rc;
    // (the verdict on silence behind the leading one needs the scan: here; the probe's verdict follows the probe, on the side stream when it runs aside)
    if (X.probe) hipLaunchKernelGGL(k_dfx_decide, dim3(1), dim3(256), 0, ctx->stream, X.flags, n, nchunk < 24 ? 1u : 0u, J.kind == 0 ? P.inner : nullptr, X.probe);
    if (aside) {   // (the side stream again, behind the reference; joined before the host's first look at the flags)
        prologue(2u);
        hipLaunchKernelGGL(k_dfx_decide, dim3(1), dim3(256), 0, side, X.flags, n, nchunk < 24 ? 1u : 0u, (const unsigned char *)nullptr, X.probe);
    }
    const dim3 grid((unsigned)(((size_t)n * nchunk + 255) / 256)), cgrid((unsigned)(((size_t)n * nchunk + AUKIT_DFX_WG - 1) / AUKIT_DFX_WG));
    // Rounds are queued two at a time with a look at the counters behind each pair (the first look is the call's one host synchronisation on signal:
    // the later rounds of a finely cut batch exist for streams with passages of silence, and queued blind they cost a batch of plain signal a dozen
    // empty launches, 0.12 ms of a 2.8 ms step)
    unsigned h[16] = {};
    for (unsigned r0 = 0; r0 < rounds; r0 += 2) {
        for (unsigned r = r0; r < std::min(rounds, r0 + 2); r++) {
            X.round = r;
            if (J.kind == 0) {
                hipLaunchKernelGGL(k_dfx_chunks<0>, cgrid, dim3(AUKIT_DFX_WG), 65536, ctx->stream, X);
                hipLaunchKernelGGL(k_dfx_fix<0>, grid, dim3(256), 65536, ctx->stream, X);
            } else {
                hipLaunchKernelGGL(k_dfx_chunks<1>, cgrid, dim3(AUKIT_DFX_WG), 0, ctx->stream, X);
                hipLaunchKernelGGL(k_dfx_fix<1>, grid, dim3(256), 0, ctx->stream, X);
            }
            hipLaunchKernelGGL(k_dfx_verify, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, X);
            if (getenv("AUKIT_DFX_TRACE")) {  // (debugging: where the first streams stand after every round)
                int hc[16] = {};
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipMemcpy(hc, X.ctl, sizeof(int) * std::min<unsigned>(8, npad), hipMemcpyDeviceToHost);
                (void)hipMemcpy(hc + 8, X.ctl + npad, sizeof(int) * std::min<unsigned>(8, npad), hipMemcpyDeviceToHost);
                unsigned fl[16] = {};
                (void)hipMemcpy(fl, X.flags, 64, hipMemcpyDeviceToHost);
                fprintf(stderr, "[dfpwm spec] round %u: %u hard streams so far, %u re-speculations so far, %u chunks run again by k_dfx_fix\n", r, fl[13], fl[10], fl[11]);
                fprintf(stderr, "[dfpwm spec] round %u: first not-final chunk of streams 0..7: %d %d %d %d %d %d %d %d; reference states %d %d %d %d %d %d %d %d\n", r, hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7],
                        hc[8], hc[9], hc[10], hc[11], hc[12], hc[13], hc[14], hc[15]);
            }
        }
        AUKIT_HIP_CHECK(hipGetLastError());
        if (aside && r0 == 0 && (rc = ctx_side_join(ctx))) return rc;
        AUKIT_HIP_CHECK(hipMemcpyAsync(h, X.flags, 64, hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const unsigned last = std::min(rounds, r0 + 2) - 1;
        if (h[6] || !h[last]) break;   // declined, or the pair's last verify left nothing to speculate again
    }
    ctx->counters[AUKIT_COUNTER_DFPWM_CHUNKS] = h[8]; ctx->counters[AUKIT_COUNTER_DFPWM_CHUNKS_REDONE] = h[11]; ctx->counters[AUKIT_COUNTER_DFPWM_RESPECULATED] = h[10];
    ctx->counters[AUKIT_COUNTER_DFPWM_HARD] = h[13];
    if (getenv("AUKIT_DFPWM_STATS"))
        fprintf(stderr, "[dfpwm spec] probe: of %u streams, %u guesses missed, %u silent where the probe ends, %u with silence behind the one they start with%s\n", n, h[14], h[15], h[12], h[6] ? ": declined" : "");
    if (getenv("AUKIT_DFPWM_STATS") && !h[6])
        fprintf(stderr, "[dfpwm spec] %u streams x %u chunks of %u blocks of %llu fed bytes (decoder-only warm-up %u, checkpoints every %u, %u rounds): %u chunks verified; %u chunks run again by k_dfx_fix (%u checkpoint intervals); %u stream rounds re-speculated (flags %u %u %u %u %u %u); %u hard streams\n",
                n, nchunk, bpc, (unsigned long long)W, Wd, G, rounds, h[8], h[11], h[9], h[10], h[0], h[1], h[2], h[3], h[4], h[5], h[13]);
    if (h[6]) return AUKIT_OK;   // declined by the probe (nothing was written): not taken, the caller runs the older schedule
    std::vector<unsigned> hs;
    if (h[13] || h[10]) {   // something was given up on or speculated again: who is not done?
        std::vector<int> cf(n);
        AUKIT_HIP_CHECK(hipMemcpy(cf.data(), X.ctl, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (uint32_t s = 0; s < n; s++) if (cf[s] != (int)nchunk) hs.push_back(s);
        ctx->counters[AUKIT_COUNTER_DFPWM_HARD] = hs.size();
    }
    const unsigned m = (unsigned)hs.size();
    if (m && J.kind == 1) {
        // Hard streams of Audio:dfpwm: the lane-per-stream encoder takes any offsets — their table entries, gathered
        std::vector<uint64_t> tab(3 * (size_t)m);
        for (unsigned i = 0; i < m; i++) { tab[i] = J.off[hs[i]]; tab[m + i] = J.count[hs[i]]; tab[2 * (size_t)m + i] = h_ooff[hs[i]]; }
        if ((rc = ctx->dfx_gather.ensure(tab.size() * 8 + 64))) return rc;
        u64 *dtab = reinterpret_cast<u64 *>(ctx->dfx_gather.p);
        if ((rc = h2d_table(ctx, dtab, tab.data(), tab.size() * 8))) return rc;
        // (a few of them: the exact parallel encoder, as a batch of that size would get on its own — 16 noise streams 5 ms instead of 11.5)
        int src = AUKIT_OK;
        if (dfpwm_encode_i8_small(ctx, reinterpret_cast<const signed char *>(J.src), tab.data(), tab.data() + m, m, out, tab.data() + 2 * (size_t)m, &src)) { if (src) return src; }
        else if ((rc = dfpwm_encode_i8(ctx, reinterpret_cast<const signed char *>(J.src), dtab, dtab + m, m, out, dtab + 2 * (size_t)m))) return rc;
    } else if (m) {
        const aukit_batch *in = J.in;
        // Hard streams — the encoder changes its class every few chunks (noise) or more often than there are rounds: the schedule with one
        // encoder lane per stream does not care.  Their bytes are gathered into a batch of their own, transcoded by
        // aukit_dfpwm_transcode_mono with this schedule switched off, and scattered to their places.
        std::vector<uint64_t> sub_off(m + 1, 0), tab(3 * (size_t)m);
        for (unsigned i = 0; i < m; i++) {
            const uint64_t len = in->off[hs[i] + 1] - in->off[hs[i]];
            tab[i] = in->off[hs[i]]; tab[m + i] = sub_off[i]; tab[2 * (size_t)m + i] = len;
            sub_off[i + 1] = sub_off[i] + len;
        }
        if ((rc = ctx->dfx_gather.ensure((size_t)sub_off[m] + 64 + tab.size() * 8))) return rc;
        unsigned char *gbuf = reinterpret_cast<unsigned char *>(ctx->dfx_gather.p);
        u64 *dtab = reinterpret_cast<u64 *>(gbuf + round_up(sub_off[m] + 16, 64));
        uint64_t maxlen = 1;
        for (unsigned i = 0; i < m; i++) maxlen = std::max(maxlen, tab[2 * (size_t)m + i]);
        if ((rc = h2d_table(ctx, dtab, tab.data(), tab.size() * 8))) return rc;
        for (unsigned i0 = 0; i0 < m; i0 += 65535)  // (gridDim.y limit; the table is indexed from i0 by pointer offset: src | dst | len planes of m)
            hipLaunchKernelGGL(k_dfx_move, dim3((unsigned)std::min<uint64_t>((maxlen + 4095) / 4096, 64), std::min<unsigned>(65535, m - i0)), dim3(256), 0, ctx->stream, in->data(), gbuf, dtab + i0, m);
        aukit_batch *sub_in = nullptr;
        if ((rc = aukit_batch_wrap_device(ctx, &sub_in, gbuf, sub_off.data(), m))) return rc;
        // (the nested call is an ordinary entry point: it must neither move the outer call's timing origin nor overwrite its counters — ADVICE r05)
        uint64_t keep[8];
        std::copy(ctx->counters, ctx->counters + 8, keep);
        ctx->dfx_disable = true;
        ctx->ktiming_nested++;
        rc = aukit_dfpwm_transcode_mono(ctx, sub_in, 2, &ctx->dfx_sub_out);
        ctx->ktiming_nested--;
        ctx->dfx_disable = false;
        std::copy(keep, keep + 8, ctx->counters);
        aukit_batch_free(sub_in);
        if (rc) return rc;
        const aukit_batch *so = ctx->dfx_sub_out;
        for (unsigned i = 0; i < m; i++) { tab[i] = so->off[i]; tab[m + i] = h_ooff[hs[i]]; tab[2 * (size_t)m + i] = so->off[i + 1] - so->off[i]; }
        if ((rc = h2d_table(ctx, dtab, tab.data(), tab.size() * 8))) return rc;
        for (unsigned i0 = 0; i0 < m; i0 += 65535)
            hipLaunchKernelGGL(k_dfx_move, dim3((unsigned)std::min<uint64_t>((maxlen / 2 + 4095) / 4096 + 1, 64), std::min<unsigned>(65535, m - i0)), dim3(256), 0, ctx->stream, so->data(), out, dtab + i0, m);
        AUKIT_HIP_CHECK(hipGetLastError());
    }
    *taken = true;
    return AUKIT_OK;
}

// aukit.dfpwm(d, 2, rate):mono():dfpwm() on a batch of DFPWM bytes (aukit_dfpwm_transcode_mono)
int dfpwm_transcode_spec(aukit_ctx *ctx, const aukit_batch *in, unsigned char *out, const u64 *d_ooff, const uint64_t *h_ooff, bool *taken) {
    DfxJob J;
    J.kind = 0; J.src = in->data(); J.in = in;
    J.off.assign(in->off.begin(), in->off.begin() + in->n);
    J.fed.resize(in->n);
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        J.fed[s] = nb ? nb + (nb + 6000 - 1) / 6000 - 1 : 0;  // 6001-byte slices advanced by 6000 (Q10)
    }
    return dfx_run(ctx, J, out, d_ooff, h_ooff, taken);
}

// Audio:dfpwm on int8 samples (aukit_dfpwm_encode behind k_dfpwm_quantize): rows at h_in_off (16-byte aligned), h_count samples each
int dfpwm_encode_spec(aukit_ctx *ctx, const signed char *rows, const uint64_t *h_in_off, const uint64_t *h_count, uint32_t n, unsigned char *out, const u64 *d_ooff,
                      const uint64_t *h_ooff, bool *taken) {
    DfxJob J;
    J.kind = 1; J.src = reinterpret_cast<const unsigned char *>(rows);
    J.off.assign(h_in_off, h_in_off + n);
    J.count.assign(h_count, h_count + n);
    J.fed.resize(n);
    for (uint32_t s = 0; s < n; s++) J.fed[s] = h_count[s] / 4;
    return dfx_run(ctx, J, out, d_ooff, h_ooff, taken);
}

}  // namespace aukit
