// flac_dev.h — what the two FLAC decoders (flac.hip: extract → chain → restore; flac_fused.hip: decode → chain → gather) share: stream / candidate /
// frame records, the byte-position hash table, the error codes of decodeFrame (aukit.lua:510-567).
#pragma once
#include "resample.h"

namespace aukit {

typedef unsigned long long u64;

enum FlacErr { FE_OK = 0, FE_EOF_START = 1 /* readByte() == nil at a frame start: clean end */, FE_NIL = 2, FE_SYNC = 3, FE_BLOCKSIZE = 4,
               FE_CHAN = 5, FE_SUBTYPE = 6, FE_RESMETHOD = 7, FE_PARTITION = 8,
               FE_LIMIT = 9 /* the candidate ran past its bit budget; extracted again without a budget if the chain needs it */,
               FE_DECLINE = 10 /* k_flac_decode (flac_fused.hip) does not serve this frame's shape: the batch goes through k_flac_extract + k_flac_restore */ };
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
enum { FLAG_OVERFLOW = 1, FLAG_INTERNAL = 2 };
struct FlacStreamInfo { u64 first_byte; double rate, nsamples; int channels, depth, status, pad; };

struct FlacGlobals {
    const unsigned char *src;  // batch data
    const u64 *w0;             // 16-byte aligned base at or below src
    u64 base_bit;              // bit offset of src relative to w0
    u64 safe_words;            // even; 16-byte vectors [w0 + 2k, w0 + 2k + 2) with 2k < safe_words touch the batch (same page as a valid byte)
    const u64 *off;
    const FlacStreamInfo *info;
};
struct Cand { unsigned stream; unsigned nolimit; u64 byte; };  // byte: absolute in the batch

// byte position → candidate index (open addressing, linear probing; empty key = ~0)
struct CandHash { u64 *keys; unsigned *vals; unsigned shift; u64 mask; };
AUKIT_DEV u64 hash_slot(const CandHash &H, u64 key) { return (key * 0x9E3779B97F4A7C15ull) >> H.shift; }
AUKIT_DEV void hash_insert(const CandHash &H, u64 key, unsigned val) {
    u64 h = hash_slot(H, key);
    for (;;) {
        const u64 prev = atomicCAS(&H.keys[h], ~0ull, key);
        if (prev == ~0ull) { H.vals[h] = val; return; }
        if (prev == key) return;
        h = (h + 1) & H.mask;
    }
}
AUKIT_DEV unsigned hash_lookup(const CandHash &H, u64 key) {
    u64 h = hash_slot(H, key);
    for (;;) {
        const u64 k = H.keys[h];
        if (k == key) return H.vals[h];
        if (k == ~0ull) return ~0u;
        h = (h + 1) & H.mask;
    }
}
struct CandInfo {
    u64 end_byte;     // absolute byte after the frame's CRC-16
    u64 scratch;      // element offset of subframe 0 in the scratch array; subframe c at + c * blocksize
    u64 sample_off;   // chain: samples of the stream before this frame
    int blocksize, chan_asgn, status, nsub;
    unsigned seq, used;
};
struct SubDesc { int order, lshift, wasted, kind; short coef[32]; };  // kind: 0 = no prediction, 1/2/3 = order <= 4/12/32
struct ChainOut { u64 L, miss_at; unsigned nframes; int status; int miss_kind; unsigned miss_ci; int bs0, uniform; };   // bs0: the first frame's block size; uniform: every frame but the last has it, the last is no longer  // miss_kind: 1 = no candidate at miss_at, 2 = candidate miss_ci hit its bit budget
struct SubJob { u64 src, dst; unsigned desc; int bs; int asgn, pad; };  // asgn: the frame's channel assignment when it decorrelates (8..10), else 0
struct FrameRec { u64 sample_off; u64 scratch; int bs, chan_asgn; unsigned stream, end_rel; };   // end_rel: the byte behind the frame, relative to its stream's start (streams below 4 GiB; 0: not recorded)   // scratch: where k_flac_decode left the frame's final values (channel c at + c * bs); 0 for the first design
struct Carve {
    size_t at = 0;
    size_t take(size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; }
};
struct Counters { u64 ncand, scratch_cursor, kind_count[16], kind_fill[16]; unsigned flags, ticket; u64 stats[8]; };   // stats: -DAUKIT_FLAC_STATS builds of k_flac_decode count their rounds and turns here (tools/r06_flac_stats.sh)

// ---- flac_fused.hip: decodeFrame (:510-567) with the prediction (:411-419), the wasted-bits shift (:467-469), the stereo decorrelation and the
// wrap (:482-507) done by the lane that reads the frame's bits — final integers leave the kernel once (k_flac_decode)
struct FusedArgs {
    FlacGlobals G;
    const Cand *cands;
    unsigned first, count;   // candidates [first, first + count)
    CandInfo *ci;
    int C, depth;            // channels / bit depth of the batch
    int *scratch;            // final values: frame k's channel c at ci[k].scratch + c * blocksize
    u64 scratch_cap;         // elements
    u64 *scratch_cursor;
    unsigned *flags;
    int limit_factor;
    unsigned *ticket;
    int out16;               // finals as int16 (k_flac_decode<..., O16>; depths <= 16)
    u64 *stats;              // Counters::stats
    int dbg;                 // ablation switches (AUKIT_FLAC_FUSED_DBG; wrong results): 1 no prediction, 2 no stores, 4 no read-back of parked values
};
int flac_fused_launch(aukit_ctx *ctx, const FusedArgs &A);
// flac_stream.hip (round 6): the same contract, values in registers from the bit stream to the store
int flac_stream_launch(aukit_ctx *ctx, const FusedArgs &A);
// flac_pq.hip (round 6): the same contract, a parser wave and a predictor wave per 64 frames — small batches
int flac_pq_launch(aukit_ctx *ctx, const FusedArgs &A);
// the chained frames' records in stream order (one lane per candidate)
int flac_frames_launch(aukit_ctx *ctx, const Cand *cands, const CandInfo *ci, unsigned ncand, const u64 *frame_base, FrameRec *frames, const u64 *stream_off);
// chained frames: scratch → contiguous int32 rows (one workgroup per frame record)
int flac_gather_launch(aukit_ctx *ctx, const FrameRec *frames, u64 nfr, int C, const u64 *row_off, const int *scratch, int *rows, bool scratch16 = false);   // scratch16: the frames hold int16 finals (FusedArgs::out16)
// the same with the loader's conversion `s / 2^depth` (:505) into the rows of an audio (dtype AUKIT_F32 / AUKIT_F64; a_meta = len[n], row_off[n], row_stride[n])
int flac_gather_convert_launch(aukit_ctx *ctx, const FrameRec *frames, u64 nfr, int C, const int *scratch, const u64 *a_meta, unsigned n, void *out, int dtype, double full, bool scratch16 = false);

}  // namespace aukit
