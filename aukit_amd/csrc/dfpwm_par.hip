// dfpwm_par.hip — chunk-parallel DFPWM decoding, bit-exact; the serial int8 encoder; and an exact parallel encoder for small batches
// (second half of the file).
//
// aukit.dfpwm feeds one decoder per stream, serially (aukit.lua:1399-1412).  One lane per stream leaves most of the GPU idle for
// batches below ~10^5 streams, and speculation on the whole state does not work: the strength counter only forgets its start at
// its floor, which noisy input rarely touches.  But the decoder's strength is a saturating ±1 counter driven by
// the INPUT bits alone (bit == previous bit ? min(s + 1, 1023) : max(s - 1, 8)), i.e. a composition of clamp-add maps
// (a, lo, hi) — associative, so it is known exactly at any position after a scan:
//   1. k_df_blockmaps   lane per (stream, map block): the composite clamp-add map of the bytes between two chunk warm-up starts;
//   2. k_df_blockscan   lane per stream: strength at every block start;
//   3. k_df_chunks      lane per (stream, chunk): starts one block early with the exact strength and previous bit, charge and
//                       filter state zero — the charge update is a contraction towards the target ((1 - s/1024) per step, plus
//                       a nudge) and the low-pass forgets at 116/256 per step, so after the 8192 warm-up steps the state is the
//                       true one in practice — records the state it reached, decodes its chunk, records the end state;
//   4. k_df_verify      wave per stream: where a chunk's recorded start state differs from the true end state of the chunk
//                       before it, that chunk is decoded again serially from the true state.  Exactness therefore never rests
//                       on the convergence argument; only the speed does (AUKIT_DFPWM_STATS=1 prints the redo count).
// Output: de-interleaved int8 rows (the loader), or the stereo → mono mix of aukit.pcm ∘ Audio:mono ∘ encodePCM as int8 (transcode).
#include <algorithm>
#include "common.h"
#include "dfpwm_dev.h"
#include "dfpwm_par_dev.h"
#ifndef AUKIT_DFF_R
#define AUKIT_DFF_R 4  // rounds of 64 samples per turn of the fused transcode's encoder (2 or 4)
#endif

namespace aukit {


// The eight clamp-add steps of one byte, given the bit before it, compose to one clamp-add map: 512 table entries per workgroup.
// (one dword per entry: eight unit steps inside [8, 1023] give a in [-8, 8], lo in [8, 16], hi in [1015, 1023]; as three ints the random
// look-ups of a wave spent 70 % of their LDS cycles in bank conflicts — SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r02_dfpwm_pmc_lds.csv)
// <NQ, SINGLE>: 16-byte vectors per turn, and whether the turn's loads are asked for at its top (no second buffer) or a turn ahead.
//   <4, false>  64 bytes per turn, the next 64 in flight: the short runs of a finely cut batch (their first loads are most of their latency)
//   <8, true>   a whole 128-byte line per turn, fetched ONCE: with four vectors the two halves of a line are two turns apart, and the lines of
//               262 144 lanes are more than the L2s hold — 4.1 GB fetched for 2.0 of input at 16 384 streams (profiles/r05_dfpwm_pmc_fetch.csv
//               before / after); with eight vectors AND a second buffer the kernel holds 145 VGPRs, three waves per SIMD, 1.25 ms for 0.97
template <int NQ, bool SINGLE>
__global__ __launch_bounds__(256) void k_df_blockmaps(const DfParParams P) {
    __shared__ unsigned bm[512];
    for (int e = threadIdx.x; e < 512; e += 256) {   // entry (byte << 1) | previous bit: nine consecutive bits of the stream
        unsigned byte = (unsigned)e >> 1;
        int prev = e & 1;
        SatMap f{0, -(1 << 28), 1 << 28};
        for (int k = 0; k < 8; k++) {
            const int bit = byte & 1;
            byte >>= 1;
            f = sm_then(f, SatMap{bit == prev ? 1 : -1, 8, 1023});
            prev = bit;
        }
        bm[e] = (unsigned)f.lo | ((unsigned)f.hi << 12) | ((unsigned)f.a << 24);   // a field each, unbiased: lo & 0xFFF, bits 12..23, the top byte signed — three instructions a look-up (six with biased 5 / 4 / 4-bit fields)
    }
    __syncthreads();
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    // a lane per (block, piece, stream): a large batch is cut into few chunks per stream — eight at 16 384 streams, two waves per SIMD if a block
    // were one lane's, each walking 15 000 bytes behind its LDS look-ups
    const unsigned F = P.msub ? P.msub : 1;
    const unsigned bp = (unsigned)(gid / P.n), s = (unsigned)(gid - (u64)bp * P.n);
    const unsigned blk = bp / F, sub = bp - blk * F;
    if (blk >= P.nblk) return;
    SatMap *const dst = P.maps + ((size_t)s * P.nblk + blk) * F + sub;
    if (P.skip_last && blk + 1 == P.nblk) { *dst = SatMap{0, -(1 << 28), 1 << 28}; return; }
    const unsigned char *p = P.src + P.off[s];
    const u64 fed = P.fed[s];
    const u64 e1 = dfp_chunk_start(P, blk + 1) - P.W;
    u64 f0 = blk ? dfp_chunk_start(P, blk) - P.W : 0, f1 = e1 < fed ? e1 : fed;
    if (F > 1 && f0 < f1) {   // pieces of whole 16-byte vectors
        const u64 piece = (((f1 - f0 + F - 1) / F) + 15) & ~15ull;
        f0 = f0 + sub * piece < f1 ? f0 + sub * piece : f1;
        f1 = f0 + piece < f1 ? f0 + piece : f1;
    }
    SatMap f{0, -(1 << 28), 1 << 28};
    bool anyflat = false;
    if (f0 < f1) {
        int prev = f0 ? (p[dfp_src_index(f0 - 1, P.feed)] >> 7) & 1 : 0;
        u64 b = f0;
        if (b == 0) {  // from the reset state (strength 0) the first step lands on 8 either way: do the stream's first byte bit by bit
            unsigned byte = p[0];
            for (int k = 0; k < 8; k++) {
                const int bit = byte & 1;
                byte >>= 1;
                f = sm_then(f, k == 0 ? SatMap{0, 8, 8} : SatMap{bit == prev ? 1 : -1, 8, 1023});
                prev = bit;
            }
            b = 1;
        }
        auto look = [&](unsigned at4) {   // at4 = 4 * entry: the LDS byte address
            const unsigned e = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(bm) + at4);
            f = sm_then(f, SatMap{(int)e >> 24, (int)(e & 0xFFFu), (int)((e >> 12) & 0xFFFu)});
        };
        auto step = [&](unsigned byte) {
            look((byte << 3) | ((unsigned)prev << 2));
            prev = byte >> 7;
        };
        // four bytes: the entries are nine-bit windows of the word (the first one borrows the bit before it), a shift and a mask each
        auto step4 = [&](unsigned word) {
            look(((word << 3) | ((unsigned)prev << 2)) & 0x7FCu);
            look((word >> 5) & 0x7FCu);
            look((word >> 13) & 0x7FCu);
            look((word >> 21) & 0x7FCu);
            prev = word >> 31;
        };
        // Run by run of the feed (inside a run the source is contiguous; a piece of a large batch's map block is longer than a run — and went
        // through fed_for_each byte by byte until this was a loop): 64 or 128 bytes per turn.  With one 16-byte vector ahead (fed_for_each) a lane
        // had 16 bytes in flight and the kernel moved 0.8 TB/s, waiting 66 % of its time.
        for (u64 k0 = b / P.feed.run; b < f1; k0++) {
            const u64 seg_end = (k0 + 1) * P.feed.run < f1 ? (k0 + 1) * P.feed.run : f1;
            const unsigned char *a = p + k0 * P.feed.stride + (b - k0 * P.feed.run);
            u64 rem = seg_end - b;
            b = seg_end;
            while (rem && ((uintptr_t)a & 15)) { step((unsigned)*a); a++; rem--; }
            // (whole turns start on a multiple of their size: with eight vectors a turn is one 128-byte line, fetched once — with four, the two
            // halves of a line are two turns apart, and 262 144 lanes' lines are more than the L2s hold: 4.1 GB fetched for 2.0 of input)
            while (rem >= 16 && ((uintptr_t)a & (16 * NQ - 1))) {
                const uint4 q = *reinterpret_cast<const uint4 *>(a);
                step4(q.x); step4(q.y); step4(q.z); step4(q.w);
                a += 16; rem -= 16;
            }
            uint4 cur[NQ], nxt[NQ];
            [[maybe_unused]] const unsigned char *const a0 = a;
            if (rem >= 16 * NQ) {
#pragma unroll
                for (int q = 0; q < NQ; q++) cur[q] = reinterpret_cast<const uint4 *>(a)[q];
            }
            while (rem >= 16 * NQ) {
                const bool more = !SINGLE && rem >= 32 * NQ;
                if (SINGLE && a != a0) {   // (the turn's own loads, asked for at its top: the other waves cover them)
#pragma unroll
                    for (int q = 0; q < NQ; q++) cur[q] = reinterpret_cast<const uint4 *>(a)[q];
                }
                if (more) {
#pragma unroll
                    for (int q = 0; q < NQ; q++) nxt[q] = reinterpret_cast<const uint4 *>(a + 16 * NQ)[q];
                }
                if (P.inner) {   // (a turn of one byte repeated: digital silence)
                    const unsigned x = cur[0].x;
                    bool flat = x == (x & 0xFFu) * 0x01010101u;
#pragma unroll
                    for (int q = 0; q < NQ; q++) flat = flat && cur[q].x == x && cur[q].y == x && cur[q].z == x && cur[q].w == x;
                    anyflat = anyflat || flat;
                }
#ifdef AUKIT_DF_MAPS_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
                for (int q = 0; q < NQ; q++) {   // (unrolled: as a loop, cur[q] is an array indexed by a variable — scratch memory, whose loads wait vmcnt(0): for the prefetch too)
                    const unsigned w4[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#ifdef AUKIT_DF_MAPS_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
                    for (int w = 0; w < 4; w++) {
                        step4(w4[w]);
                    }
                }
                if (more) {
#pragma unroll
                    for (int q = 0; q < NQ; q++) cur[q] = nxt[q];
                }
                a += 16 * NQ; rem -= 16 * NQ;
            }
            while (rem) { step((unsigned)*a); a++; rem--; }
        }
    }
    *dst = f;
    if (P.inner && anyflat && f0 >= (P.lead_on ? (u64)P.lead_on[s] : 0ull)) P.inner[s] = 1;
}

__global__ __launch_bounds__(64) void k_df_blockscan(const DfParParams P) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= P.n) return;
    int st = P.init ? P.init[(s % (unsigned)P.init_n) * 6 + 1] : 0;
    int *o = P.s_start + (size_t)s * (P.nblk + 1);
    o[0] = st;
    const unsigned F = P.msub ? P.msub : 1;
    const SatMap *m = P.maps + (size_t)s * P.nblk * F;
    for (unsigned b = 0; b < P.nblk; b++) {
        for (unsigned j = 0; j < F; j++) st = sm_apply(m[(size_t)b * F + j], st);
        o[b + 1] = st;
    }
}


__global__ __launch_bounds__(256) void k_df_chunks(const DfParParams P) {
    extern __shared__ signed char lut[];  // 64 KiB in mix mode only (rows mode runs at full occupancy)
    if (P.mode == 1) {
        for (int i = threadIdx.x; i < 65536; i += blockDim.x) lut[i] = (signed char)dfp_mix((i >> 8) - 128, (i & 255) - 128);
        __syncthreads();
    }
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned cr = (unsigned)(gid / P.n), s = (unsigned)(gid - (u64)cr * P.n);  // a wave = one chunk index of 64 streams
    const unsigned c = cr + P.c_lo;
    if (c >= P.c_hi) return;
    const unsigned char *p = P.src + P.off[s];
    const u64 fed = P.fed[s];
    const u64 f0 = dfp_chunk_start(P, c), e1 = dfp_chunk_start(P, c + 1), f1 = e1 < fed ? e1 : fed;
    int *ss = P.st_start + ((size_t)s * P.nchunk + c) * 6, *se = P.st_end + ((size_t)s * P.nchunk + c) * 6;
    if (f0 >= fed && c > 0) { ss[1] = -1; se[1] = -1; return; }
    const DfOut O = dfp_out(P, s, lut);
    DfDec d{};
    if (c == 0 && P.init) dfp_unpack(P.init + (s % (unsigned)P.init_n) * 6, d);
    if (c > 0) {  // warm-up over the block before the chunk: exact strength and previous bit, everything else from zero
        const u64 fw = f0 - P.W;
        d.p.strength = P.s_start[(size_t)s * (P.nblk + 1) + c];
        d.p.pb = fw ? (int)((p[dfp_src_index(fw - 1, P.feed)] >> 6) & 2) - 1 : -1;
        dfp_run<false>(p, fw, f0, d, O);
    }
    dfp_pack(d, ss);
    dfp_run<true>(p, f0, f1, d, O);
    dfp_pack(d, se);
}

// A WAVE per stream (a lane per stream walked the chunks one dependent load after the other — 235 chunks of a short-chunk batch 0.15 ms, more
// than the chunk lanes took beside 63 idle SIMDs): 64 chunks' recorded start states are compared at once with the recorded end states before
// them; a mismatch is decoded again from the true state by lane 0, and its new end state decides about the chunk behind it.
__global__ __launch_bounds__(64) void k_df_verify(const DfParParams P) {
    const unsigned s = blockIdx.x, lane = threadIdx.x;
    if (s >= P.n) return;
    const unsigned char *p = P.src + P.off[s];
    const u64 fed = P.fed[s];
    const DfOut O = dfp_out(P, s, nullptr);
    const unsigned c_first = P.c_lo > 1 ? P.c_lo : 1;  // the chunk before it was verified by the slice before this one (chunk 0 starts from the reset state: always true)
    const int *SS = P.st_start + (size_t)s * P.nchunk * 6;
    int *SE = P.st_end + (size_t)s * P.nchunk * 6;
    if (SE[(size_t)(c_first - 1) * 6 + 1] < 0) return;  // the stream ended before this slice
    unsigned redone = 0, chunks = P.c_lo == 0 ? 1 : 0;
    bool carry = false;   // the last chunk of the group before was decoded again: `ce` is its true end state
    int ce[6] = {0, 0, 0, 0, 0, 0};
    for (unsigned c0 = c_first; c0 < P.c_hi; c0 += 64) {
        const unsigned c = c0 + lane;
        const bool valid = c < P.c_hi;
        int ss[6], pe[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { ss[i] = valid ? SS[(size_t)c * 6 + i] : -1; pe[i] = valid ? SE[(size_t)(c - 1) * 6 + i] : 0; }
        if (carry && lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) pe[i] = ce[i];
        }
        carry = false;
        const u64 none = __ballot(!valid || ss[1] < 0);          // the walk ends at the stream's first missing chunk
        const unsigned jx = none ? (unsigned)__builtin_ctzll(none) : 64u;
        const bool exists = lane < jx;
        chunks += jx;
        bool same = true;
#pragma unroll
        for (int i = 0; i < 6; i++) same = same && ss[i] == pe[i];
        u64 mask = __ballot(exists && !same);
        while (mask) {
            const unsigned j = (unsigned)__builtin_ctzll(mask);
            mask &= mask - 1;
            int truth[6];
#pragma unroll
            for (int i = 0; i < 6; i++) truth[i] = __shfl(pe[i], (int)j);
            const unsigned cc = c0 + j;
            if (lane == 0) {
                DfDec d;
                dfp_unpack(truth, d);
                const u64 f0 = dfp_chunk_start(P, cc), e1 = dfp_chunk_start(P, cc + 1), f1 = e1 < fed ? e1 : fed;
                if (P.mode == 1) {  // no table in this kernel: mix computed on the spot (and repeated dwords skipped: DfRepeat)
                    u64 i = 4 * f0;
                    auto four = [&](unsigned byte) -> unsigned {
                        unsigned packed = 0;
                        for (int k = 0; k < 4; k++) {
                            const int l = df_decode_bit(d, byte & 1), r = df_decode_bit(d, (byte >> 1) & 1);
                            byte >>= 2;
                            packed |= ((unsigned)(unsigned char)dfp_mix(l, r)) << (8 * k);
                        }
                        return packed;
                    };
                    DfRepeat rp;
                    fed_for_each<true>(p, f0, f1, P.feed,
                                 [&](unsigned byte) { *reinterpret_cast<unsigned *>(O.base + i) = four(byte); i += 4; rp.fixed = false; },
                                 [&](unsigned word) {
                                     dfp_u32x4a v;
                                     if (rp.hit(word)) v = rp.va;
                                     else {
                                         rp.before(d);
                                         v.x = four(word & 0xFF); v.y = four((word >> 8) & 0xFF); v.z = four((word >> 16) & 0xFF); v.w = four(word >> 24);
                                         rp.after(word, d);
                                         rp.va = v;
                                     }
                                     *reinterpret_cast<dfp_u32x4a *>(O.base + i) = v;
                                     i += 16;
                                 });
                } else dfp_run<true, true>(p, f0, f1, d, O);
                dfp_pack(d, truth);
                for (int i = 0; i < 6; i++) SE[(size_t)cc * 6 + i] = truth[i];
            }
            redone++;
#pragma unroll
            for (int i = 0; i < 6; i++) truth[i] = __shfl(truth[i], 0);   // the chunk's true end state
            if (j + 1 < 64) {   // the chunk behind it is judged by that, not by the end state the chunk lane had recorded
                bool again = false;
                if (lane == j + 1 && exists) {
                    bool sm = true;
#pragma unroll
                    for (int i = 0; i < 6; i++) { pe[i] = truth[i]; sm = sm && ss[i] == truth[i]; }
                    again = !sm;
                }
                mask = (mask & ~(1ull << (j + 1))) | __ballot(again);
            } else {
                carry = true;
#pragma unroll
                for (int i = 0; i < 6; i++) ce[i] = truth[i];
            }
        }
        if (jx < 64) break;
    }
    if (lane == 0) {
        if (redone) atomicAdd(&P.stats[0], redone);
        atomicAdd(&P.stats[1], chunks);
    }
}

// the exact strength at every chunk's warm-up start (P.s_start), for planners outside this file (dfpwm_spec.hip)
int dfpwm_strength_scan(aukit_ctx *ctx, const DfParParams &P) {
    const unsigned F = P.msub ? P.msub : 1;
    const dim3 grid((unsigned)(((size_t)P.n * P.nblk * F + 255) / 256));
    if ((uint64_t)P.bpc * P.W / F >= 4096) hipLaunchKernelGGL((k_df_blockmaps<8, true>), grid, dim3(256), 0, ctx->stream, P);   // long runs per lane: whole lines
    else hipLaunchKernelGGL((k_df_blockmaps<4, false>), grid, dim3(256), 0, ctx->stream, P);
    hipLaunchKernelGGL(k_df_blockscan, dim3((P.n + 63) / 64), dim3(64), 0, ctx->stream, P);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

// Audio:dfpwm on int8 mono samples (the transcode's second half): one encoder per stream, serial (its bits depend on its state)
__global__ __launch_bounds__(64) void k_dfpwm_encode_i8(const signed char *in, const u64 *in_off, const u64 *count, unsigned n, unsigned char *out, const u64 *ooff) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const signed char *p = in + in_off[s];  // 16-byte aligned (host)
    const u64 L = count[s];
    unsigned char *o = out + ooff[s];
    DfEnc e{};
    u64 w = 0;
    // 64 samples per round, the next 64 in flight meanwhile: a lone wave per SIMD has nothing else to hide the load latency with,
    // and 16 samples are only ~900 cycles of work
    uint4 cur[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = 16u * k < L ? *reinterpret_cast<const uint4 *>(p + 16 * k) : make_uint4(0, 0, 0, 0);
    u64 i = 0;
    typedef unsigned u32x2u __attribute__((ext_vector_type(2), aligned(1)));  // the packed rows start anywhere: one unaligned 8-byte store per round
    for (; i + 64 <= L; i += 64) {                                            // (a byte per store instruction was 64 cache lines for 64 bytes)
        uint4 nxt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) nxt[k] = i + 64 + 16 * k < L ? *reinterpret_cast<const uint4 *>(p + i + 64 + 16 * k) : make_uint4(0, 0, 0, 0);
        unsigned ob[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned words[4] = {cur[k].x ^ 0x80808080u, cur[k].y ^ 0x80808080u, cur[k].z ^ 0x80808080u, cur[k].w ^ 0x80808080u};  // u = v + 128
#pragma unroll
            for (int h = 0; h < 2; h++) {
                unsigned byte = 0;
#pragma unroll
                for (int b = 0; b < 8; b++) byte |= df_encode_u(e, (words[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFF) & (1u << b);
                ob[k >> 1] |= (byte & 255u) << (8 * (2 * (k & 1) + h));
            }
        }
        u32x2u v; v.x = ob[0]; v.y = ob[1];
        *reinterpret_cast<u32x2u *>(o + w) = v;
        w += 8;
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
    for (int k = 0; k < 4 && i < L; k++, i += 16) {  // the last < 64 samples
        const unsigned words[4] = {cur[k].x ^ 0x80808080u, cur[k].y ^ 0x80808080u, cur[k].z ^ 0x80808080u, cur[k].w ^ 0x80808080u};
        for (int h = 0; h < 2 && i + 8 * h < L; h++) {
            unsigned byte = 0;
            for (int b = 0; b < 8; b++) {
                const u64 idx = i + 8 * h + b;
                const unsigned u = idx < L ? (words[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFF : 128u;  // the last byte is padded with samples of value 0
                byte |= df_encode_u(e, u) & (1u << b);
            }
            o[w++] = (unsigned char)byte;
        }
    }
}

// The same encoder over the samples [lo, hi) of every stream (lo, hi multiples of 64; `last`: to the end of the stream, tail and padding
// included), its state carried in `state` ([n] × {charge + 128, strength, previous bit}): the slices of a batch are encoded one after
// the other on their own HIP stream while the decoder works on the next slice (dfpwm_transcode_sliced below).
// Output: `out` is a STAGING buffer with one row of `ostride` bytes (a multiple of 16) per stream — eight samples make a byte and a round of 64
// samples one aligned 8-byte store per lane.  (Storing byte by byte straight into the packed result — rows 60 010 bytes apart, 64 lanes,
// 64 cache lines per instruction — is what made the encoder lose a factor of 2-4 as soon as it shared a CU's memory path with anything:
// traced in profiles/.)  k_dfpwm_compact moves the rows to their packed places afterwards.
__global__ __launch_bounds__(64) void k_dfpwm_encode_i8_slice(const signed char *in, const u64 *in_off, const u64 *count, unsigned n, unsigned char *out, u64 ostride, int *state,
                                                              u64 lo, u64 hi, int first, int last) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const u64 L = count[s];
    hi = last ? L : (hi < L ? hi : L & ~63ull);  // a stream that ends inside a middle slice: whole rounds only, the rest belongs to the last slice
    const signed char *p = in + in_off[s];  // 16-byte aligned (host)
    unsigned char *o = out + (u64)s * ostride;
    DfEnc e{};
    u64 i = 0;  // the stream's own progress (a short stream is not where the slice starts)
    int *st = state + 6 * (size_t)s;
    if (!first) { e.cu = st[0]; e.strength = st[1]; e.pb = st[2]; i = (u64)(unsigned)st[4] | (u64)(unsigned)st[5] << 32; }
    (void)lo;
    u64 w = i >> 3;
    uint4 cur[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = i + 16u * k < L ? *reinterpret_cast<const uint4 *>(p + i + 16 * k) : make_uint4(0, 0, 0, 0);
    for (; i + 64 <= hi; i += 64) {
        uint4 nxt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) nxt[k] = i + 64 + 16 * k < L ? *reinterpret_cast<const uint4 *>(p + i + 64 + 16 * k) : make_uint4(0, 0, 0, 0);
        unsigned ob[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned words[4] = {cur[k].x ^ 0x80808080u, cur[k].y ^ 0x80808080u, cur[k].z ^ 0x80808080u, cur[k].w ^ 0x80808080u};  // u = v + 128
#pragma unroll
            for (int h = 0; h < 2; h++) {
                unsigned byte = 0;
#pragma unroll
                for (int b = 0; b < 8; b++) byte |= df_encode_u(e, (words[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFF) & (1u << b);
                ob[k >> 1] |= (byte & 255u) << (8 * (2 * (k & 1) + h));
            }
        }
        *reinterpret_cast<uint2 *>(o + w) = make_uint2(ob[0], ob[1]);  // w is a multiple of 8 here
        w += 8;
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
    if (last) {
        for (int k = 0; k < 4 && i < L; k++, i += 16) {  // the last < 64 samples
            const unsigned words[4] = {cur[k].x ^ 0x80808080u, cur[k].y ^ 0x80808080u, cur[k].z ^ 0x80808080u, cur[k].w ^ 0x80808080u};
            for (int h = 0; h < 2 && i + 8 * h < L; h++) {
                unsigned byte = 0;
                for (int b = 0; b < 8; b++) {
                    const u64 idx = i + 8 * h + b;
                    const unsigned u = idx < L ? (words[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFF : 128u;  // the last byte is padded with samples of value 0
                    byte |= df_encode_u(e, u) & (1u << b);
                }
                o[w++] = (unsigned char)byte;
            }
        }
    }
    st[0] = e.cu; st[1] = e.strength; st[2] = e.pb; st[4] = (int)(unsigned)i; st[5] = (int)(unsigned)(i >> 32);
}

// staging rows (stride `sstride`) → the packed batch: stream s's `ooff[s + 1] - ooff[s]` bytes go to out + ooff[s]
__global__ __launch_bounds__(256) void k_dfpwm_compact(const unsigned char *stage, u64 sstride, unsigned char *out, const u64 *ooff, unsigned n) {
    const unsigned s = blockIdx.y;
    const u64 len = ooff[s + 1] - ooff[s];
    const unsigned char *src = stage + (u64)s * sstride;
    unsigned char *dst = out + ooff[s];
    // destination dwords: a head of up to 3 bytes, aligned dwords assembled from the (4-byte aligned) staging row, a tail
    const u64 head = (4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j * 4 < len + 4; j += (u64)gridDim.x * 256) {
        if (j == 0) { for (u64 k = 0; k < head && k < len; k++) dst[k] = src[k]; continue; }
        const u64 d0 = head + 4 * (j - 1);  // byte index in the row
        if (d0 >= len) continue;
        if (d0 + 4 <= len) {
            const u64 a = d0 & ~3ull;
            const unsigned sh = (unsigned)(d0 & 3) * 8;
            const unsigned lo = *reinterpret_cast<const unsigned *>(src + a), hi = *reinterpret_cast<const unsigned *>(src + a + 4);
            *reinterpret_cast<unsigned *>(dst + d0) = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
        } else for (u64 k = d0; k < len; k++) dst[k] = src[k];
    }
}

// host: plan + launch for `n` (pseudo-)streams given by their first source byte and fed byte count.  `out`: rows (mode 0, C
// channels, per-stream row offsets / strides as device arrays) or mono mix (mode 1, per-stream element offsets).  Returns false
// (nothing launched) when the batch is better served one lane per stream.
bool dfpwm_decode_parallel_feed(aukit_ctx *ctx, const unsigned char *src, const std::vector<uint64_t> &h_off, const std::vector<uint64_t> &h_fed, uint64_t run,
                                uint64_t stride, int mode, int C, signed char *out, const u64 *d_out_off, const u64 *d_out_stride, uint64_t lead, int *rc,
                                const DfSliceHook *hook) {
    const uint32_t n = (uint32_t)h_off.size();
    uint64_t fed_max = 0;
    for (uint64_t f : h_fed) fed_max = std::max(fed_max, f);
    uint64_t W = n >= 4096 ? 1024 : 512;  // small batches: more, shorter chunks
    const bool sliced = hook && hook->slices > 1 && !getenv("AUKIT_DFPWM_CHUNKS") && !getenv("AUKIT_DFPWM_BLOCK");
    if (sliced) W = 256;  // time slices need `slices` x more chunks per stream: shorter warm-up blocks keep the warm-up at 1/15 of a chunk
    if (const char *e = getenv("AUKIT_DFPWM_BLOCK")) W = std::max<uint64_t>(2, strtoull(e, nullptr, 10) & ~1ull);
    const unsigned nblk = (unsigned)((fed_max + W - 1) / W);
    // chunks per stream for one full round of lanes: 1024 per CU, 512 in mix mode (two workgroups per CU next to their 64 KiB tables)
    unsigned want = (unsigned)std::max<uint64_t>(1, (uint64_t)ctx->num_cus * (mode == 1 ? 512 : 1024) / std::max<uint32_t>(n, 1));
    if (sliced) want *= (unsigned)hook->slices;
    if (const char *e = getenv("AUKIT_DFPWM_CHUNKS")) want = (unsigned)std::max(1, atoi(e));
    unsigned bpc = std::max<unsigned>(nblk ? (nblk + want - 1) / want : 1, getenv("AUKIT_DFPWM_CHUNKS") ? 1u : 6u);  // warm-up (1 block) <= 1/6 of a chunk
    // ... for a batch that fills the chip.  A few streams leave most SIMDs idle and the launch takes the time of ONE lane: short chunks then,
    // whatever their warm-up share (one stream of ten seconds: 40 lanes of 7 blocks, 1.24 ms; 235 lanes of 2 blocks, a third of it) — the
    // shortest that keeps the lanes under 3/4 of a wave per SIMD
    if (!getenv("AUKIT_DFPWM_CHUNKS") && !getenv("AUKIT_DFPWM_BPC_MIN6"))
        for (unsigned b = std::max<unsigned>(nblk ? (nblk + want - 1) / want : 1, 1); b < 6; b++)
            if ((uint64_t)n * ((nblk + b - 1) / b) * 4 <= (uint64_t)ctx->num_cus * 4 * 64 * 3) { bpc = b; break; }
    const unsigned nchunk = nblk ? (nblk + bpc - 1) / bpc : 0;
    if (n == 0 || nchunk < 2 || getenv("AUKIT_DFPWM_SERIAL")) return false;
    // scratch: stream table, maps, strengths, states, stats
    const size_t b_tab = (size_t)n * 16, b_maps = (size_t)n * nchunk * sizeof(SatMap), b_ss = (size_t)n * (nchunk + 1) * 4, b_st = (size_t)n * nchunk * 6 * 4;
    if ((*rc = ctx->tmp_buf2.ensure(b_tab + b_maps + b_ss + 2 * b_st + 256 + 64))) return true;
    char *B = reinterpret_cast<char *>(ctx->tmp_buf2.p);
    if ((*rc = h2d_table(ctx, B, h_off.data(), (size_t)n * 8)) || (*rc = h2d_table(ctx, B + (size_t)n * 8, h_fed.data(), (size_t)n * 8))) return true;
    DfParParams P{};
    P.src = src; P.off = reinterpret_cast<const u64 *>(B); P.fed = P.off + n; P.feed = Feed{run, stride};
    P.n = n; P.nblk = nchunk; P.bpc = bpc; P.nchunk = nchunk; P.W = W;
    P.maps = reinterpret_cast<SatMap *>(B + b_tab); P.s_start = reinterpret_cast<int *>(B + b_tab + b_maps);
    P.st_start = reinterpret_cast<int *>(B + b_tab + b_maps + b_ss); P.st_end = reinterpret_cast<int *>(B + b_tab + b_maps + b_ss + b_st);
    P.stats = reinterpret_cast<unsigned *>(B + b_tab + b_maps + b_ss + 2 * b_st);
    P.init = nullptr;
    if (ctx->sb_dfpwm_on && !hook) {
        int *di = reinterpret_cast<int *>(B + b_tab + b_maps + b_ss + 2 * b_st + 256);
        int both[12];
        for (int i = 0; i < 6; i++) { both[i] = ctx->sb_dfpwm[i]; both[6 + i] = ctx->sb_dfpwm2[i]; }
        if ((*rc = h2d_table(ctx, di, both, sizeof both))) return true;
        P.init = di;
        P.init_n = ctx->sb_dfpwm_n == 2 ? 2 : 1;
    }
    P.mode = mode; P.C = C; P.out = out; P.out_off = d_out_off; P.out_stride = d_out_stride; P.lead = lead;
    if (hipMemsetAsync(P.stats, 0, 8, ctx->stream) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "hipMemsetAsync failed"); return true; }
    hipLaunchKernelGGL((k_df_blockmaps<4, false>), dim3((unsigned)(((size_t)n * nchunk + 255) / 256)), dim3(256), 0, ctx->stream, P);
    hipLaunchKernelGGL(k_df_blockscan, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, P);
    const unsigned cb = 256;  // (1024 lanes sharing one table, 8 waves per SIMD instead of 2: 7 % slower — the kernel is issue-bound)
    const unsigned nsl = sliced ? std::min<unsigned>((unsigned)hook->slices, nchunk) : 1u;
    for (unsigned k = 0; k < nsl; k++) {  // time slices: chunk indices [c_lo, c_hi) of every stream, in order
        P.c_lo = (unsigned)((u64)nchunk * k / nsl);
        P.c_hi = (unsigned)((u64)nchunk * (k + 1) / nsl);
        hipLaunchKernelGGL(k_df_chunks, dim3((unsigned)(((size_t)n * (P.c_hi - P.c_lo) + cb - 1) / cb)), dim3(cb), mode == 1 ? 65536 : 0, ctx->stream, P);
        hipLaunchKernelGGL(k_df_verify, dim3(n), dim3(64), 0, ctx->stream, P);
        if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "parallel DFPWM decode launch failed"); return true; }
        if (sliced && (*rc = hook->after_slice(k, nsl, (u64)P.c_lo * bpc * W, (u64)P.c_hi * bpc * W))) return true;  // fed bytes [lo, hi) of every stream are final
    }
    if (getenv("AUKIT_DFPWM_STATS") || ctx->collect_stats) {
        unsigned h[2] = {0, 0};
        (void)hipMemcpyAsync(h, P.stats, 8, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->counters[AUKIT_COUNTER_DFPWM_CHUNKS] = h[1]; ctx->counters[AUKIT_COUNTER_DFPWM_CHUNKS_REDONE] = h[0];
        if (getenv("AUKIT_DFPWM_STATS")) fprintf(stderr, "[dfpwm] %u streams x %u chunks of %u blocks of %llu fed bytes: %u of %u chunks redone serially\n", n, nchunk, bpc, (unsigned long long)W, h[0], h[1]);
    }
    *rc = AUKIT_OK;
    return true;
}

// aukit.dfpwm / stream.dfpwm on a batch: overlapping slices advanced by `adv`
bool dfpwm_decode_parallel(aukit_ctx *ctx, const aukit_batch *in, int mode, int C, signed char *out, const u64 *d_out_off, const u64 *d_out_stride, int *rc,
                           uint64_t adv, uint64_t lead, const DfSliceHook *hook) {
    std::vector<uint64_t> h_off(in->off.begin(), in->off.begin() + in->n), h_fed(in->n);
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        h_fed[s] = nb ? nb + (nb + adv - 1) / adv - 1 : 0;  // Σ min(adv + 1, nb - adv k)
    }
    return dfpwm_decode_parallel_feed(ctx, in->data(), h_off, h_fed, adv + 1, adv, mode, C, out, d_out_off, d_out_stride, lead, rc, hook);
}

// aukit.dfpwm(d, 2, rate):mono():dfpwm() on a large batch (config 4).  The encoder is one serial chain per stream (its bits depend on its
// state): ten seconds take it 11.7 ms whatever the batch, on the 256 SIMDs its 256 waves land on.  Run after the decoder (12.5 ms on the
// whole chip) that was 27 ms per step.  Here the batch is cut into time slices: the decoder works through chunk indices [c_lo, c_hi) of
// every stream, and as soon as a slice is verified the encoder follows on a second HIP stream (high priority: it is the critical
// path), carrying its per-stream state from slice to slice — at the end only the last slice's encoding is left.  Bit-identical bytes.
int dfpwm_transcode_sliced(aukit_ctx *ctx, const aukit_batch *in, signed char *mono, const u64 *d_moff, const u64 *d_mcount, unsigned char *out, const u64 *d_ooff,
                           const uint64_t *h_ooff, int slices, bool *taken) {
    *taken = false;
    int enc_cus = 0;  // CUs reserved for the encoder; 0 (default): decoder and encoder share every CU — see below, reserving lost
    if (const char *e = getenv("AUKIT_DFPWM_ENC_CUS")) enc_cus = std::max(0, std::min(atoi(e), ctx->num_cus - 8));
    if (!ctx->aux_stream || ctx->aux_enc_cus != enc_cus) {
        // Sharing every CU (the default): an encoder wave beside two decoder waves on one SIMD slows both by ~40 % (traced, profiles/),
        // and as one encoder wave lands on every CU every decoder workgroup has a slowed wave; with s_setprio 3 in the encoder it keeps
        // its speed and the decoder runs 1.8x longer (27 ms per step: no gain).  Disjoint CU masks (AUKIT_DFPWM_ENC_CUS = 32 / 48 / 64;
        // the masks do what they say: tools/micro/cumask_probe.hip) lose outright — 41 / 39 / 34 ms per step: 256 encoder waves packed
        // on few CUs run several times slower than one per CU, whatever their store width.
        if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); ctx->aux_stream = nullptr; }
        if (ctx->dec_stream) { (void)hipStreamSynchronize(ctx->dec_stream); (void)hipStreamDestroy(ctx->dec_stream); ctx->dec_stream = nullptr; }
        if (enc_cus > 0) {
            const int words = (ctx->num_cus + 31) / 32;
            std::vector<uint32_t> m_enc(words, 0), m_dec(words, 0);
            for (int cu = 0; cu < ctx->num_cus; cu++) (cu < enc_cus ? m_enc : m_dec)[cu >> 5] |= 1u << (cu & 31);
            AUKIT_HIP_CHECK(hipExtStreamCreateWithCUMask(&ctx->aux_stream, (uint32_t)words, m_enc.data()));
            AUKIT_HIP_CHECK(hipExtStreamCreateWithCUMask(&ctx->dec_stream, (uint32_t)words, m_dec.data()));
        } else {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            AUKIT_HIP_CHECK(hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, hi));
        }
        for (int i = 0; i < 10; i++) if (!ctx->aux_ev[i]) AUKIT_HIP_CHECK(hipEventCreateWithFlags(&ctx->aux_ev[i], hipEventDisableTiming));
        ctx->aux_enc_cus = enc_cus;
    }
    hipStream_t const user_stream = ctx->stream;
    struct Restore { aukit_ctx *c; hipStream_t s; ~Restore() { c->stream = s; } } restore{ctx, user_stream};
    if (ctx->dec_stream) {  // the decoder's launches go to the masked stream: ctx->stream is swapped for the duration of this call
        AUKIT_HIP_CHECK(hipEventRecord(ctx->aux_ev[9], user_stream));
        AUKIT_HIP_CHECK(hipStreamWaitEvent(ctx->dec_stream, ctx->aux_ev[9], 0));
        ctx->stream = ctx->dec_stream;
    }
    const uint32_t n = in->n;
    uint64_t max_out = 0;
    for (uint32_t s = 0; s < n; s++) max_out = std::max<uint64_t>(max_out, h_ooff[s + 1] - h_ooff[s]);
    const u64 sstride = round_up(max_out + 16, 16);  // the staging row holds whole 8-byte rounds (+ slack for the compaction's dword reads)
    int rc = ctx->enc_state_buf.ensure((size_t)n * 24 + 64 + (size_t)n * sstride);
    if (rc) return rc;
    int *state = reinterpret_cast<int *>(ctx->enc_state_buf.p);
    unsigned char *stage = reinterpret_cast<unsigned char *>(ctx->enc_state_buf.p) + round_up((size_t)n * 24 + 16, 64);
    DfSliceHook hook;
    hook.slices = std::min(slices, 8);
    // (the table of mono offsets `d_moff` was uploaded on the caller's stream: ordered by the event above)
    hook.after_slice = [&](unsigned k, unsigned nsl, u64 fed_lo, u64 fed_hi) -> int {
        AUKIT_HIP_CHECK(hipEventRecord(ctx->aux_ev[k], ctx->stream));
        AUKIT_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_ev[k], 0));
        hipLaunchKernelGGL(k_dfpwm_encode_i8_slice, dim3((n + 63) / 64), dim3(64), 0, ctx->aux_stream, mono, d_moff, d_mcount, n, stage, sstride, state, 4 * fed_lo, 4 * fed_hi,
                           k == 0 ? 1 : 0, k + 1 == nsl ? 1 : 0);  // stereo: four mono samples per fed byte
        AUKIT_HIP_CHECK(hipGetLastError());
        if (k + 1 == nsl) {
            hipLaunchKernelGGL(k_dfpwm_compact, dim3((unsigned)std::min<u64>((max_out / 4 + 256) / 256, 4), n), dim3(256), 0, ctx->aux_stream, stage, sstride, out, d_ooff, n);
            AUKIT_HIP_CHECK(hipGetLastError());
            AUKIT_HIP_CHECK(hipEventRecord(ctx->aux_ev[8], ctx->aux_stream));
            AUKIT_HIP_CHECK(hipStreamWaitEvent(user_stream, ctx->aux_ev[8], 0));  // the encoder's last slice ends the call (it waited for the decoder's)
        }
        return AUKIT_OK;
    };
    int prc = AUKIT_OK;
    const bool ran = dfpwm_decode_parallel(ctx, in, 1, 2, mono, d_moff, nullptr, &prc, 6000, 0, &hook);
    if (ctx->dec_stream) {  // whatever was enqueued on the masked stream is ordered before the caller's next work
        AUKIT_HIP_CHECK(hipEventRecord(ctx->aux_ev[9], ctx->dec_stream));
        AUKIT_HIP_CHECK(hipStreamWaitEvent(user_stream, ctx->aux_ev[9], 0));
    }
    if (!ran) return AUKIT_OK;  // not taken: the caller runs the plain sequence
    *taken = true;
    return prc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The transcode as ONE persistent launch (config 4 when the batch has at most one group of 64 streams per CU).
// The time slices above cost max(decoder, encoder) per slice: every CU hosts one encoder wave, that wave's SIMD carries the encoder's
// work on top of its share of the decoder's, and a slice ends when its slowest SIMD does.  Here one workgroup per CU (448 threads =
// 7 waves: the dispatcher deals them round-robin over the four SIMDs, so one wave sits alone on its SIMD — found at run time from
// HW_ID) runs until the batch is done:
//   * the six waves that share three SIMDs are decoders: they take units (chunk index c, group of 64 streams j) off a ticket counter
//     in time order, decode them exactly as k_df_chunks does, and publish flag[c][j] behind a release fence;
//   * the lone wave is the ENCODER of the workgroup's own group of 64 streams: it waits for flag[c][j] (acquire), checks every lane's
//     recorded start state against the true end state before it (k_df_verify's job, one chunk at a time: a mismatching lane decodes
//     its chunk again serially from the true state before it encodes it), and encodes the chunk's mono samples, state in registers.
//     While its next chunk is not there yet it takes decoder tickets itself, so its SIMD never idles.
// Nothing waits for a wave that is not running: decoders wait for nobody, an encoder only for units that running waves have taken
// or will take.  Bit-identical bytes, written straight into the packed result (8 bytes per round: no staging rows, no compaction pass).
struct DfFusedParams {
    DfParParams P;        // src, off, fed, feed, n, bpc, nchunk, W, s_start, out = mono samples, out_off, stats
    int *fst;             // [nchunk][10][npad]: start state (5 ints) and end state (5 ints) of every chunk lane, stream index fastest
    unsigned *flags;      // [nchunk][G] unit done;  flags[nchunk * G] = the ticket counter, [nchunk * G + 1] = workgroups whose encoder sat alone
    unsigned G, npad, total;
    unsigned dbg;         // AUKIT_DFPWM_FUSED_DBG, A/B only: 1 = the encoder waves never take decoder tickets (same bytes)
    const u64 *mcount;    // mono samples per stream
    unsigned char *enc_out;  // the packed result
    const u64 *ooff;         // [n + 1] byte offset of every stream's bytes in it
};

AUKIT_DEV void dff_decode_unit(const DfFusedParams &F, unsigned c, unsigned j, const signed char *lut, unsigned lane) {
    const DfParParams &P = F.P;
    const unsigned s = j * 64 + lane;
    if (s < P.n) {
        const unsigned char *p = P.src + P.off[s];
        const u64 fed = P.fed[s];
        const u64 f0 = dfp_chunk_start(P, c), e1 = dfp_chunk_start(P, c + 1), f1 = e1 < fed ? e1 : fed;
        int *st = F.fst + (size_t)c * 10 * F.npad + s;
        if (f0 >= fed && c > 0) st[(size_t)F.npad] = -1;  // strength -1: no such chunk
        else {
            DfOut O = dfp_out(P, s, lut);
            O.mode = 1; O.C = 2;  // (constants here: the other output modes of dfp_run drop out of this kernel)
            DfDec d{};
            if (c > 0) {
                const u64 fw = f0 - P.W;
                d.p.strength = P.s_start[(size_t)s * (P.nblk + 1) + c];
                d.p.pb = fw ? (int)((p[dfp_src_index(fw - 1, P.feed)] >> 6) & 2) - 1 : -1;
                dfp_run<false>(p, fw, f0, d, O);
            }
            int v[6];
            dfp_pack(d, v);
#pragma unroll
            for (int i = 0; i < 5; i++) st[(size_t)i * F.npad] = v[i];
            dfp_run<true>(p, f0, f1, d, O);
            dfp_pack(d, v);
#pragma unroll
            for (int i = 0; i < 5; i++) st[(size_t)(5 + i) * F.npad] = v[i];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the wave's stores are visible chip-wide (other XCDs' L2 included) before the flag is
    if (lane == 0) __hip_atomic_store(&F.flags[(size_t)c * F.G + j], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// one ticket, wave-uniform
AUKIT_DEV unsigned dff_take(const DfFusedParams &F, unsigned lane) {
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(&F.flags[(size_t)F.P.nchunk * F.G], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
}

AUKIT_DEV void dff_encoder(const DfFusedParams &F, unsigned j, const signed char *lut, unsigned lane) {
    const DfParParams &P = F.P;
    const unsigned s = j * 64 + lane;
    const bool act = s < P.n;
    const u64 L = act ? F.mcount[s] : 0, fed = act ? P.fed[s] : 0;
    const unsigned char *src = act ? P.src + P.off[s] : P.src;
    const signed char *p = P.out + (act ? P.out_off[s] : 0);  // this stream's mono samples (16-byte aligned: host)
    unsigned char *o = F.enc_out + (act ? F.ooff[s] : 0);  // rows are 60 010 bytes apart: 2-byte aligned, the stores below say so
    DfEnc e{};
    u64 i = 0, w = 0;
    int truth[6] = {0, 0, 0, 0, 0, 0};
    unsigned redone = 0, chunks = 0;
    bool tickets = !(F.dbg & 1);
    // one round: 64 samples → 8 bytes
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2u __attribute__((ext_vector_type(2), aligned(1)));  // (unaligned global stores are single instructions on gfx950)
    auto round_bits = [&](const u32x4 &q0, const u32x4 &q1, const u32x4 &q2, const u32x4 &q3) -> uint2 {
        unsigned ob[2] = {0, 0};
        const u32x4 q[4] = {q0, q1, q2, q3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned words[4] = {q[k].x ^ 0x80808080u, q[k].y ^ 0x80808080u, q[k].z ^ 0x80808080u, q[k].w ^ 0x80808080u};  // u = v + 128
#pragma unroll
            for (int h = 0; h < 2; h++) {
                unsigned byte = 0;
#pragma unroll
                for (int b = 0; b < 8; b++) byte |= df_encode_u(e, (words[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFF) & (1u << b);
                ob[k >> 1] |= (byte & 255u) << (8 * (2 * (k & 1) + h));
            }
        }
        i += 64;
        return make_uint2(ob[0], ob[1]);
    };
    unsigned long long t_wait = 0, t_enc = 0, tt = wall_clock64();
    unsigned c = 0;
    while (c < P.nchunk) {
        // wait for the next unit (decoding others meanwhile), then take every further unit that is already there: one span
        const unsigned *fl = &F.flags[(size_t)c * F.G + j];
        unsigned ready = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (!ready) {
            if (tickets) {
                const unsigned t = dff_take(F, lane);
                if (t < F.total) dff_decode_unit(F, t / F.G, t % F.G, lut, lane);
                else tickets = false;
            } else __builtin_amdgcn_s_sleep(8);
            ready = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned c1 = c + 1;
        for (unsigned k = 1; k < 8; k++) {  // (bounded, single exit)
            const bool more = c1 == c + k && c1 < P.nchunk && __hip_atomic_load(fl + (size_t)k * F.G * (c + k < P.nchunk ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (more) c1++;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        { const unsigned long long now = wall_clock64(); t_wait += now - tt; tt = now; }
        for (unsigned cc = c; cc < c1; cc++) {
            const u64 f0 = dfp_chunk_start(P, cc);
            if (act && (f0 < fed || cc == 0)) {
                const int *st = F.fst + (size_t)cc * 10 * F.npad + s;
                chunks++;
                bool same = true;
                if (cc > 0) {
#pragma unroll
                    for (int k = 0; k < 5; k++) same = same && st[(size_t)k * F.npad] == truth[k];
                }
                if (same) {
#pragma unroll
                    for (int k = 0; k < 5; k++) truth[k] = st[(size_t)(5 + k) * F.npad];
                } else {  // (rare on signal: the warm-up converges) decode the chunk again from the true state
                    DfDec d;
                    dfp_unpack(truth, d);
                    const u64 e1 = dfp_chunk_start(P, cc + 1), f1 = e1 < fed ? e1 : fed;
                    DfOut O = dfp_out(P, s, lut);
                    O.mode = 1; O.C = 2;
                    dfp_run<true, true>(src, f0, f1, d, O);   // (digital silence does not converge: DfRepeat)
                    dfp_pack(d, truth);
                    redone++;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // the samples just stored are read back below
                }
            }
        }
        // encode the samples these chunks made final (whole rounds of 64; the last chunk takes every stream to its end).  Under the
        // decoders' traffic a load takes longer than a round (~1.4 us of dependent instructions): four rounds are kept in flight.
        const bool last = c1 == P.nchunk;
        const u64 lim = 4 * dfp_chunk_start(P, c1);
        const u64 hi = last ? L : (lim < L ? lim : L & ~63ull);
        // Four rounds (256 samples, ~6 us of dependent instructions) per turn.  Beside six decoder waves a load or a store of this wave
        // takes several microseconds to come back, and the only wait hipcc's own scheduling leaves in a loop like this is a wait for
        // everything; so the schedule is made by hand: the loads of the NEXT turn first (inline asm: they stay where they are;
        // their registers pass through the wait as operands, so nothing reads or reuses them earlier), then this turn's arithmetic,
        // then ONE wait — for those loads and for the stores of the turn before, both a whole turn old — and this turn's four stores.
        constexpr int R = AUKIT_DFF_R;  // rounds per turn
        if (i + 64 * R <= hi) {
            u32x4 B[4 * R];
#define AUKIT_DFF_ISSUE8(b, ptr, o0, o1, o2, o3, o4, o5, o6, o7)                                                                           \
            asm volatile("global_load_dwordx4 %0, %8, off offset:" #o0 "\n\tglobal_load_dwordx4 %1, %8, off offset:" #o1 "\n\t"                \
                         "global_load_dwordx4 %2, %8, off offset:" #o2 "\n\tglobal_load_dwordx4 %3, %8, off offset:" #o3 "\n\t"                \
                         "global_load_dwordx4 %4, %8, off offset:" #o4 "\n\tglobal_load_dwordx4 %5, %8, off offset:" #o5 "\n\t"                \
                         "global_load_dwordx4 %6, %8, off offset:" #o6 "\n\tglobal_load_dwordx4 %7, %8, off offset:" #o7                       \
                         : "=&v"(B[b]), "=&v"(B[b + 1]), "=&v"(B[b + 2]), "=&v"(B[b + 3]), "=&v"(B[b + 4]), "=&v"(B[b + 5]), "=&v"(B[b + 6]), "=&v"(B[b + 7]) \
                         : "v"(ptr) : "memory")
#define AUKIT_DFF_TIE8(b, text)                                                                                                         \
            asm volatile(text : "+v"(B[b]), "+v"(B[b + 1]), "+v"(B[b + 2]), "+v"(B[b + 3]), "+v"(B[b + 4]), "+v"(B[b + 5]), "+v"(B[b + 6]), "+v"(B[b + 7]) : : "memory")
#if AUKIT_DFF_R == 4
#define AUKIT_DFF_ISSUE(ptr) do { AUKIT_DFF_ISSUE8(0, ptr, 0, 16, 32, 48, 64, 80, 96, 112); AUKIT_DFF_ISSUE8(8, ptr, 128, 144, 160, 176, 192, 208, 224, 240); } while (0)
#define AUKIT_DFF_WAIT() do { AUKIT_DFF_TIE8(0, "s_waitcnt vmcnt(0)"); AUKIT_DFF_TIE8(8, ""); } while (0)
#else
#define AUKIT_DFF_ISSUE(ptr) AUKIT_DFF_ISSUE8(0, ptr, 0, 16, 32, 48, 64, 80, 96, 112)
#define AUKIT_DFF_WAIT() AUKIT_DFF_TIE8(0, "s_waitcnt vmcnt(0)")
#endif
            u32x4 A[4 * R];
            {
                const signed char *pn = p + i;
                AUKIT_DFF_ISSUE(pn);
                AUKIT_DFF_WAIT();
#pragma unroll
                for (int k = 0; k < 4 * R; k++) A[k] = B[k];
            }
            while (i + 64 * R <= hi) {
                // never past what is final: a lane with nothing further reads its current samples again
                const signed char *pn = p + (i + 128 * R <= hi ? i + 64 * R : i);
                AUKIT_DFF_ISSUE(pn);
                __builtin_amdgcn_sched_barrier(0);
                uint2 ob[R];
#pragma unroll
                for (int r = 0; r < R; r++) ob[r] = round_bits(A[4 * r], A[4 * r + 1], A[4 * r + 2], A[4 * r + 3]);
                __builtin_amdgcn_sched_barrier(0);
                AUKIT_DFF_WAIT();  // nothing asynchronous is alive across the loop's back edge (hipcc copies loop-carried values around there)
#pragma unroll
                for (int r = 0; r < R; r++) { u32x2u v; v.x = ob[r].x; v.y = ob[r].y; *reinterpret_cast<u32x2u *>(o + w + 8 * r) = v; }  // waited for a turn later
                w += 8 * R;
#pragma unroll
                for (int k = 0; k < 4 * R; k++) A[k] = B[k];
            }
#undef AUKIT_DFF_ISSUE
#undef AUKIT_DFF_ISSUE8
#undef AUKIT_DFF_TIE8
#undef AUKIT_DFF_WAIT
        }
        while (i + 64 <= hi) {  // what is left of a span that is not a multiple of 256 samples (stream ends, odd chunk sizes)
            const u32x4 *q = reinterpret_cast<const u32x4 *>(p + i);
            const u32x4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const uint2 ob = round_bits(q0, q1, q2, q3);
            u32x2u v; v.x = ob.x; v.y = ob.y;
            *reinterpret_cast<u32x2u *>(o + w) = v;
            w += 8;
        }
        if (last && i < L) {  // the last < 64 samples, the last byte padded with samples of value 0
            for (; i < L; i += 8) {
                unsigned byte = 0;
                for (int b = 0; b < 8; b++) {
                    const unsigned u = i + b < L ? (unsigned)((int)p[i + b] + 128) : 128u;
                    byte |= df_encode_u(e, u) & (1u << b);
                }
                o[w++] = (unsigned char)byte;
            }
        }
        c = c1;
        { const unsigned long long now = wall_clock64(); t_enc += now - tt; tt = now; }
    }
    if (lane == 0) { atomicAdd(&F.flags[(size_t)P.nchunk * F.G + 3], (unsigned)(t_wait >> 4)); atomicAdd(&F.flags[(size_t)P.nchunk * F.G + 4], (unsigned)(t_enc >> 4)); }
    if (redone) atomicAdd(&P.stats[0], redone);
    if (chunks) atomicAdd(&P.stats[1], chunks);
}

__global__ __launch_bounds__(448) void k_df_fused(const DfFusedParams F) {
    extern __shared__ signed char lut[];  // 64 KiB mix table (+ padding so that a CU holds one workgroup)
    __shared__ unsigned simd_of[8];
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) lut[i] = (signed char)dfp_mix((i >> 8) - 128, (i & 255) - 128);
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0) simd_of[wave] = (hw >> 4) & 3u;
    __syncthreads();
    // the encoder: the wave that has its SIMD to itself (wave 3 when the seven waves were dealt round-robin)
    unsigned enc = 3, alone = 0;
    for (unsigned a = 0; a < 7; a++) {
        unsigned others = 0;
        for (unsigned b = 0; b < 7; b++) others += (b != a && simd_of[b] == simd_of[a]) ? 1u : 0u;
        if (others == 0 && !alone) { enc = a; alone = 1; }
    }
    if (threadIdx.x == 0 && alone) atomicAdd(&F.flags[(size_t)F.P.nchunk * F.G + 1], 1u);
    if (wave == enc && blockIdx.x < F.G) {
        const unsigned long long t0 = wall_clock64();
        dff_encoder(F, blockIdx.x, lut, lane);
        if (lane == 0) atomicMax(&F.flags[(size_t)F.P.nchunk * F.G + 2], (unsigned)(wall_clock64() - t0));  // (AUKIT_DFPWM_STATS: the slowest encoder, 100 MHz ticks)
        return;
    }
    const unsigned long long td0 = wall_clock64();
    unsigned t = dff_take(F, lane);
    while (t < F.total) {
        dff_decode_unit(F, t / F.G, t % F.G, lut, lane);
        t = dff_take(F, lane);
    }
    if (lane == 0) {  // (AUKIT_DFPWM_STATS) the slowest and the mean decoder wave
        atomicMax(&F.flags[(size_t)F.P.nchunk * F.G + 5], (unsigned)(wall_clock64() - td0));
        atomicAdd(&F.flags[(size_t)F.P.nchunk * F.G + 6], (unsigned)((wall_clock64() - td0) >> 8));
    }
}

// host side of the fused transcode; *taken = false (nothing launched) when the batch does not fit its shape
int dfpwm_transcode_fused(aukit_ctx *ctx, const aukit_batch *in, signed char *mono, const u64 *d_moff, const u64 *d_mcount, unsigned char *out, const u64 *d_ooff,
                          const uint64_t *h_ooff, bool *taken) {
    *taken = false;
    const uint32_t n = in->n;
    if (n == 0 || getenv("AUKIT_DFPWM_SERIAL")) return AUKIT_OK;
    // more than one 64-stream group per CU: sub-batches of whole groups, one launch after the other (every CU hosts one encoder wave)
    unsigned cap = (unsigned)ctx->num_cus;
    if (const char *e = getenv("AUKIT_DFPWM_FUSED_GROUPS")) cap = (unsigned)std::max(1, std::min(atoi(e), ctx->num_cus));  // (tests: sub-batches of a small batch)
    const unsigned G_all = (n + 63) / 64, nsub = (G_all + cap - 1) / cap, G_sub = (G_all + nsub - 1) / nsub;
    const uint32_t n_sub = std::min<uint32_t>(n, G_sub * 64u);
    // a launch takes the encoder's serial 13 ms however few streams it has: batches that leave the last launches thinly populated
    // (20 000 streams = 313 groups: two launches at 61 %) are left to the time-sliced version, which scales with the stream count
    if (nsub > 1 && (double)G_all < 0.75 * (double)nsub * cap && !getenv("AUKIT_DFPWM_FUSED")) return AUKIT_OK;
    std::vector<uint64_t> h_off(in->off.begin(), in->off.begin() + n), h_fed(n);
    uint64_t fed_max = 0;
    for (uint32_t s = 0; s < n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        h_fed[s] = nb ? nb + (nb + 6000 - 1) / 6000 - 1 : 0;  // 6001-byte slices advanced by 6000 (Q10)
        fed_max = std::max(fed_max, h_fed[s]);
    }
    (void)h_ooff;
    uint64_t W = 256;
    if (const char *e = getenv("AUKIT_DFPWM_BLOCK")) W = std::max<uint64_t>(16, strtoull(e, nullptr, 10) & ~15ull);  // 4 W mono samples = whole encoder rounds
    const unsigned nblk = (unsigned)((fed_max + W - 1) / W);
    // units per decoder wave: enough of them that the last ones end close together, long enough that the warm-up stays 1/8 of a chunk
    unsigned want = (unsigned)std::max<uint64_t>(1, (uint64_t)ctx->num_cus * 384 * 10 / n_sub);
    if (const char *e = getenv("AUKIT_DFPWM_CHUNKS")) want = (unsigned)std::max(1, atoi(e));
    const unsigned bpc = std::max<unsigned>(nblk ? (nblk + want - 1) / want : 1, getenv("AUKIT_DFPWM_CHUNKS") ? 1u : 8u);
    const unsigned nchunk = nblk ? (nblk + bpc - 1) / bpc : 0;
    if (nchunk < 2) return AUKIT_OK;
    const unsigned npad = (unsigned)round_up(n_sub, 64);
    const size_t b_tab = (size_t)n * 16, b_maps = (size_t)n_sub * nchunk * sizeof(SatMap), b_ss = (size_t)n_sub * (nchunk + 1) * 4, b_st = (size_t)nchunk * 10 * npad * 4,
                 b_fl = round_up(((size_t)nchunk * G_sub + 8) * 4, 64);
    int rc = ctx->tmp_buf2.ensure(b_tab + b_maps + b_ss + b_st + b_fl + 256);
    if (rc) return rc;
    char *B = reinterpret_cast<char *>(ctx->tmp_buf2.p);
    if ((rc = h2d_table(ctx, B, h_off.data(), (size_t)n * 8)) || (rc = h2d_table(ctx, B + (size_t)n * 8, h_fed.data(), (size_t)n * 8))) return rc;
    const unsigned lds = 65536 + 20480;  // more than half of a CU's 160 KiB: one workgroup per CU
    if (!ctx->fused_attr_set) { AUKIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_df_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ctx->fused_attr_set = true; }
    for (uint32_t s0 = 0; s0 < n; s0 += n_sub) {
        const uint32_t ns = std::min<uint32_t>(n_sub, n - s0);
        const unsigned G = (ns + 63) / 64;
        DfFusedParams F{};
        DfParParams &P = F.P;
        P.src = in->data(); P.off = reinterpret_cast<const u64 *>(B) + s0; P.fed = reinterpret_cast<const u64 *>(B) + n + s0; P.feed = Feed{6001, 6000};
        P.n = ns; P.nblk = nchunk; P.bpc = bpc; P.nchunk = nchunk; P.W = W;
        P.maps = reinterpret_cast<SatMap *>(B + b_tab); P.s_start = reinterpret_cast<int *>(B + b_tab + b_maps);
        F.fst = reinterpret_cast<int *>(B + b_tab + b_maps + b_ss);
        F.flags = reinterpret_cast<unsigned *>(B + b_tab + b_maps + b_ss + b_st);
        P.stats = reinterpret_cast<unsigned *>(B + b_tab + b_maps + b_ss + b_st + b_fl);
        P.mode = 1; P.C = 2; P.out = mono; P.out_off = d_moff + s0; P.out_stride = nullptr; P.lead = 0;
        if (const char *e = getenv("AUKIT_DFPWM_FUSED_DBG")) F.dbg = (unsigned)atoi(e);
        F.G = G; F.npad = npad; F.total = nchunk * G; F.mcount = d_mcount + s0;
        F.enc_out = out; F.ooff = d_ooff + s0;
        if (hipMemsetAsync(F.flags, 0, b_fl + 8, ctx->stream) != hipSuccess) return fail(AUKIT_E_HIP, "hipMemsetAsync failed");
        hipLaunchKernelGGL((k_df_blockmaps<4, false>), dim3((unsigned)(((size_t)ns * nchunk + 255) / 256)), dim3(256), 0, ctx->stream, P);
        hipLaunchKernelGGL(k_df_blockscan, dim3((ns + 63) / 64), dim3(64), 0, ctx->stream, P);
        hipLaunchKernelGGL(k_df_fused, dim3((unsigned)std::max<int>(ctx->num_cus, (int)G)), dim3(448), lds, ctx->stream, F);
        AUKIT_HIP_CHECK(hipGetLastError());
        if (getenv("AUKIT_DFPWM_STATS") || ctx->collect_stats) {
            unsigned h[2] = {0, 0}, al[6] = {0, 0, 0, 0, 0, 0};
            (void)hipMemcpyAsync(h, P.stats, 8, hipMemcpyDeviceToHost, ctx->stream);
            (void)hipMemcpyAsync(al, F.flags + (size_t)nchunk * G + 1, 24, hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            if (s0 == 0) { ctx->counters[AUKIT_COUNTER_DFPWM_CHUNKS] = 0; ctx->counters[AUKIT_COUNTER_DFPWM_CHUNKS_REDONE] = 0; }
            ctx->counters[AUKIT_COUNTER_DFPWM_CHUNKS] += h[1]; ctx->counters[AUKIT_COUNTER_DFPWM_CHUNKS_REDONE] += h[0];
            if (getenv("AUKIT_DFPWM_STATS")) fprintf(stderr, "[dfpwm fused] %u streams x %u chunks of %u blocks of %llu fed bytes: %u of %u chunks redone by their encoder lane; encoder alone on its SIMD in %u of %d workgroups, slowest encoder %.2f ms (mean over encoders: waiting %.2f ms, verify + encode %.2f ms); decoder waves: slowest %.2f ms, mean %.2f ms\n",
                    ns, nchunk, bpc, (unsigned long long)W, h[0], h[1], al[0], std::max<int>(ctx->num_cus, (int)G), al[1] * 1e-5, al[2] * 16e-5 / G, al[3] * 16e-5 / G, al[4] * 1e-5, al[5] * 256e-5 / (6.0 * std::max<int>(ctx->num_cus, (int)G)));
        }
    }
    *taken = true;
    return AUKIT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Exact parallel ENCODING for small batches (one file is the usual batch of auconvert: 12 ms of serial encoder for ten seconds).
// The encoder's bits depend on its state, so nothing is scannable — but its state space collapses: started 2048 samples early from
// every (strength, previous bit) pair and an arbitrary charge, the 2032 candidate encoders end in a handful of distinct states
// (4-11 on synthetic audio, CPU experiment), and the true state is one of them.  So, per chunk of ~7500 samples:
//   1. k_dfe_cand      2032 lanes warm up over the 2048 samples before the chunk                      → candidate start states
//   2. k_dfe_distinct  the distinct ones (LDS hash set, at most 64)
//   3. k_dfe_ends      one lane per distinct start state runs the chunk                               → its end state
//   4. k_dfe_resolve   one lane per stream walks its chunks from the reset state: the true start of a chunk is the end of the one
//                      before it; looked up among the chunk's candidates — or, if it is not there, the chunk is run on the spot
//   5. k_dfe_emit      one lane per chunk encodes it from its true start state
// Exact by construction (step 4 only ever follows true states); the candidates decide the speed, not the bytes.  Work grows by
// ~500x per stream, so this is for batches of a few streams; larger ones keep one lane per stream.
struct DfeChunk { u64 in_off, out_off; unsigned count, stream, first, last; };
constexpr unsigned DFE_W = 2048, DFE_NC = 2032, DFE_INIT = 128u;  // warm-up samples; candidates = 1016 strengths x 2; reset state packed
AUKIT_DEV unsigned dfe_pack(const DfEnc &e) { return (unsigned)e.cu | (unsigned)e.strength << 8 | (e.pb > 0 ? 1u << 18 : 0u); }
AUKIT_DEV DfEnc dfe_unpack(unsigned v) { DfEnc e; e.cu = (int)(v & 255u); e.strength = (int)((v >> 8) & 1023u); e.pb = (v >> 18) & 1u ? 1 : -1; return e; }

// runs `count` samples of one chunk; EMIT: writes count / 8 bytes (+ one padded with samples of value 0 when `pad`)
template <bool EMIT>
AUKIT_DEV void dfe_run(const signed char *p, unsigned count, DfEnc &e, unsigned char *o, bool pad) {
    unsigned i = 0;
    uint4 cur = count ? *reinterpret_cast<const uint4 *>(p) : make_uint4(0, 0, 0, 0);  // chunk starts are 16-byte aligned, rows are padded
    for (; i + 16 <= count; i += 16) {
        uint4 nxt = cur;
        if (i + 16 < count) nxt = *reinterpret_cast<const uint4 *>(p + i + 16);
        const unsigned w[4] = {cur.x ^ 0x80808080u, cur.y ^ 0x80808080u, cur.z ^ 0x80808080u, cur.w ^ 0x80808080u};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            unsigned byte = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) byte |= df_encode_u(e, (w[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFF) & (1u << b);
            if (EMIT) o[(i >> 3) + h] = (unsigned char)byte;
        }
        cur = nxt;
    }
    if (i < count) {  // the stream's last samples
        const unsigned w[4] = {cur.x ^ 0x80808080u, cur.y ^ 0x80808080u, cur.z ^ 0x80808080u, cur.w ^ 0x80808080u};
        for (unsigned k0 = i; k0 < count; k0 += 8) {
            if (!pad && k0 + 8 > count) break;
            unsigned byte = 0;
            for (int b = 0; b < 8; b++) {
                const unsigned k = k0 + b;
                const unsigned u = k < count ? (w[(k - i) >> 2] >> (8 * ((k - i) & 3))) & 0xFF : 128u;
                byte |= df_encode_u(e, u) & (1u << b);
            }
            if (EMIT) o[k0 >> 3] = (unsigned char)byte;
        }
    }
}

__global__ __launch_bounds__(256) void k_dfe_cand(const signed char *in, const DfeChunk *chunks, unsigned *cand) {
    __shared__ unsigned char warm[DFE_W];
    const DfeChunk ch = chunks[blockIdx.y];
    const unsigned id = blockIdx.x * 256 + threadIdx.x;
    unsigned *dst = cand + (size_t)blockIdx.y * 2048 + id;
    if (ch.first) { *dst = 0xFFFFFFFFu; return; }  // the stream's first chunk starts from the reset state
    const signed char *p = in + ch.in_off - DFE_W;
    *reinterpret_cast<uint2 *>(warm + 8 * threadIdx.x) = *reinterpret_cast<const uint2 *>(p + 8 * threadIdx.x);
    __syncthreads();
    if (id >= DFE_NC) { *dst = 0xFFFFFFFFu; return; }
    DfEnc e;
    e.cu = 128; e.strength = 8 + (int)(id >> 1); e.pb = id & 1 ? 1 : -1;
    for (unsigned k = 0; k < DFE_W; k += 4) {
        const unsigned w = *reinterpret_cast<const unsigned *>(warm + k) ^ 0x80808080u;
#pragma unroll
        for (int b = 0; b < 4; b++) df_encode_u(e, (w >> (8 * b)) & 0xFF);
    }
    *dst = dfe_pack(e);
}

__global__ __launch_bounds__(256) void k_dfe_distinct(const DfeChunk *chunks, const unsigned *cand, unsigned *dist, unsigned *dcount) {
    __shared__ unsigned table[128];
    __shared__ int overflow;
    const unsigned c = blockIdx.x;
    if (threadIdx.x < 128) table[threadIdx.x] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) overflow = 0;
    __syncthreads();
    if (chunks[c].first) {
        if (threadIdx.x == 0) { dist[(size_t)c * 64] = DFE_INIT; dcount[c] = 1; }
        return;
    }
    for (unsigned i = threadIdx.x; i < 2048; i += 256) {
        const unsigned v = cand[(size_t)c * 2048 + i];
        if (v == 0xFFFFFFFFu) continue;
        unsigned h = (v * 2654435761u) >> 25;
        int probes = 0;
        for (;; h = (h + 1) & 127) {
            const unsigned old = atomicCAS(&table[h], 0xFFFFFFFFu, v);
            if (old == 0xFFFFFFFFu || old == v) break;
            if (++probes > 127) { overflow = 1; break; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned k = 0;
        for (int h = 0; h < 128; h++)
            if (table[h] != 0xFFFFFFFFu) { if (k < 64) dist[(size_t)c * 64 + k] = table[h]; k++; }
        dcount[c] = (k > 64 || overflow) ? 0u : k;  // too many survivors: the resolve pass runs this chunk itself
    }
}

__global__ __launch_bounds__(64) void k_dfe_ends(const signed char *in, const DfeChunk *chunks, const unsigned *dist, const unsigned *dcount, unsigned *dend) {
    const unsigned c = blockIdx.x, k = threadIdx.x;
    if (k >= dcount[c]) return;
    const DfeChunk ch = chunks[c];
    DfEnc e = dfe_unpack(dist[(size_t)c * 64 + k]);
    dfe_run<false>(in + ch.in_off, ch.count, e, nullptr, false);
    dend[(size_t)c * 64 + k] = dfe_pack(e);
}

__global__ __launch_bounds__(64) void k_dfe_resolve(const signed char *in, const DfeChunk *chunks, const unsigned *stream_first, unsigned n, const unsigned *dist,
                                                   const unsigned *dcount, const unsigned *dend, unsigned *start, unsigned *stats) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    unsigned state = DFE_INIT, missed = 0;
    for (unsigned c = stream_first[s]; c < stream_first[s + 1]; c++) {
        start[c] = state;
        unsigned found = 0xFFFFFFFFu;
        for (unsigned k = 0; k < dcount[c]; k++)
            if (dist[(size_t)c * 64 + k] == state) found = k;
        if (found != 0xFFFFFFFFu) state = dend[(size_t)c * 64 + found];
        else {  // not among the candidates: run the chunk from the true state here
            const DfeChunk ch = chunks[c];
            DfEnc e = dfe_unpack(state);
            dfe_run<false>(in + ch.in_off, ch.count, e, nullptr, false);
            state = dfe_pack(e);
            missed++;
        }
    }
    if (missed) atomicAdd(&stats[0], missed);
}

__global__ __launch_bounds__(64) void k_dfe_emit(const signed char *in, const DfeChunk *chunks, unsigned nchunks, const unsigned *start, unsigned char *out) {
    const unsigned c = blockIdx.x * 64 + threadIdx.x;
    if (c >= nchunks) return;
    const DfeChunk ch = chunks[c];
    DfEnc e = dfe_unpack(start[c]);
    dfe_run<true>(in + ch.in_off, ch.count, e, out + ch.out_off, ch.last != 0);
}

// host: returns true when the small-batch path took the encode (*rc = its status).  h_* are host copies of the device tables.
bool dfpwm_encode_i8_small(aukit_ctx *ctx, const signed char *in, const uint64_t *h_in_off, const uint64_t *h_count, uint32_t n, unsigned char *out, const uint64_t *h_ooff,
                           int *rc) {
    // measured on the transcode of random bytes (a noise-like mix: the worst case, many starts are not among the candidates):
    // 1 stream 2.5 vs 15.1 ms, 16 streams 8.7 vs 15.0 ms, 64 streams 14.6 vs 13.1 ms — so up to 16 streams
    if (n == 0 || n > 16 || getenv("AUKIT_DFPWM_SERIAL") || getenv("AUKIT_DFPWM_ENC_SERIAL")) return false;
    uint64_t maxc = 0;
    for (uint32_t s = 0; s < n; s++) maxc = std::max(maxc, h_count[s]);
    if (maxc < 65536 || maxc > 0xF0000000ull) return false;
    std::vector<DfeChunk> chunks;
    std::vector<unsigned> sfirst(n + 1, 0);
    unsigned want = 64;  // chunks per stream
    if (const char *e = getenv("AUKIT_DFPWM_ENC_CHUNKS")) want = (unsigned)std::max(1, atoi(e));
    for (uint32_t s = 0; s < n; s++) {
        sfirst[s] = (unsigned)chunks.size();
        const uint64_t N = h_count[s];
        uint64_t L = std::max<uint64_t>(4096, ((N + want - 1) / want + 63) & ~63ull);  // a multiple of 64 samples, at least 2 warm-ups long
        for (uint64_t p = 0; p < N || p == 0; p += L) {
            DfeChunk c;
            c.in_off = h_in_off[s] + p; c.out_off = h_ooff[s] + p / 8; c.count = (unsigned)std::min<uint64_t>(L, N - p); c.stream = s;
            c.first = p == 0 ? 1u : 0u; c.last = p + L >= N ? 1u : 0u;
            chunks.push_back(c);
            if (N == 0) break;
        }
    }
    sfirst[n] = (unsigned)chunks.size();
    const size_t nch = chunks.size();
    const size_t b_chunks = nch * sizeof(DfeChunk), b_sf = ((size_t)n + 1) * 4, b_cand = nch * 2048 * 4, b_dist = nch * 64 * 4, b_cnt = nch * 4;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_chunks = take(b_chunks), o_sf = take(b_sf), o_cand = take(b_cand), o_dist = take(b_dist), o_dcount = take(b_cnt), o_dend = take(b_dist), o_start = take(b_cnt), o_stats = take(16);
    if ((*rc = ctx->tmp_buf3.ensure(o + 64))) return true;
    char *B = reinterpret_cast<char *>(ctx->tmp_buf3.p);
    if (h2d_table(ctx, B + o_chunks, chunks.data(), b_chunks) || h2d_table(ctx, B + o_sf, sfirst.data(), b_sf) ||
        hipMemsetAsync(B + o_stats, 0, 16, ctx->stream) != hipSuccess) {
        *rc = fail(AUKIT_E_HIP, "upload of the DFPWM encoder chunk table failed");
        return true;
    }
    const DfeChunk *dch = reinterpret_cast<const DfeChunk *>(B + o_chunks);
    unsigned *cand = reinterpret_cast<unsigned *>(B + o_cand), *dist = reinterpret_cast<unsigned *>(B + o_dist), *dcount = reinterpret_cast<unsigned *>(B + o_dcount);
    unsigned *dend = reinterpret_cast<unsigned *>(B + o_dend), *start = reinterpret_cast<unsigned *>(B + o_start), *stats = reinterpret_cast<unsigned *>(B + o_stats);
    hipLaunchKernelGGL(k_dfe_cand, dim3(8, (unsigned)nch), dim3(256), 0, ctx->stream, in, dch, cand);
    hipLaunchKernelGGL(k_dfe_distinct, dim3((unsigned)nch), dim3(256), 0, ctx->stream, dch, cand, dist, dcount);
    hipLaunchKernelGGL(k_dfe_ends, dim3((unsigned)nch), dim3(64), 0, ctx->stream, in, dch, dist, dcount, dend);
    hipLaunchKernelGGL(k_dfe_resolve, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, in, dch, reinterpret_cast<const unsigned *>(B + o_sf), n, dist, dcount, dend, start, stats);
    hipLaunchKernelGGL(k_dfe_emit, dim3((unsigned)((nch + 63) / 64)), dim3(64), 0, ctx->stream, in, dch, (unsigned)nch, start, out);
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "parallel DFPWM encode launch failed"); return true; }
    if (getenv("AUKIT_DFPWM_STATS")) {
        unsigned h[2] = {0, 0};
        (void)hipMemcpyAsync(h, stats, 8, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[dfpwm] encoder: %u streams in %zu chunks, %u chunk starts not among the candidates (run serially)\n", n, nch, h[0]);
    }
    *rc = AUKIT_OK;
    return true;
}

int dfpwm_encode_i8(aukit_ctx *ctx, const signed char *in, const u64 *d_in_off, const u64 *d_count, uint32_t n, unsigned char *out, const u64 *d_ooff) {
    hipLaunchKernelGGL(k_dfpwm_encode_i8, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, in, d_in_off, d_count, n, out, d_ooff);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

}  // namespace aukit
