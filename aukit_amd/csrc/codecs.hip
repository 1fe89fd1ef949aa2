// codecs.hip — sequential codecs on gfx950: DFPWM1a (decode, encode, fused stereo→mono transcode) and IMA ADPCM.
//
// DFPWM (cc.audio.dfpwm, restated — parity unpinned): 1 bit per sample; `strength` is a
//   saturating counter but `charge` has a floor-division + nudge, so the recurrence is not associative.  It is
//   also only 0.125 B/sample of input: latency/ALU-bound, never HBM-bound.  One lane per stream, state in VGPRs.
// IMA ADPCM (aukit.lua:1241-1273, :2797-2815): both recurrences are clamp(x + d, lo, hi) maps, which compose:
//   (a1,lo1,hi1) then (a2,lo2,hi2) = (a1+a2, clamp(lo1+a2,lo2,hi2), clamp(hi1+a2,lo2,hi2)).  One wave decodes one
//   (block, channel): 16 nibbles per lane, wave64 shuffle scan for the step index, again for the predictor.
//   The stream.adpcm path keeps the decoded block in LDS and resamples it in the same kernel (fp64, reference order).
#include <algorithm>
#include <type_traits>
#include <chrono>
#include <map>
#include "resample.h"
#include "resample_dev.h"
#include "dfpwm_dev.h"

namespace aukit {

int audio_from_int_rows(aukit_ctx *ctx, int src_kind, const void *rows_dev, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len,
                        uint32_t n, int channels, double rate, double new_rate, int interp, bool do_resample, int dtype, double norm_pos,
                        double norm_neg, aukit_audio **out);

// ================================================================= DFPWM1a
// aukit.dfpwm  aukit.lua:1399-1412: 6001-byte slices advanced by 6000 (Q10); output de-interleaved int8 rows
__global__ __launch_bounds__(64) void k_dfpwm_decode(const unsigned char *src, const unsigned long long *off, unsigned n, int C,
                                                    signed char *out, const unsigned long long *row_off, const unsigned long long *row_stride) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const unsigned char *p = src + off[s];
    const unsigned long long nb = off[s + 1] - off[s];
    signed char *o = out + row_off[s];
    const unsigned long long st = row_stride[s];
    DfDec d{};
    unsigned long long i = 0;  // index within a channel
    int c = 0;
    for (unsigned long long pos = 0; pos < nb; pos += 6000) {
        const unsigned long long cnt = nb - pos < 6001 ? nb - pos : 6001;
        for (unsigned long long b = 0; b < cnt; b++) {
            unsigned byte = p[pos + b];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int v = df_decode_bit(d, byte & 1);
                byte >>= 1;
                o[(unsigned long long)c * st + i] = (signed char)v;
                if (++c == C) { c = 0; i++; }
            }
        }
    }
}

// Audio:dfpwm  aukit.lua:1005-1018 + encodePCM :874, in two steps: every sample is quantised to floor(d * (d < 0 and 128 or 127)) in
// parallel (range-checked: the encoder raises outside [-128, 127]) into an int8 row in ENCODING order — interleaved, or channel
// after channel (:1011-1014) — and the serial encoder (k_dfpwm_encode_i8, dfpwm_par.hip) then runs on bytes.  With the fp64 work
// inside the serial lane this took 108 ms for ten seconds of mono audio (a lone wave pays every branch, load and fp64 conversion
// in full); now 12 ms.
AUKIT_DEV int dfpwm_q1(double dd, bool &bad) {
    const double fv = floor(dd * (dd < 0 ? 128 : 127));
    if (!(fv <= 127 && fv >= -128)) { bad = true; return 0; }  // "Amplitude at position ... should be between -128 and 127"
    return (int)fv;
}
template <typename T>
__global__ __launch_bounds__(256) void k_dfpwm_quantize(const T *in, const unsigned long long *len, const unsigned long long *roff, const unsigned long long *rstride,
                                                       int C, int interleaved, signed char *q, const unsigned long long *qoff, int *err) {
    const unsigned s = blockIdx.y;
    const unsigned long long L = len[s], total = L * (unsigned long long)C, st = rstride[s];
    const T *base = in + roff[s];
    signed char *o = q + qoff[s];
    bool bad = false;
    if (C == 1) {
        // one channel (what the encoder of a transcode or a mono file gets): 16 samples per thread — rows and the int8 rows are 16-element aligned —
        // vector loads, one 16-byte store (a sample per thread with a 64-bit division in front moved 1.4 TB/s: 2.8 of the 4.3 ms of a 2048-stream call)
        const unsigned long long nv = total / 16;
        for (unsigned long long v = (unsigned long long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (unsigned long long)gridDim.x * 256) {
            const T *r = base + 16 * v;
            unsigned w[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                double x[4];
                if constexpr (sizeof(T) == 4) { const float4 f = *reinterpret_cast<const float4 *>(r + 4 * g); x[0] = f.x; x[1] = f.y; x[2] = f.z; x[3] = f.w; }
                else { const double2 a = *reinterpret_cast<const double2 *>(r + 4 * g), b = *reinterpret_cast<const double2 *>(r + 4 * g + 2); x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y; }
                w[g] = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) w[g] |= ((unsigned)dfpwm_q1(x[e], bad) & 0xFFu) << (8 * e);
            }
            *reinterpret_cast<uint4 *>(o + 16 * v) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (blockIdx.x == 0)
            for (unsigned long long k = 16 * nv + threadIdx.x; k < total; k += 256) o[k] = (signed char)dfpwm_q1((double)base[k], bad);
    } else {
        for (unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (unsigned long long)gridDim.x * 256) {
            unsigned long long c, i;
            if (interleaved) { i = k / (unsigned long long)C; c = k - i * (unsigned long long)C; }
            else { c = k / L; i = k - c * L; }
            o[k] = (signed char)dfpwm_q1((double)base[c * st + i], bad);
        }
    }
    if (bad) atomicCAS(err, 0, 1);
}

// fused  aukit.dfpwm(d, C, sr):mono():dfpwm()  — decode (Q10 slices) → /128|/127 → mean over channels → encodePCM → encode
__global__ __launch_bounds__(64) void k_dfpwm_transcode_mono(const unsigned char *src, const unsigned long long *off, unsigned n, int C,
                                                            unsigned char *out, const unsigned long long *ooff) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const unsigned char *p = src + off[s];
    const unsigned long long nb = off[s + 1] - off[s];
    unsigned char *o = out + ooff[s];
    DfDec d{};
    DfEnc e{};
    double acc = 0;
    int c = 0, nbits = 0;
    unsigned byte_out = 0;
    unsigned long long w = 0;
    for (unsigned long long pos = 0; pos < nb; pos += 6000) {
        const unsigned long long cnt = nb - pos < 6001 ? nb - pos : 6001;
        for (unsigned long long b = 0; b < cnt; b++) {
            unsigned byte = p[pos + b];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int v = df_decode_bit(d, byte & 1);
                byte >>= 1;
                const double x = (double)v;
                acc = acc + x / (x < 0 ? 128 : 127);       // aukit.pcm table input :1082, Audio:mono :685
                if (++c == C) {
                    const double m = acc / C;               // :686
                    const double pv = m * (m < 0 ? 128 : 127);  // encodePCM :874
                    byte_out = (byte_out >> 1) | (df_encode_sample(e, (int)floor(pv)) ? 128u : 0u);
                    if (++nbits == 8) { o[w++] = (unsigned char)byte_out; nbits = 0; byte_out = 0; }
                    c = 0;
                    acc = 0;
                }
            }
        }
    }
    if (nbits) {  // pad the last byte with samples of value 0
        while (nbits < 8) { byte_out = (byte_out >> 1) | (df_encode_sample(e, 0) ? 128u : 0u); nbits++; }
        o[w++] = (unsigned char)byte_out;
    }
}

// Stereo specialisation of the fused transcode.  Profile of the generic kernel (profiles/r01_r1_dfpwm_*): 68 VALU
// instructions per decoded bit at 6.7 cycles each — one wave per CU, a single dependent chain, nothing to overlap.
// Here (a) the fp64 part  floor(m * (m<0 and 128 or 127)),  m = ((0 + L/(128|127)) + R/(128|127)) / 2  is a pure function
// of the two int8 decoder outputs, so it is tabulated once per workgroup (65 536 entries of int8 in LDS, computed with the
// reference's own fp64 operations) and the inner loop is integer-only; (b) input arrives as 16-byte vectors with the next
// vector already in flight; (c) the decoder of the next sample pair and the encoder of the previous one are independent
// chains inside one loop body, which the scheduler interleaves.
__global__ __launch_bounds__(64) void k_dfpwm_transcode_stereo(const unsigned char *src, const unsigned long long *off, unsigned n,
                                                              unsigned char *out, const unsigned long long *ooff) {
    __shared__ signed char lut[65536];
    for (int i = threadIdx.x; i < 65536; i += 64) {
        const double l = (double)((i >> 8) - 128), r = (double)((i & 255) - 128);
        double acc = 0;
        acc = acc + l / (l < 0 ? 128 : 127);   // aukit.pcm table input :1082 (left), Audio:mono :685
        acc = acc + r / (r < 0 ? 128 : 127);
        const double m = acc / 2;               // :686
        lut[i] = (signed char)(int)floor(m * (m < 0 ? 128 : 127));  // encodePCM :874 + the encoder's floor
    }
    __syncthreads();
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const unsigned char *p = src + off[s];  // 16-byte aligned (checked on the host)
    const unsigned long long nb = off[s + 1] - off[s];
    unsigned char *o = out + ooff[s];
    DfDec d{};
    DfEnc e{};
    int nbits = 0, have_l = 0, vl = 0;
    unsigned byte_out = 0;
    unsigned long long w = 0;
    const unsigned long long nvec = (nb + 15) >> 4;
    uint4 cur = nvec ? *reinterpret_cast<const uint4 *>(p) : make_uint4(0, 0, 0, 0), nxt = cur;
    unsigned long long cur_vec = 0;
    auto feed = [&](unsigned byte) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int v = df_decode_bit(d, byte & 1);
            byte >>= 1;
            if (!have_l) { vl = v; have_l = 1; }
            else {
                have_l = 0;
                const int pv = lut[((vl + 128) << 8) | (v + 128)];
                byte_out = (byte_out >> 1) | (df_encode_sample(e, pv) ? 128u : 0u);
                if (++nbits == 8) { o[w++] = (unsigned char)byte_out; nbits = 0; byte_out = 0; }
            }
        }
    };
    for (unsigned long long pos = 0; pos < nb; pos += 6000) {  // 6001-byte slices advanced by 6000 (Q10); 6000 = 375 vectors
        const unsigned long long cnt = nb - pos < 6001 ? nb - pos : 6001;
        for (unsigned long long b = 0; b < cnt; b += 16) {
            const unsigned long long vi = (pos + b) >> 4;
            if (vi != cur_vec) { cur = nxt; cur_vec = vi; }
            if (vi + 1 < nvec) nxt = *reinterpret_cast<const uint4 *>(p + 16 * (vi + 1));  // in flight while `cur` is decoded
            const unsigned words[4] = {cur.x, cur.y, cur.z, cur.w};
            const unsigned long long take = cnt - b < 16 ? cnt - b : 16;
#pragma unroll
            for (int wi = 0; wi < 4; wi++) {
                unsigned wv = words[wi];
#pragma unroll 1
                for (int jb = 0; jb < 4; jb++) {  // not unrolled: keeps the loop body inside the instruction cache
                    if ((unsigned long long)(4 * wi + jb) < take) feed(wv & 0xFF);
                    wv >>= 8;
                }
            }
        }
        // the 6001st byte of a slice is the first byte of the next vector, which the next slice starts from again
    }
    if (nbits) {  // pad the last byte with samples of value 0
        while (nbits < 8) { byte_out = (byte_out >> 1) | (df_encode_sample(e, 0) ? 128u : 0u); nbits++; }
        o[w++] = (unsigned char)byte_out;
    }
}

bool dfpwm_decode_parallel(aukit_ctx *ctx, const aukit_batch *in, int mode, int C, signed char *out, const unsigned long long *d_out_off,
                           const unsigned long long *d_out_stride, int *rc, uint64_t adv = 6000, uint64_t lead = 0, const DfSliceHook *hook = nullptr);
int dfpwm_transcode_sliced(aukit_ctx *ctx, const aukit_batch *in, signed char *mono, const unsigned long long *d_moff, const unsigned long long *d_mcount, unsigned char *out,
                           const unsigned long long *d_ooff, const uint64_t *h_ooff, int slices, bool *taken);
int dfpwm_transcode_spec(aukit_ctx *ctx, const aukit_batch *in, unsigned char *out, const unsigned long long *d_ooff, const uint64_t *h_ooff, bool *taken);  // dfpwm_spec.hip
int dfpwm_encode_spec(aukit_ctx *ctx, const signed char *rows, const uint64_t *h_in_off, const uint64_t *h_count, uint32_t n, unsigned char *out, const unsigned long long *d_ooff,
                      const uint64_t *h_ooff, bool *taken);  // dfpwm_spec.hip
int dfpwm_transcode_fused(aukit_ctx *ctx, const aukit_batch *in, signed char *mono, const unsigned long long *d_moff, const unsigned long long *d_mcount, unsigned char *out,
                          const unsigned long long *d_ooff, const uint64_t *h_ooff, bool *taken);
bool dfpwm_encode_i8_small(aukit_ctx *ctx, const signed char *in, const uint64_t *h_in_off, const uint64_t *h_count, uint32_t n, unsigned char *out, const uint64_t *h_ooff,
                           int *rc);  // exact parallel encoder for batches of a few streams (dfpwm_par.hip)
int dfpwm_encode_i8(aukit_ctx *ctx, const signed char *in, const unsigned long long *d_in_off, const unsigned long long *d_count, uint32_t n, unsigned char *out,
                    const unsigned long long *d_ooff);

// fed bytes of aukit.dfpwm's slice loop (Q10): Σ min(6001, nb - 6000k)
static uint64_t dfpwm_fed_bytes(uint64_t nb) {
    uint64_t f = 0;
    for (uint64_t pos = 0; pos < nb; pos += 6000) f += std::min<uint64_t>(6001, nb - pos);
    return f;
}

static int dfpwm_decode_audio(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, double new_rate, int interp, bool do_resample,
                              int dtype, aukit_audio **out) {
    const int C = d->channels;
    if (C < 1) return fail(AUKIT_E_ARG, "bad argument #2 (number outside of range)");
    if (d->sample_rate < 1) return fail(AUKIT_E_ARG, "bad argument #3 (number outside of range)");
    if (C > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "at most %d channels are supported", AUKIT_MAX_PLANAR_CHANNELS);
    std::vector<uint64_t> row_off((size_t)in->n * C), row_len((size_t)in->n * C), soff(in->n), sstride(in->n);
    uint64_t tot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t samples = dfpwm_fed_bytes(in->off[s + 1] - in->off[s]) * 8;
        if (samples % (uint64_t)C != 0) return fail(AUKIT_E_ARG, "bad argument #1 (uneven amount of data per channel)");  // aukit.pcm :1064
        const uint64_t L = samples / C, stride = round_up(std::max<uint64_t>(L, 1), 16);
        soff[s] = tot; sstride[s] = stride;
        for (int c = 0; c < C; c++) { row_off[(size_t)s * C + c] = tot + (uint64_t)c * stride; row_len[(size_t)s * C + c] = L; }
        tot += stride * C;
    }
    int rc = ctx->tmp_buf.ensure((size_t)tot + 64);
    if (rc) return rc;
    std::vector<uint64_t> tab(soff);
    tab.insert(tab.end(), sstride.begin(), sstride.end());
    if ((rc = upload_table(ctx, ctx->misc_buf, tab.data(), tab.size() * 8))) return rc;
    if (in->n) {
        const unsigned long long *t = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        int prc = AUKIT_OK;
        if (dfpwm_decode_parallel(ctx, in, 0, C, reinterpret_cast<signed char *>(ctx->tmp_buf.p), t, t + in->n, &prc)) {  // chunk-parallel, bit-exact (dfpwm_par.hip)
            if (prc) return prc;
            if ((rc = ctx_end_kernel(ctx, "k_df_chunks(parallel dfpwm decode)", in->total() + tot))) return rc;
            return audio_from_int_rows(ctx, SRC_I8, ctx->tmp_buf.p, row_off, row_len, in->n, C, d->sample_rate, new_rate, interp, do_resample, dtype, 127, 128, out);
        }
        hipLaunchKernelGGL(k_dfpwm_decode, dim3((in->n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off),
                           in->n, C, reinterpret_cast<signed char *>(ctx->tmp_buf.p), t, t + in->n);
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_dfpwm_decode", in->total() + tot))) return rc;
    }
    return audio_from_int_rows(ctx, SRC_I8, ctx->tmp_buf.p, row_off, row_len, in->n, C, d->sample_rate, new_rate, interp, do_resample, dtype, 127, 128, out);
}

// ================================================================= IMA ADPCM
__constant__ int c_ima_step[89] = {
    7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45,
    50, 55, 60, 66, 73, 80, 88, 97, 107, 118, 130, 143, 157, 173, 190, 209, 230, 253, 279, 307,
    337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066,
    2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899,
    15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};  // aukit.lua:161-171

struct Sat { int a, lo, hi; };  // x -> clamp(x + a, lo, hi)
// clamp(v, lo, hi) for lo <= hi as ONE v_med3_i32 (left to itself hipcc builds part of these out of compares and selects; the IMA kernels are bound
// by their VALU instructions)
AUKIT_DEV int clampi(int v, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi)); return r; }
AUKIT_DEV Sat sat_then(const Sat &f, const Sat &g) { return Sat{f.a + g.a, clampi(f.lo + g.a, g.lo, g.hi), clampi(f.hi + g.a, g.lo, g.hi)}; }
AUKIT_DEV int sat_apply(const Sat &f, int x) { return clampi(x + f.a, f.lo, f.hi); }
AUKIT_DEV Sat sat_id() { return Sat{0, -(1 << 28), 1 << 28}; }
// The scan's moves as DPP (VALU latency, no LDS crossbar): a lane without a source lane — and the rows a row_bcast does not write — keep `old`,
// which is the identity map here, so no step needs a select.  CTRL: row_shr:n = 0x110 + n, row_bcast15 = 0x142 (rows 1, 3), row_bcast31 = 0x143
// (rows 2, 3), wave_shr:1 = 0x138.
template <int CTRL, int ROW_MASK = 0xF>
AUKIT_DEV Sat sat_dpp(const Sat &v) {
    const Sat id = sat_id();
    return Sat{__builtin_amdgcn_update_dpp(id.a, v.a, CTRL, ROW_MASK, 0xF, false), __builtin_amdgcn_update_dpp(id.lo, v.lo, CTRL, ROW_MASK, 0xF, false),
               __builtin_amdgcn_update_dpp(id.hi, v.hi, CTRL, ROW_MASK, 0xF, false)};
}
AUKIT_DEV Sat wave_scan_incl(Sat v, int) {
    v = sat_then(sat_dpp<0x111>(v), v);
    v = sat_then(sat_dpp<0x112>(v), v);
    v = sat_then(sat_dpp<0x114>(v), v);
    v = sat_then(sat_dpp<0x118>(v), v);
    v = sat_then(sat_dpp<0x142, 0xA>(v), v);
    v = sat_then(sat_dpp<0x143, 0xC>(v), v);
    return v;
}
AUKIT_DEV Sat wave_excl_of(const Sat &inc) { return sat_dpp<0x138>(inc); }   // the lane before; lane 0: the identity
AUKIT_DEV Sat wave_last_of(const Sat &inc) { return Sat{__builtin_amdgcn_readlane(inc.a, 63), __builtin_amdgcn_readlane(inc.lo, 63), __builtin_amdgcn_readlane(inc.hi, 63)}; }
AUKIT_DEV int ima_index_delta(int nib) { return (nib & 4) ? ((nib & 3) + 1) * 2 : -1; }  // {-1,-1,-1,-1,2,4,6,8} :156-159

// nibble q of one (block, channel) sequence
struct NibSeq {
    const unsigned char *base;  // WAV: first data word of the block (after the 4C-byte header); raw: stream start
    int mode;                   // 0 = WAV words (C-interleaved 4-byte words, low nibble first), 1 = raw byte stream
    int C, c, top_first, interleaved;
    unsigned long long chan_len;  // raw planar: nibbles per channel
};
AUKIT_DEV unsigned nib_get(const NibSeq &s, unsigned long long q) {
    if (s.mode == 0) {
        const unsigned long long wi = q >> 3;
        const unsigned k = (unsigned)q & 7;
        const unsigned byte = s.base[(wi * s.C + s.c) * 4 + (k >> 1)];
        return (k & 1) ? byte >> 4 : byte & 15;
    }
    const unsigned long long g = s.interleaved ? q * s.C + s.c : (unsigned long long)s.c * s.chan_len + q;  // :1224-1228, :1246-1262
    const unsigned byte = s.base[g >> 1];
    const bool first = (g & 1) == 0;
    return (first == (s.top_first != 0)) ? byte >> 4 : byte & 15;
}

// decode nibbles [q0, q0 + 1024) ∩ [0, nb) of one sequence with the whole wave: lane L owns nibbles q0+16L .. +15.
// (pred, idx) enter as the state before nibble q0 and leave as the state after the chunk.  emit(q, predictor) per nibble.
template <class Emit>
AUKIT_DEV void ima_wave_chunk(const NibSeq &seq, unsigned long long q0, unsigned long long nb, int lane, int &pred, int &idx, Emit emit) {
    // straight-line code on purpose: conditional stores into the unrolled register arrays make hipcc copy the whole
    // array per iteration (1000+ v_mov_b64 in the first version of this kernel)
    unsigned nibs[16];
    const unsigned long long qa = q0 + 16ull * lane;
    const int valid = qa >= nb ? 0 : (nb - qa >= 16 ? 16 : (int)(nb - qa));
    if (seq.mode == 0) {  // WAV words: this lane's 16 nibbles are two 4-byte words of channel c, low nibble first
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const bool ok = 8 * j < valid;
            const unsigned char *wp = seq.base + (((qa >> 3) + (ok ? j : 0)) * seq.C + seq.c) * 4;
            const unsigned w = ok ? ((unsigned)wp[0] | (unsigned)wp[1] << 8 | (unsigned)wp[2] << 16 | (unsigned)wp[3] << 24) : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) nibs[8 * j + k] = (w >> (4 * k)) & 15u;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const bool ok = k < valid;
            const unsigned v = nib_get(seq, ok ? qa + k : 0);
            nibs[k] = ok ? v : 0u;
        }
    }
    // step index: composite of this lane's clamp-adds, exclusive scan, then replay
    Sat f = sat_id();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const Sat g = k < valid ? Sat{ima_index_delta(nibs[k]), 0, 88} : sat_id();
        f = sat_then(f, g);
    }
    const Sat inc = wave_scan_incl(f, lane);
    const Sat exc = wave_excl_of(inc);
    int si = sat_apply(exc, idx);
    const int idx_end = sat_apply(wave_last_of(inc), idx);
    int delta[16];
    Sat g = sat_id();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const bool ok = k < valid;
        const int step = c_ima_step[si];                                              // :2807 (si stays a valid index when !ok)
        const int nsi = clampi(si + ima_index_delta(nibs[k]), 0, 88);                 // :2808
        si = ok ? nsi : si;
        const int diff = (int)(__umul24(nibs[k] & 7, (unsigned)step) >> 2) + (step >> 3);  // :2809 (Q5)
        const int d = (nibs[k] & 8) ? -diff : diff;
        delta[k] = ok ? d : 0;
        g = sat_then(g, ok ? Sat{d, -32768, 32767} : sat_id());                       // :2810-2811
    }
    const Sat ginc = wave_scan_incl(g, lane);
    const Sat gexc = wave_excl_of(ginc);
    int p = sat_apply(gexc, pred);
    const int pred_end = sat_apply(wave_last_of(ginc), pred);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        p = clampi(p + delta[k], -32768, 32767);
        if (k < valid) emit(qa + k, p);
    }
    pred = pred_end;
    idx = idx_end;
}

// The same for a whole 1024-nibble chunk whose 16 nibbles per lane arrive as two little-endian words (one-channel WAV blocks: the
// common case of stream.adpcm): no validity predicates, no exec-masked emits, the step table read from LDS (`steps`, 89 ints) — the
// __constant__ array indexed per lane is a global load per nibble.
template <class Emit>
AUKIT_DEV void ima_wave_chunk_full(unsigned w0, unsigned w1, const int *steps, int lane, int &pred, int &idx, Emit emit) {
    unsigned nibs[16];
#pragma unroll
    for (int k = 0; k < 8; k++) { nibs[k] = (w0 >> (4 * k)) & 15u; nibs[8 + k] = (w1 >> (4 * k)) & 15u; }
    Sat f = sat_id();
#pragma unroll
    for (int k = 0; k < 16; k++) f = sat_then(f, Sat{ima_index_delta(nibs[k]), 0, 88});
    const Sat inc = wave_scan_incl(f, lane);
    const Sat exc = wave_excl_of(inc);
    int si = sat_apply(exc, idx);
    const int idx_end = sat_apply(wave_last_of(inc), idx);
    int delta[16];
    Sat g = sat_id();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int step = steps[si];                                                   // :2807
        si = clampi(si + ima_index_delta(nibs[k]), 0, 88);                            // :2808
        const int diff = (int)(__umul24(nibs[k] & 7, (unsigned)step) >> 2) + (step >> 3);  // :2809 (Q5)
        delta[k] = (nibs[k] & 8) ? -diff : diff;
        g = sat_then(g, Sat{delta[k], -32768, 32767});                                // :2810-2811
    }
    const Sat ginc = wave_scan_incl(g, lane);
    const Sat gexc = wave_excl_of(ginc);
    int p = sat_apply(gexc, pred);
    const int pred_end = sat_apply(wave_last_of(ginc), pred);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        p = clampi(p + delta[k], -32768, 32767);
        emit(k, p);
    }
    pred = pred_end;
    idx = idx_end;
}

// one row of int16 predictors per (stream, channel): whole-stream raw ADPCM or a sequence of WAV blocks
struct ImaRowJob {
    unsigned long long src_off;   // byte offset of the stream
    unsigned long long nbytes;    // stream bytes
    unsigned long long out_off;   // element offset of the int16 row
    unsigned long long nib_total; // raw: nibbles per channel
    int c;                        // channel
    int pad;
};
struct ImaParams {
    const unsigned char *src;
    const ImaRowJob *jobs;
    unsigned njobs;
    int C, block_align, mode, top_first, interleaved;
    int init_pred[AUKIT_MAX_PLANAR_CHANNELS], init_idx[AUKIT_MAX_PLANAR_CHANNELS];
    int mask_mono_index;          // aukit.wav masks the mono header step index with 0x0F (Q8)
    short *out;
    int *err;
};

// Audio path: aukit.adpcm (mode 1: one continuous sequence) / aukit.wav IMA blocks (mode 0: wave per (stream, channel), blocks in sequence)
__global__ __launch_bounds__(64) void k_ima_rows(const ImaParams P) {
    const int lane = threadIdx.x;
    const ImaRowJob job = P.jobs[blockIdx.x];
    short *orow = P.out + job.out_off;
    if (P.mode == 1) {
        NibSeq seq{P.src + job.src_off, 1, P.C, job.c, P.top_first, P.interleaved, job.nib_total};
        int pred = P.init_pred[job.c], idx = P.init_idx[job.c];
        for (unsigned long long q0 = 0; q0 < job.nib_total; q0 += 1024)
            ima_wave_chunk(seq, q0, job.nib_total, lane, pred, idx, [&](unsigned long long q, int p) { orow[q] = (short)p; });
        return;
    }
    const unsigned long long ba = (unsigned long long)P.block_align;
    const unsigned long long spb_full = (ba - 4ull * P.C) * 2 / P.C;
    unsigned long long written = 0;
    for (unsigned long long b0 = 0; b0 < job.nbytes; b0 += ba) {
        const unsigned char *blk = P.src + job.src_off + b0;
        const unsigned long long rem = job.nbytes - b0;
        unsigned long long nb = spb_full;
        if (rem < ba) nb = rem > 4ull * P.C ? (rem - 4ull * P.C) * 2 / P.C : 0;  // mono: str_sub is simply shorter (:1545); stereo validated on the host
        int pred = (short)(blk[4 * job.c] | blk[4 * job.c + 1] << 8);
        int idx = blk[4 * job.c + 2];
        if (P.C == 1 && P.mask_mono_index) idx &= 0x0F;              // :1544
        if (idx > 88) { if (lane == 0) atomicCAS(P.err, 0, 2); return; }  // expect.range(step_index, 0, 88)
        NibSeq seq{blk + 4 * P.C, 0, P.C, job.c, 0, 1, 0};
        for (unsigned long long q0 = 0; q0 < nb; q0 += 1024)
            ima_wave_chunk(seq, q0, nb, lane, pred, idx, [&](unsigned long long q, int p) { orow[written + q] = (short)p; });
        written += nb;
    }
}

// aukit.wav's IMA blocks (mode 0), one wave per (stream, channel, BLOCK) — round 4.  k_ima_rows walks a stream's blocks one after another: 4096
// streams are 4096 waves for a chip that holds 8192 and wants far more to hide a wave's scan latencies (PMC: 18 % of the wave cycles active),
// although every block starts from its own header (:1513-1545) and 220 blocks per stream wait to run side by side.
#ifndef AUKIT_IMA_BPW
#define AUKIT_IMA_BPW 8   // blocks a one-channel wave takes in turn (a wave per block — 901 120 workgroups for config 3b, each with its own table set-up — ran
#endif                    // at the dispatcher's pace: 1.11 ms for 2.25 GB)
__global__ __launch_bounds__(64) void k_ima_rows_blocks(const ImaParams P, unsigned max_blocks) {
    const int lane = threadIdx.x;
    if (P.C == 1) {
        // one channel (the shape of nearly every IMA WAV file): the lean chunk of stream.adpcm — two words per lane, the step table in LDS (indexed
        // per lane a __constant__ array is a global load per nibble), no predicates; words past the block's end are zeros, and what they leave in
        // (pred, idx) is never used: every block starts from its own header.  A lane's sixteen predictors leave as two 16-byte stores
        // (rows start on 16-byte boundaries and a block of 512 B is 1016 samples = 127 x 16 bytes).  A wave takes AUKIT_IMA_BPW blocks of its
        // stream in turn, the next block's words requested before this one's scans.
        const unsigned per = (max_blocks + AUKIT_IMA_BPW - 1) / AUKIT_IMA_BPW;
        const unsigned ji = blockIdx.x / per, bg = blockIdx.x - ji * per;
        const ImaRowJob job = P.jobs[ji];
        const unsigned long long ba = (unsigned long long)P.block_align;
        const unsigned long long spb_full = (ba - 4ull) * 2;
        __shared__ int steps[89];
        for (int i = lane; i < 89; i += 64) steps[i] = c_ima_step[i];
        __syncthreads();
        typedef unsigned u32u __attribute__((aligned(1)));
        struct Blk { unsigned w0, w1, hdr; unsigned long long nb; bool on; };
        auto load = [&](unsigned bi) -> Blk {
            Blk k{0u, 0u, 0u, 0ull, false};
            const unsigned long long b0 = (unsigned long long)bi * ba;
            if (bi >= max_blocks || b0 >= job.nbytes) return k;
            const unsigned char *blk = P.src + job.src_off + b0;
            const unsigned long long rem = job.nbytes - b0;
            k.nb = rem < ba ? (rem > 4ull ? (rem - 4ull) * 2 : 0) : spb_full;   // str_sub is simply shorter (:1545)
            k.on = true;
            k.hdr = (unsigned)blk[0] | (unsigned)blk[1] << 8 | (unsigned)blk[2] << 16;
            if (k.nb <= 1024) {   // (the usual block: one chunk — its words in flight while the block before is decoded)
                const unsigned long long nw = (k.nb + 7) / 8;
                k.w0 = 2ull * lane < nw ? *reinterpret_cast<const u32u *>(blk + 4 + 8 * lane) : 0u;
                k.w1 = 2ull * lane + 1 < nw ? *reinterpret_cast<const u32u *>(blk + 4 + 8 * lane + 4) : 0u;
            }
            return k;
        };
        Blk cur = load(bg * AUKIT_IMA_BPW);
        for (unsigned t = 0; t < AUKIT_IMA_BPW; t++) {
            const unsigned bi = bg * AUKIT_IMA_BPW + t;
            const Blk nxt = t + 1 < AUKIT_IMA_BPW ? load(bi + 1) : Blk{0u, 0u, 0u, 0ull, false};
            if (!cur.on) break;
            int pred = (short)(cur.hdr & 0xFFFFu);
            int idx = (int)(cur.hdr >> 16);
            if (P.mask_mono_index) idx &= 0x0F;              // :1544
            if (idx > 88) { if (lane == 0) atomicCAS(P.err, 0, 2); cur = nxt; continue; }  // expect.range(step_index, 0, 88)
            short *orow = P.out + job.out_off + (unsigned long long)bi * spb_full;
            const unsigned char *wbase = P.src + job.src_off + (unsigned long long)bi * ba + 4;
            const unsigned long long nb = cur.nb;
            for (unsigned long long q0 = 0; q0 < nb; q0 += 1024) {
                unsigned w0 = cur.w0, w1 = cur.w1;
                if (nb > 1024) {   // (block sizes beyond 516 bytes: chunk by chunk, loaded where they are used)
                    const unsigned long long nw = (nb - q0 + 7) / 8, wi = q0 / 8 + 2ull * lane;
                    w0 = 2ull * lane < nw ? *reinterpret_cast<const u32u *>(wbase + 4 * wi) : 0u;
                    w1 = 2ull * lane + 1 < nw ? *reinterpret_cast<const u32u *>(wbase + 4 * wi + 4) : 0u;
                }
                int pv[16];
                ima_wave_chunk_full(w0, w1, steps, lane, pred, idx, [&](int k, int p) { pv[k] = p; });
                const unsigned long long qa = q0 + 16ull * lane;
                short *o = orow + qa;
                if (qa + 16 <= nb && ((uintptr_t)o & 15) == 0) {
                    uint4 a, b;
                    a.x = (unsigned)(pv[0] & 0xFFFF) | (unsigned)pv[1] << 16; a.y = (unsigned)(pv[2] & 0xFFFF) | (unsigned)pv[3] << 16;
                    a.z = (unsigned)(pv[4] & 0xFFFF) | (unsigned)pv[5] << 16; a.w = (unsigned)(pv[6] & 0xFFFF) | (unsigned)pv[7] << 16;
                    b.x = (unsigned)(pv[8] & 0xFFFF) | (unsigned)pv[9] << 16; b.y = (unsigned)(pv[10] & 0xFFFF) | (unsigned)pv[11] << 16;
                    b.z = (unsigned)(pv[12] & 0xFFFF) | (unsigned)pv[13] << 16; b.w = (unsigned)(pv[14] & 0xFFFF) | (unsigned)pv[15] << 16;
                    reinterpret_cast<uint4 *>(o)[0] = a;
                    reinterpret_cast<uint4 *>(o)[1] = b;
                } else {
#pragma unroll
                    for (int k = 0; k < 16; k++) if (qa + k < nb) o[k] = (short)pv[k];
                }
            }
            cur = nxt;
        }
        return;
    }
    const unsigned ji = blockIdx.x / max_blocks, bi = blockIdx.x - ji * max_blocks;
    const ImaRowJob job = P.jobs[ji];
    const unsigned long long ba = (unsigned long long)P.block_align, b0 = (unsigned long long)bi * ba;
    if (b0 >= job.nbytes) return;
    const unsigned long long spb_full = (ba - 4ull * P.C) * 2 / P.C;
    short *orow = P.out + job.out_off + (unsigned long long)bi * spb_full;
    const unsigned char *blk = P.src + job.src_off + b0;
    const unsigned long long rem = job.nbytes - b0;
    unsigned long long nb = spb_full;
    if (rem < ba) nb = rem > 4ull * P.C ? (rem - 4ull * P.C) * 2 / P.C : 0;  // stereo validated on the host
    int pred = (short)(blk[4 * job.c] | blk[4 * job.c + 1] << 8);
    int idx = blk[4 * job.c + 2];
    if (idx > 88) { if (lane == 0) atomicCAS(P.err, 0, 2); return; }  // expect.range(step_index, 0, 88)
    NibSeq seq{blk + 4 * P.C, 0, P.C, job.c, 0, 1, 0};
    for (unsigned long long q0 = 0; q0 < nb; q0 += 1024)
        ima_wave_chunk(seq, q0, nb, lane, pred, idx, [&](unsigned long long q, int p) { orow[q] = (short)p; });
}

// aukit.wav's IMA blocks, one channel, one LANE per block (round 6).  k_ima_rows_blocks decodes a block with a whole wave — sixteen nibbles a lane, two
// scans of saturating maps over the wave for the step index and the predictor (:2807-2811 are clamped recurrences) — and pays 684 wave instructions a
// block of 1016 samples for it: 616 M on config 3b, the VALU pipe full for 1.0 ms.  A block is a serial chain, but 901 120 of them wait side by side: a
// lane that walks its own block pays 13 instructions a sample and no scan (13 x 1016 a wave of 64 blocks = 206 a block).  What made the lane-per-block
// kernels of round 1 slow was their memory side, not their arithmetic; here
//   * a lane reads its block 32 bytes at a time (two dwordx4, requested an iteration ahead; the block's first and last seven words, which do not fill a
//     line of output, one by one at the start),
//   * the 64 samples of an iteration are one 128-byte LINE of the row: the lane decodes h <= 7 words first so that its line boundaries are the row's
//     (rows and blocks start at multiples of 16 bytes), writes the line into its 128-byte row of LDS, and the wave stores 64 rows as 16-byte pieces,
//     eight lanes a line — whole lines, eight of them per store instruction.  Only a block's first h and last 7 - h words leave as 16-byte pieces of
//     their own (7 of 127).
// diff of (step index, nibble & 7) comes from a table in LDS (89 x 8 entries of ((n & 7) * step >> 2) + (step >> 3): one read where :2807-2809 are a read,
// a multiply, two shifts and an add), the index delta {-1,-1,-1,-1,2,4,6,8} from a constant's bit-field.
struct ImaLaneParams {
    const unsigned char *src;
    const ImaRowJob *jobs;
    const unsigned long long *blk0;   // per stream: the global index of its first block (njobs + 1 entries)
    const unsigned *wave_j0;          // per wave: the stream of its first block
    unsigned njobs;
    unsigned long long nblocks;
    int block_align, mask_mono_index;
    short *out;
    int *err;
};
constexpr int IML_RS = 36;   // dwords per lane's LDS row: a 128-byte line + 16 (b128 accesses at this stride meet no bank twice)
__global__ __launch_bounds__(64, 3) void k_ima_lanes(const ImaLaneParams P) {
    __shared__ int lut[89 * 8];
    __shared__ __attribute__((aligned(16))) unsigned rows[64 * IML_RS];
    __shared__ unsigned long long s_dst[64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 89 * 8; i += 64) { const int st = c_ima_step[i >> 3]; lut[i] = (int)(((unsigned)(i & 7) * (unsigned)st) >> 2) + (st >> 3); }   // :2809 (Q5)
    __syncthreads();
    typedef unsigned u32u __attribute__((aligned(1)));
    typedef unsigned v4uu __attribute__((ext_vector_type(4), aligned(4)));
    typedef unsigned v4ua __attribute__((ext_vector_type(4)));
    const unsigned long long g = (unsigned long long)blockIdx.x * 64 + (unsigned)lane;
    const bool on = g < P.nblocks;
    // the lane's block: stream ji (the last one whose first block is not beyond g), block bi of it
    // (the wave's first block's stream comes from the host — a binary search here was twelve global loads one after the other in front of every wave's
    // 33 us of work; the lanes walk on from it: a wave spans a stream or two unless the streams are shorter than a block)
    unsigned lo = P.wave_j0[blockIdx.x];
    while (__any(on && lo + 1 < P.njobs && P.blk0[lo + 1] <= g)) { if (on && lo + 1 < P.njobs && P.blk0[lo + 1] <= g) lo++; }
    const ImaRowJob job = P.jobs[lo];
    const unsigned long long ba = (unsigned long long)P.block_align, bi = on ? g - P.blk0[lo] : 0ull, b0 = bi * ba;
    const unsigned char *blk = P.src + job.src_off + b0;
    const unsigned long long rem = job.nbytes > b0 ? job.nbytes - b0 : 0ull;
    const unsigned long long nbytes = on ? (rem < ba ? rem : ba) : 0ull;
    const unsigned nb = nbytes > 4 ? (unsigned)(nbytes - 4) * 2u : 0u;   // samples of the block: str_sub is simply shorter (:1545)
    const unsigned nwf = nb >> 3, rag = nb & 7u;                        // whole words, samples of a last part word (a stream's short last block only)
    int pred = 0, idx32 = 0;
    if (nbytes >= 3) {
        pred = (short)((unsigned)blk[0] | (unsigned)blk[1] << 8);
        int idx = blk[2];
        if (P.mask_mono_index) idx &= 0x0F;              // :1544
        if (idx > 88) { atomicCAS(P.err, 0, 2); idx = 88; }   // expect.range(step_index, 0, 88)
        idx32 = idx * 32;
    }
    const unsigned long long e0 = job.out_off + bi * ((ba - 4ull) * 2ull);   // the block's first sample in the rows (elements)
    const unsigned phi = (unsigned)((2ull * e0) & 127ull);               // (rows, and blocks, start at multiples of 16 bytes)
    const unsigned hw = min(nwf, ((128u - phi) & 127u) >> 4);            // words in front of the block's first whole line
    const unsigned m = (nwf - hw) >> 3;                                  // whole lines
    const unsigned tw = nwf - hw - 8u * m;                               // words behind the last whole line
    const unsigned char *wp = blk + 4;
    char *const ob = reinterpret_cast<char *>(P.out) + 2ull * e0;
    // the head's and the tail's words: requested now, used when their turn comes
    unsigned hdw[7], tlw[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        hdw[i] = (unsigned)i < hw ? *reinterpret_cast<const u32u *>(wp + 4 * i) : 0u;
        tlw[i] = (unsigned)i < tw ? *reinterpret_cast<const u32u *>(wp + 4 * (hw + 8u * m + (unsigned)i)) : 0u;
    }
    unsigned ragw = 0;
    if (rag) { const unsigned char *q = wp + 4 * nwf; for (unsigned i = 0; i < rag / 2; i++) ragw |= (unsigned)q[i] << (8 * i); }
    v4ua va = v4ua{0, 0, 0, 0}, vb = v4ua{0, 0, 0, 0};
    if (m > 0) { va = *reinterpret_cast<const v4uu *>(wp + 4 * hw); vb = *reinterpret_cast<const v4uu *>(wp + 4 * hw + 16); }
    const char *const lutb = reinterpret_cast<const char *>(lut);
    // one word = eight nibbles, low nibble first (:1546 / :2803-2806) -> eight predictors as four dwords
    auto word = [&](unsigned w, unsigned &o0, unsigned &o1, unsigned &o2, unsigned &o3) {
        int pv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned sh = __builtin_amdgcn_ubfe(w, 4 * k, 3) << 2;                                   // 4 (n & 7)
            const int diff = *reinterpret_cast<const int *>(lutb + (unsigned)idx32 + sh);                  // :2807, :2809
            const int sg = __builtin_amdgcn_sbfe((int)w, 4 * k + 3, 1);                                    // n & 8: all ones
            pred = clampi(pred + ((diff ^ sg) - sg), -32768, 32767);                                       // :2810-2811
            const int dd = (int)__builtin_amdgcn_ubfe(0x97530000u, sh, 4);                                 // index delta + 1: {0,0,0,0,3,5,7,9}
            idx32 = clampi(idx32 + (dd << 5) - 32, 0, 88 * 32);                                            // :2808
            pv[k] = pred;
        }
        o0 = __builtin_amdgcn_perm((unsigned)pv[1], (unsigned)pv[0], 0x05040100u); o1 = __builtin_amdgcn_perm((unsigned)pv[3], (unsigned)pv[2], 0x05040100u);
        o2 = __builtin_amdgcn_perm((unsigned)pv[5], (unsigned)pv[4], 0x05040100u); o3 = __builtin_amdgcn_perm((unsigned)pv[7], (unsigned)pv[6], 0x05040100u);
    };
    // ---- the head: h words in front of the block's first whole line.  Their 16-byte pieces wait in registers: they share a line with the last pieces of the
    // block BEFORE (the lane before, where that is the same stream's previous block) and leave with them at the end — whole lines again.  (Stored one by one
    // as they were decoded, the head's and the tail's pieces were 896 of a wave's 1856 line writes, each 16 bytes of a line: 0.17 of the kernel's 0.72 ms)
    v4ua hp[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        hp[i] = v4ua{0, 0, 0, 0};
        if (__any((unsigned)i < hw)) {
            unsigned o0, o1, o2, o3;
            const int p0 = pred, i0 = idx32;
            word(hdw[i], o0, o1, o2, o3);
            if ((unsigned)i < hw) hp[i] = v4ua{o0, o1, o2, o3};
            else { pred = p0; idx32 = i0; }
        }
    }
    // ---- the whole lines
    unsigned *const row = rows + lane * IML_RS;
    char *const lb = ob + 16ull * hw;   // the lane's first whole line
    unsigned mmax = m;
    for (int o = 32; o; o >>= 1) mmax = max(mmax, (unsigned)__shfl_xor((int)mmax, o));
    // (the order inside a turn: wait for this line's words — asked for a turn ago —, THEN store the line before, then ask for the next line's words, then
    // decode.  Loads and stores share vmcnt and hipcc waits vmcnt(0) where both kinds are in flight: with the stores at the end of a turn the wait at the
    // top of the next one was a wait for them, a store latency per line and lane — 0.72 ms for 0.35 ms of instructions)
    bool pending = false;
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int piece = lane & 7;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int r = 8 * i + (lane >> 3);
            const unsigned long long d = s_dst[r];
#ifndef AUKIT_IML_NOSTORE
            if (d != ~0ull) *reinterpret_cast<v4ua *>(reinterpret_cast<char *>(P.out) + d + 16 * piece) = *reinterpret_cast<const v4ua *>(rows + r * IML_RS + 4 * piece);
#else
            if (d == 1ull) *reinterpret_cast<v4ua *>(reinterpret_cast<char *>(P.out) + d + 16 * piece) = *reinterpret_cast<const v4ua *>(rows + r * IML_RS + 4 * piece);
#endif
        }
        __builtin_amdgcn_wave_barrier();   // (the rows are free again)
    };
    for (unsigned it = 0; it < mmax; it++) {
        const bool act = it < m;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(va), "+v"(vb) : : "memory");
        const v4ua ca = va, cb = vb;
        if (pending) flush();
        if (it + 1 < m) { const unsigned char *q = wp + 4 * (hw + 8u * (it + 1)); va = *reinterpret_cast<const v4uu *>(q); vb = *reinterpret_cast<const v4uu *>(q + 16); }   // the next line's words
        pending = __any(act);
        if (pending) {
            const int p0 = pred, i0 = idx32;
            unsigned o[32];
            word(ca.x, o[0], o[1], o[2], o[3]);     word(ca.y, o[4], o[5], o[6], o[7]);
            word(ca.z, o[8], o[9], o[10], o[11]);   word(ca.w, o[12], o[13], o[14], o[15]);
            word(cb.x, o[16], o[17], o[18], o[19]); word(cb.y, o[20], o[21], o[22], o[23]);
            word(cb.z, o[24], o[25], o[26], o[27]); word(cb.w, o[28], o[29], o[30], o[31]);
            if (!act) { pred = p0; idx32 = i0; }
            if (act) {
#pragma unroll
                for (int q = 0; q < 8; q++) *reinterpret_cast<v4ua *>(row + 4 * q) = v4ua{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
            }
            s_dst[lane] = act ? (unsigned long long)(lb - reinterpret_cast<char *>(P.out)) + 128ull * it : ~0ull;
        }
    }
    if (pending) flush();
    // ---- the tail: the words behind the last whole line go into the lane's row, the head of the lane after joins them there (the same line of the same
    // row of samples); a lane whose block has no such neighbour before it in the wave stores its head pieces itself
    char *const tb = lb + 128ull * m;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        if (__any((unsigned)i < tw)) {
            unsigned o0, o1, o2, o3;
            const int p0 = pred, i0 = idx32;
            word(tlw[i], o0, o1, o2, o3);
            if ((unsigned)i < tw) *reinterpret_cast<v4ua *>(row + 4 * i) = v4ua{o0, o1, o2, o3};
            else { pred = p0; idx32 = i0; }
        }
    }
    // (the lane before holds the same stream's previous block, and that block is a whole one: then its tail line is this block's head line)
    const unsigned lo_b = (unsigned)__shfl((int)lo, lane > 0 ? lane - 1 : 0);
    const unsigned tw_b = (unsigned)__shfl((int)tw, lane > 0 ? lane - 1 : 0);
    const bool joined = on && hw > 0 && lane > 0 && lo_b == lo && bi > 0 && tw_b + hw == 8u;
    if (joined) {
        unsigned *const prow = rows + (lane - 1) * IML_RS;
#pragma unroll
        for (int i = 0; i < 7; i++) if ((unsigned)i < hw) *reinterpret_cast<v4ua *>(prow + 4 * (tw_b + (unsigned)i)) = hp[i];
    } else {
#pragma unroll
        for (int i = 0; i < 7; i++) if ((unsigned)i < hw) *reinterpret_cast<v4ua *>(ob + 16 * i) = hp[i];
    }
    const bool joined_n = __shfl((int)joined, lane < 63 ? lane + 1 : 63) != 0 && lane < 63;
    const unsigned hw_n = (unsigned)__shfl((int)hw, lane < 63 ? lane + 1 : 63);
    const unsigned cover = tw + (joined_n ? hw_n : 0u);   // pieces 0 .. cover - 1 of the tail line are in the row
    s_dst[lane] = cover ? ((unsigned long long)(tb - reinterpret_cast<char *>(P.out)) | ((unsigned long long)cover << 56)) : ~0ull;
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int piece = lane & 7;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int r = 8 * i + (lane >> 3);
            const unsigned long long d = s_dst[r];
            if (d != ~0ull && (unsigned)piece < (unsigned)(d >> 56))
                *reinterpret_cast<v4ua *>(reinterpret_cast<char *>(P.out) + (d & 0x00FFFFFFFFFFFFFFull) + 16 * piece) = *reinterpret_cast<const v4ua *>(rows + r * IML_RS + 4 * piece);
        }
    }
    if (__any(rag != 0u)) {
        unsigned o[4];
        word(ragw, o[0], o[1], o[2], o[3]);
        if (rag) { short *t = reinterpret_cast<short *>(tb + 16 * tw); for (unsigned k = 0; k < rag; k++) t[k] = (short)(o[k >> 1] >> (16 * (k & 1))); }
    }
}

// stream.adpcm  aukit.lua:2788-2831: wave per block, decoded block kept in LDS as the reference's doubles, resampled in place
struct ImaStreamParams {
    const unsigned char *src;
    const unsigned long long *off;       // per stream byte offsets
    const unsigned long long *blk0;      // per stream: index of its first block in the global block list (n+1)
    const unsigned long long *out_off;   // per stream: element offset of output channel 0
    const unsigned long long *out_stride;
    unsigned nstreams;
    unsigned long long nblocks;
    int C, block_align, mono;
    unsigned newlen_full;                // floor(samplesPerBlock * ratio)
    double ratio, rcp;
    int exact_rcp;
    int cap;                             // LDS doubles per channel
    void *out;
    int *err;
    // mono, linear / cubic, integer sample rate: guarded short-cut under the floor() (same argument as floor_wave.hip):
    // x - 1 = j * fa / fb exactly; fmagic = ceil(2^32 / fb)
    int fast;
    unsigned fa, fb, fmagic;
    double inv_fb;
    unsigned fdq, fdr;                   // 64 fa = fdq fb + fdr
    unsigned long long bps;              // k_ima_stream_f32: blocks per stream when that is the same for every stream, else 0
    unsigned mid_lo, mid_end_full;       // k_ima_stream_f32: the clean rows of a full block (host-made: two 64-bit divisions per block otherwise)
    unsigned qstep;                      // k_ima_stream_f32<…, PH>: 64 PH fa / fb, the table entries that 64 PH outputs span (a whole number)
    unsigned *audit;                     // k_ima_stream_f32<…, 0, true>: [0] bits of the largest |tier 1 - tier 2| (a non-negative float), [1] outputs compared
};

template <int INTERP, typename OUT_T>
__global__ __launch_bounds__(256) void k_ima_stream(const ImaStreamParams P) {
    extern __shared__ double smd[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwv = (int)(blockDim.x >> 6);  // waves per workgroup (fewer when a block's LDS footprint is large)
    double *sm = smd + (size_t)wave * P.cap * P.C;
    const unsigned long long ba = (unsigned long long)P.block_align;
    const unsigned long long nwr = (ba - 4ull * P.C) / (4ull * P.C);  // real word groups per block
    const double spb = (double)(ba - 4ull * P.C) * 2 / P.C;           // samplesPerBlock :2765
    ResampleParams RP;  // only the position fields are used by pos_of / eval_at
    RP.ratio = P.ratio; RP.rcp = P.rcp; RP.exact_rcp = P.exact_rcp; RP.sinc_w = 10;
    for (unsigned long long gb = (unsigned long long)blockIdx.x * nwv + wave; gb < P.nblocks; gb += (unsigned long long)gridDim.x * nwv) {
        // stream of this block: binary search in blk0 (wave-uniform)
        unsigned lo = 0, hi = P.nstreams;
        while (hi - lo > 1) { unsigned mid = (lo + hi) >> 1; if (P.blk0[mid] <= gb) lo = mid; else hi = mid; }
        const unsigned s = lo;
        const unsigned long long bi = gb - P.blk0[s];
        const unsigned long long nbytes = P.off[s + 1] - P.off[s], b0 = bi * ba, rem = nbytes - b0;
        const unsigned char *blk = P.src + P.off[s] + b0;
        // word groups decoded: i = 4C .. blockAlign inclusive (junk word, Q6) while #data >= n + i + 4C  :2800-2802
        long long ng = (long long)((rem - 1) / (4ull * P.C)) - 1;
        if (ng > (long long)nwr + 1) ng = (long long)nwr + 1;
        if (ng < 0) ng = 0;
        const unsigned long long nb = (unsigned long long)ng * 8;     // #d[1]
        unsigned newlen = P.newlen_full;
        if ((double)nb < spb) newlen = (unsigned)floor((double)nb * P.ratio);  // :2817
        bool bad = false;
        for (int c = 0; c < P.C; c++) {
            int pred = (short)(blk[4 * c] | blk[4 * c + 1] << 8);
            int idx = blk[4 * c + 2];                                  // used unmasked :2799
            if (idx > 88 && nb > 0) { bad = true; break; }
            NibSeq seq{blk + 4 * P.C, 0, P.C, c, 0, 1, 0};
            double *ch = sm + (size_t)c * P.cap;
            for (unsigned long long q0 = 0; q0 < nb; q0 += 1024)
                ima_wave_chunk(seq, q0, nb, lane, pred, idx, [&](unsigned long long q, int p) { ch[q + (q >> 4)] = p < 0 ? (double)p * (1.0 / 128) : div_rcp((double)p, 127.0, 1.0 / 127.0); });  // :2812, skewed slots
        }
        if (bad) { if (lane == 0) atomicCAS(P.err, 0, 3); continue; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        Seg sg;
        sg.w_lo = 1; sg.w_hi = (int)nb;
        OUT_T *obase = reinterpret_cast<OUT_T *>(P.out) + P.out_off[s] + bi * (unsigned long long)P.newlen_full;
        const unsigned long long ostride = P.out_stride[s];
        if constexpr (INTERP == AUKIT_INTERP_LINEAR || INTERP == AUKIT_INTERP_CUBIC) {
            if (P.fast) {  // wave-uniform: one channel.  The output is floor()ed (:2823): evaluate the polynomial with exact rational
                // positions and FMA Horner form, keep the result unless it is within 1e-6 of an integer (then, and at the table's
                // ends where the nil fall-backs apply, run the reference-order code).  Bit-exact: see floor_wave.hip for the margin.
                const int nbi = (int)nb;
                // floor(j fa / fb) and the remainder: by the reciprocal for the lane's first output (exact: (newlen * fa + fb) * fb
                // < 2^32), then advanced by additions — 64 outputs further is 64 fa = dq fb + dr
                unsigned q0 = __umulhi((unsigned)lane * P.fa, P.fmagic);
                unsigned rem = (unsigned)lane * P.fa - q0 * P.fb;
                for (unsigned j = lane; j < newlen; j += 64, q0 += P.fdq, rem += P.fdr) {
                    if (rem >= P.fb) { rem -= P.fb; q0++; }
                    const int k = (int)q0 + 1;                  // floor(x)
                    // straight-line for the whole wave (branches cost more than the arithmetic they skip): taps outside the table are
                    // read from a safe slot and the lane is sent to the reference-order code; rem == 0 gives fx = 0 and v = p1 exactly
                    bool ok = INTERP == AUKIT_INTERP_CUBIC ? (k >= 3 && k + 2 <= nbi) : (k >= 2 && k + 1 <= nbi);  // one spare tap on the left (x may round below an integer)
                    const int s1 = ok ? k - 1 : 2;  // slot of table index k (skewed: one pad per 16)
                    const double p1 = sm[s1 + (s1 >> 4)];
                    const double fx = (double)rem * P.inv_fb;
                    const double p2 = sm[s1 + 1 + ((s1 + 1) >> 4)];
                    double v;
                    if constexpr (INTERP == AUKIT_INTERP_LINEAR) v = __builtin_fma(p2 - p1, fx, p1);
                    else {
                        const double p0 = sm[s1 - 1 + ((s1 - 1) >> 4)], p3 = sm[s1 + 2 + ((s1 + 2) >> 4)];
                        const double c3 = __builtin_fma(1.5, p1 - p2, 0.5 * (p3 - p0));
                        const double c2 = __builtin_fma(-2.5, p1, p0) + __builtin_fma(2.0, p2, -0.5 * p3);
                        const double c1 = 0.5 * (p2 - p0);
                        v = __builtin_fma(__builtin_fma(__builtin_fma(c3, fx, c2), fx, c1), fx, p1);
                    }
                    const double fr = v - floor(v);
                    ok = ok && fr > 1e-6 && fr < 1 - 1e-6;
                    if (!ok) { bool isint; v = eval_at<INTERP, true>(RP, sg, sm, 1, j, &isint); }
                    obase[j] = (OUT_T)(int)lua_clamp(floor(v), -128, 127);
                }
                __builtin_amdgcn_wave_barrier();
                continue;
            }
        }
        for (unsigned j = lane; j < newlen; j += 64) {  // :2818-2828
            bool isint;
            if (P.mono) {
                double acc = 0;
                for (int c = 0; c < P.C; c++) acc = acc + eval_at<INTERP, true>(RP, sg, sm + (size_t)c * P.cap, 1, j, &isint);
                const double v = lua_clamp(floor(acc / P.C), -128, 127);
                obase[j] = (OUT_T)(int)v;
            } else {
                for (int c = 0; c < P.C; c++) {
                    const double v = lua_clamp(floor(eval_at<INTERP, true>(RP, sg, sm + (size_t)c * P.cap, 1, j, &isint)), -128, 127);
                    obase[(unsigned long long)c * ostride + j] = (OUT_T)(int)v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// a table entry of k_ima_stream_f32 — the f32 sample RN(p * (1 / 128 | RN(1 / 127))) — read back as the reference's double p / (p < 0 and 128 or 127)
// (:2812): the predictor is recovered exactly (|p| < 2^15: the entry's relative error of 2^-23 moves p * 127 by less than 1/2)
struct CvImaF32 {
    static AUKIT_DEV double cv(double v) {
        const double p = rint(v * (v < 0 ? 128.0 : 127.0));
        return p < 0 ? p * (1.0 / 128) : div_rcp(p, 127.0, 1.0 / 127.0);
    }
};

// ---- stream.adpcm, one channel, linear / cubic, integer rates with at most 512 output phases (22 050 → 48 000 Hz has 320): config 3a.
// The block is kept in LDS as the int16 predictors themselves (exact floats, 4 bytes a sample) and every output is decided in up to
// three tiers, each taken only when it is certain of the integer interval the reference's value falls into (the output is floor()ed,
// :2823) — the argument of floor_wave.hip, with one more difficulty: the samples are p / 128 (p < 0) or p / 127 (:2812), not exact in
// f32 and with two scales inside one interpolation window.
//   1. f32, straight-line for the whole wave, on the samples themselves as f32 table entries (round 3; rounds 1-2 kept the integer predictors and
//      evaluated two tap-weighted sums, over p and over max(p, 0), to keep both scales exact: eight FMAs, four v_med3 and a combine per output
//      in a kernel that is bound by its VALU instructions).  An entry is within 2^-23 relative of the sample (3.1e-5 at the largest magnitude,
//      258); the weights of the output's phase come from an LDS table (computed in fp64 on the host, rounded once; the position itself is
//      exact); one multiply and three FMAs round at magnitudes below 512 (3e-5 each): < 2.5e-4 in all — the bound k_ms_wave's mono path
//      lives by.  Taken when the value lies more than 5e-4 (TIER1_GUARD; 1e-3 until late round 3) away from an integer.
//   2. (about one output in 500) the polynomial in fp64 on exact doubles and the exact rational position, FMA Horner form, taken
//      when more than 1e-6 away from an integer (the reference's x carries < 1024 · 2^-53 of rounding, times a slope below 1600).
//   3. otherwise, and where the nil fall-backs of the block's ends apply: the reference-order code on the same table.
// Same-box A/B against the fp64-only short-cut of k_ima_stream (AUKIT_IMA_F64=1) in DESIGN.md §3.
// PH > 0 (round 3): the phases in registers.  A block's outputs start at phase 0 and rows R and R + PH (PH = fb / gcd(fb, 64): 5 at 22 050 and 44 100 Hz)
// share their phases, so a lane keeps the weights and the first-tap offsets of its PH phases for the whole launch; the clean rows of a block then run
// in groups of PH rows without position arithmetic, the guard's turn-downs noted in a bit each and redone behind the group by one copy of tiers 2-3.
// Same f32 operations on the same values as the generic rows: the same tier decisions, bit for bit.
// tier 1's guard: its error is below 2.5e-4 (entries 5.3e-5, four roundings below 512: 9.3e-5 — 1.5e-4 by the count above, 2.5e-4 with slack); 1e-3 until late
// round 3.  The turn-downs cost this kernel a sixth of its instructions; 5e-4 halves them and keeps a factor of two over the bound.
#ifndef AUKIT_IMA_TIER1_GUARD
#define AUKIT_IMA_TIER1_GUARD 5e-4f
#endif
constexpr float TIER1_GUARD = AUKIT_IMA_TIER1_GUARD;
constexpr unsigned IMA_DL_CAP = 32;   // k_ima_stream_f32's list of deferred outputs (entries; at most 64 lanes: one per lane when flushed)

template <int INTERP, typename OUT_T, int PH, bool AUDIT = false>
__global__ __launch_bounds__(256) void k_ima_stream_f32(const ImaStreamParams P, const float *__restrict__ wg) {
    [[maybe_unused]] float amax = 0.f;     // AUDIT (floor_wave.hip): what tier 1's f32 value really differs from tier 2's fp64 one by
    [[maybe_unused]] unsigned acnt = 0;
    extern __shared__ float smf[];
    constexpr int WF = INTERP == AUKIT_INTERP_CUBIC ? 4 : 1;  // floats per phase: w0..w3 / fx
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned i = threadIdx.x; i < P.fb * WF; i += 256) smf[i] = wg[i];
    int *const steps = reinterpret_cast<int *>(smf + ((P.fb * WF + 3) & ~3u));  // ima_step_table (:161-171), 89 entries (96 slots)
    if (threadIdx.x < 89) steps[threadIdx.x] = c_ima_step[threadIdx.x];
    __syncthreads();
    const float *const wt = smf;
    float *const sm = smf + ((P.fb * WF + 3) & ~3u) + 96 + (size_t)wave * P.cap;  // sm[q] = d[q + 1], unskewed: see the decode below
    const unsigned long long ba = (unsigned long long)P.block_align;
    const unsigned long long nwr = (ba - 4ull) / 4ull;     // real word groups per block
    const double spb = (double)(ba - 4ull) * 2;            // samplesPerBlock :2765
    ResampleParams RP;  // only the position fields are used by pos_of / eval_at
    RP.ratio = P.ratio; RP.rcp = P.rcp; RP.exact_rcp = P.exact_rcp; RP.sinc_w = 10;
    const float c127 = 1.0f / 127.0f;
    constexpr int NPH = PH > 0 ? PH : 1;
    [[maybe_unused]] float4 pw[NPH];     // cubic: w0..w3 of the phase; linear: .x = fx
    [[maybe_unused]] unsigned qo[NPH];
    if constexpr (PH > 0) {
#pragma unroll
        for (int i = 0; i < PH; i++) {
            const unsigned n = (unsigned)(64 * i + lane) * P.fa;
            const unsigned q = __umulhi(n, P.fmagic), r = n - q * P.fb;
            qo[i] = q;
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) pw[i] = make_float4(wt[r], 0.f, 0.f, 0.f);
            else pw[i] = *reinterpret_cast<const float4 *>(wt + 4 * r);
        }
    }
    for (unsigned long long gb = (unsigned long long)blockIdx.x * 4 + wave; gb < P.nblocks; gb += (unsigned long long)gridDim.x * 4) {
        unsigned lo = 0, hi = P.nstreams;  // stream of this block (wave-uniform): a division when every stream has the same number of blocks, else a binary search in blk0
        if (P.bps) lo = (unsigned)(gb / P.bps);
        else while (hi - lo > 1) { unsigned mid = (lo + hi) >> 1; if (P.blk0[mid] <= gb) lo = mid; else hi = mid; }
        const unsigned s = lo;
        const unsigned long long bi = gb - P.blk0[s];
        const unsigned long long nbytes = P.off[s + 1] - P.off[s], b0 = bi * ba, rem_bytes = nbytes - b0;
        const unsigned char *blk = P.src + P.off[s] + b0;
        long long ng = (long long)((rem_bytes - 1) / 4ull) - 1;  // word groups decoded, junk word included (Q6)  :2800-2802
        if (ng > (long long)nwr + 1) ng = (long long)nwr + 1;
        if (ng < 0) ng = 0;
        const unsigned long long nb = (unsigned long long)ng * 8;  // #d[1]
        unsigned newlen = P.newlen_full;
        if ((double)nb < spb) newlen = (unsigned)floor((double)nb * P.ratio);  // :2817
        int pred = (short)(blk[0] | blk[1] << 8);
        int idx = blk[2];  // used unmasked :2799
        if (idx > 88 && nb > 0) { if (lane == 0) atomicCAS(P.err, 0, 3); continue; }
        // The table is NOT skewed: a lane owns 16 consecutive entries, so its ds_write_b32 stores collide 16-way in the banks — a few
        // hundred LDS cycles per block on an LDS pipe that is a twentieth busy, against 2 address instructions per tap and 3 per
        // store on the VALU, which is what the kernel is bound by (PMC: profiles/).
        NibSeq seq{blk + 4, 0, 1, 0, 0, 1, 0};
        if (nb == 1024) {  // a whole block (all but possibly the stream's last): two words per lane, straight-line
            const unsigned char *wp = blk + 4 + 8 * lane;
            unsigned w0, w1;
            if ((reinterpret_cast<uintptr_t>(blk) & 3) == 0) { const uint2 ww = *reinterpret_cast<const uint2 *>(wp); w0 = ww.x; w1 = ww.y; }  // wave-uniform
            else { w0 = (unsigned)wp[0] | (unsigned)wp[1] << 8 | (unsigned)wp[2] << 16 | (unsigned)wp[3] << 24; w1 = (unsigned)wp[4] | (unsigned)wp[5] << 8 | (unsigned)wp[6] << 16 | (unsigned)wp[7] << 24; }
            float *const mine = sm + 16 * lane;
            ima_wave_chunk_full(w0, w1, steps, lane, pred, idx, [&](int k, int p) { mine[k] = (float)p * (p < 0 ? 1.0f / 128.0f : c127); });
        } else {
            for (unsigned long long q0 = 0; q0 < nb; q0 += 1024) ima_wave_chunk(seq, q0, nb, lane, pred, idx, [&](unsigned long long q, int p) { sm[q] = (float)p * (p < 0 ? 1.0f / 128.0f : c127); });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        Seg sg;
        sg.w_lo = 1; sg.w_hi = (int)nb;
        OUT_T *obase = reinterpret_cast<OUT_T *>(P.out) + P.out_off[s] + bi * (unsigned long long)P.newlen_full;
        const int nbi = (int)nb;
        auto exact_v = [&](int k, unsigned rem) -> double {   // tier 2's value (taps inside the table)
            const int s1 = k - 1;
            const double p1 = CvImaF32::cv((double)sm[s1]), p2 = CvImaF32::cv((double)sm[s1 + 1]);
            const double fx = (double)rem * P.inv_fb;
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) return __builtin_fma(p2 - p1, fx, p1);
            else {
                const double p0 = CvImaF32::cv((double)sm[s1 - 1]), p3 = CvImaF32::cv((double)sm[s1 + 2]);
                const double c3 = __builtin_fma(1.5, p1 - p2, 0.5 * (p3 - p0));
                const double c2 = __builtin_fma(-2.5, p1, p0) + __builtin_fma(2.0, p2, -0.5 * p3);
                const double c1 = 0.5 * (p2 - p0);
                return __builtin_fma(__builtin_fma(__builtin_fma(c3, fx, c2), fx, c1), fx, p1);
            }
        };
        auto slow = [&](int k, unsigned rem, unsigned j, bool inside) -> float {  // tiers 2 and 3
            double v = 0;
            bool ok = false;
            if (inside) {
                v = exact_v(k, rem);
                const double fr = v - floor(v);
                ok = fr > 1e-6 && fr < 1 - 1e-6;
            }
            if (!ok) { bool isint; v = eval_at<INTERP, false, float, CvImaF32>(RP, sg, sm, 1, j, &isint); }
            return (float)lua_clamp(floor(v), -128, 127);
        };
        // floor(j fa / fb) and the remainder: by the reciprocal for the lane's first output (exact: (newlen * fa + fb) * fb < 2^32),
        // then advanced by additions — 64 outputs further is 64 fa = dq fb + dr
        unsigned q0 = __umulhi((unsigned)lane * P.fa, P.fmagic);
        unsigned rem = (unsigned)lane * P.fa - q0 * P.fb;
        // Rows of 64 outputs whose taps all lie inside the table and which are all wanted — every row of a block but its first and its last one
        // or two — run a copy of the loop body without the `inside` / `active` bookkeeping (round 3: the kernel is bound by its VALU instructions,
        // and those were six of a row's ~50): rows rb with rb fa >= 2 fb (floor(x) >= 3) and floor((rb + 63) fa / fb) + 3 <= nb, rb + 63 < newlen
        const unsigned mid_lo = P.mid_lo;   // round_up64(ceil(3 fb / fa)) (3 fb: a margin of one sample)
        unsigned mid_end = P.mid_end_full;  // rows rb with rb + 64 <= mid_end are clean
        if (nb != 1024 || newlen != P.newlen_full) mid_end = 0;
        if (mid_end == 0 && (nb != 1024 || newlen != P.newlen_full) && newlen >= 64 && nbi >= 4) {   // a stream's last, shorter block
            const unsigned long long jm = (((unsigned long long)(nbi - 3) * P.fb) / P.fa);       // outputs j <= jm - 1 have floor(j fa / fb) + 3 <= nb
            const unsigned long long lim = jm < newlen ? jm : newlen;
            mid_end = (unsigned)(lim & ~63ull);
        }
        unsigned *const dl = reinterpret_cast<unsigned *>(sm + (P.cap - (int)IMA_DL_CAP));   // the table's last IMA_DL_CAP slots (the host's capf leaves them free)
        unsigned dcnt = 0;
        auto flush_deferred = [&]() {
            if (!dcnt) return;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if ((unsigned)lane < dcnt) {
                const unsigned j = dl[lane], n = j * P.fa;
                const unsigned q = __umulhi(n, P.fmagic);
                const float fl = slow((int)q + 1, n - q * P.fb, j, true);
                obase[j] = (OUT_T)(int)__builtin_amdgcn_fmed3f(fl, -128.0f, 127.0f);
            }
            dcnt = 0;
            __builtin_amdgcn_wave_barrier();
        };
        auto row = [&](unsigned rb, auto cleanc) {
            constexpr bool CLEAN = decltype(cleanc)::value;
            const unsigned j = rb + lane;
            const bool active = CLEAN || j < newlen;
            if (rem >= P.fb) { rem -= P.fb; q0++; }
            const int k = (int)q0 + 1;  // floor(x)
            // straight-line for the whole wave: taps outside the table are read from a safe slot and the lane is sent to tier 3
            const bool inside = CLEAN || (INTERP == AUKIT_INTERP_CUBIC ? (k >= 3 && k + 2 <= nbi) : (k >= 2 && k + 1 <= nbi));  // one spare tap on the left (x may round below an integer)
            const int s1 = inside ? k - 1 : 2;
            const float *tp = sm + s1;
            const float p1 = tp[0], p2 = tp[1];
            float v;
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) v = __builtin_fmaf(p2 - p1, wt[rem], p1);
            else {
                const float4 w = *reinterpret_cast<const float4 *>(wt + 4 * rem);
                const float p0 = tp[-1], p3 = tp[2];
                v = __builtin_fmaf(w.w, p3, __builtin_fmaf(w.z, p2, __builtin_fmaf(w.y, p1, w.x * p0)));
            }
            if constexpr (AUDIT) { if (active && inside) { amax = fmaxf(amax, (float)fabs((double)v - exact_v(k, rem))); acnt++; } }
            float fl = floorf(v);
            const float fr = v - fl;
            const bool accept = inside && fr > TIER1_GUARD && fr < 1 - TIER1_GUARD;
            if (active && !accept) fl = slow(k, rem, j, inside);
            if (active) obase[j] = (OUT_T)(int)__builtin_amdgcn_fmed3f(fl, -128.0f, 127.0f);  // :2824 (one v_med3; fl is finite)
        };
        // (the turn-downs of a GROUP are redone behind the group; noting them per block — a bit per row — and redoing them behind all the rows
        // measured 7 % more VALU instructions, not fewer: 1.81 G against 1.68 G per launch of config 3a)
        // ROT (round 4, last): the group starts ROT rows behind a multiple of PH rows — its row i has phase (i + ROT) mod PH, one cycle of table entries
        // (qstep) further on where that wraps.  With ROT = 1 the groups begin right behind a block's first (unclean) row: six groups instead of five
        // in a 2229-output block, five rows instead of ten left to the row-by-row code
        [[maybe_unused]] auto group = [&](unsigned rb, unsigned gq, auto rotc) {   // PH clean rows from output rb (64 (PH g + ROT)), table entries from gq on
            constexpr int ROT = decltype(rotc)::value;
            unsigned need = 0;
            OUT_T *const ob = obase + (rb + lane);      // one 64-bit address per group: the rows' stores are immediate offsets from it
#pragma unroll
            for (int i = 0; i < NPH; i++) {
                const int p = (i + ROT) % NPH;   // (a constant once the loop is unrolled)
                const float *tp = sm + (qo[p] + gq + ((i + ROT) >= NPH ? P.qstep : 0u));   // s1 = floor(x) - 1
                const float p1 = tp[0], p2 = tp[1];
                float v;
                if constexpr (INTERP == AUKIT_INTERP_LINEAR) v = __builtin_fmaf(p2 - p1, pw[p].x, p1);
                else {
                    const float p0 = tp[-1], p3 = tp[2];
                    v = __builtin_fmaf(pw[p].w, p3, __builtin_fmaf(pw[p].z, p2, __builtin_fmaf(pw[p].y, p1, pw[p].x * p0)));
                }
                asm volatile("" : "+v"(v));   // rows stay scalar: paired into v_pk_fma_f32 they cost a dozen v_mov per pair to line the operands up
                const float fl = floorf(v);
                const float fr = v - fl;
                const bool accept = fr > TIER1_GUARD && fr < 1 - TIER1_GUARD;
                need |= accept ? 0u : 1u << i;
                ob[64 * i] = (OUT_T)(int)__builtin_amdgcn_fmed3f(fl, -128.0f, 127.0f);
            }
            if (__any(need != 0)) {
                // (round 4, late) the turned-down outputs go on the wave's list (their index j, in the table's spare slots) and tiers 2 - 3 take them
                // together, a lane each, when the list is full and behind the block's rows: redone here — one pass per row with a turn-down, one or
                // two lanes alive in it — they were a twelfth of the kernel's instructions for a five-hundredth of its outputs
#pragma unroll
                for (int i = 0; i < NPH; i++) {
                    const bool nd = (need >> i) & 1u;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(nd);
                    if (m) {   // (wave-uniform)
                        const unsigned c = (unsigned)__builtin_popcountll(m);
                        if (dcnt + c > IMA_DL_CAP) flush_deferred();
                        if (c > IMA_DL_CAP) {   // (more than the list holds in one row — silence, garbage: in place)
                            if (nd) {
                                const unsigned j = rb + 64 * i + lane, n = j * P.fa;
                                const unsigned q = __umulhi(n, P.fmagic);
                                const float fl = slow((int)q + 1, n - q * P.fb, j, true);
                                obase[j] = (OUT_T)(int)__builtin_amdgcn_fmed3f(fl, -128.0f, 127.0f);
                            }
                        } else {
                            if (nd) dl[dcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = rb + 64 * i + lane;
                            dcnt += c;
                        }
                    }
                }
            }
        };
        const bool rot1 = PH > 1 && mid_lo == 64;   // (wave-uniform) the groups begin at the block's second row
        unsigned gq = 0, gnext = rot1 ? 64u : 0u;   // the next group boundary (output index) and its table offset
        for (unsigned rb = 0; rb < newlen;) {
            if constexpr (PH > 0) {
                if (rb == gnext) {
                    if (rb >= mid_lo && rb + 64 * PH <= mid_end) {
                        if (rot1) group(rb, gq, std::integral_constant<int, 1>{}); else group(rb, gq, std::integral_constant<int, 0>{});
                        rb += 64 * PH; q0 += P.qstep;          // 64 PH fa = qstep fb exactly: rem stays
                        gnext = rb; gq += P.qstep;
                        continue;
                    }
                    gnext += 64 * PH; gq += P.qstep;
                }
            }
            if (rb >= mid_lo && rb + 64 <= mid_end) row(rb, std::true_type{}); else row(rb, std::false_type{});
            rb += 64; q0 += P.fdq; rem += P.fdr;
        }
        flush_deferred();
        __builtin_amdgcn_wave_barrier();  // the next block's decode overwrites the table
    }
    if constexpr (AUDIT) {
        for (int o = 32; o; o >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, o)); acnt += __shfl_xor(acnt, o); }
        if (lane == 0 && P.audit) { atomicMax(&P.audit[0], __float_as_uint(amax)); atomicAdd(&P.audit[1], acnt); }
    }
}

static int ima_decode_audio(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, double new_rate, int interp, bool do_resample,
                            int dtype, aukit_audio **out, const uint64_t *nibble_counts = nullptr) {
    const int C = d->channels;
    const bool wav = d->codec == AUKIT_CODEC_ADPCM_WAV;
    if (C < 1) return fail(AUKIT_E_ARG, "bad argument #2 (number outside of range)");
    if (d->sample_rate < 1) return fail(AUKIT_E_ARG, "bad argument #3 (number outside of range)");
    if (C > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "at most %d channels are supported", AUKIT_MAX_PLANAR_CHANNELS);
    if (wav) {
        if (C > 2) return fail(AUKIT_E_UNSUPPORTED, "the WAV IMA splitter handles 1 or 2 channels (aukit.lua:1512-1546)");
        if (d->block_align <= 4 * C || (d->block_align - 4 * C) % (4 * C) != 0) return fail(AUKIT_E_ARG, "bad blockAlign");
    } else {
        for (int c = 0; c < C; c++) {
            if (d->predictor[c] < -32768 || d->predictor[c] > 32767) return fail(AUKIT_E_ARG, "bad argument #6 (number outside of range)");
            if (d->step_index[c] < 0 || d->step_index[c] > 88) return fail(AUKIT_E_ARG, "bad argument #7 (number outside of range)");
        }
    }
    std::vector<ImaRowJob> jobs((size_t)in->n * C);
    std::vector<uint64_t> row_off((size_t)in->n * C), row_len((size_t)in->n * C);
    uint64_t tot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        uint64_t L;
        if (wav) {
            if (nb == 0) return fail(AUKIT_E_LUA, "attempt to index a nil value (field '?')");  // blocks[1]:concat
            const uint64_t ba = (uint64_t)d->block_align, full = nb / ba, rem = nb % ba;
            L = full * ((ba - 4ull * C) * 2 / C);
            if (rem) {
                if (C == 2) return fail(AUKIT_E_LUA, "bad argument #1 to 'band' (number expected, got nil)");  // partial stereo block :1516
                if (rem < 3) return fail(AUKIT_E_LUA, "data string too short");
                L += rem > 4 ? (rem - 4) * 2 : 0;
            }
        } else L = nibble_counts ? nibble_counts[s] / (uint64_t)C : nb * 2 / (uint64_t)C;  // :1231 (a table of nibbles: `len = #data / channels`, :1237)
        const uint64_t stride = round_up(std::max<uint64_t>(L, 1), 8);
        for (int c = 0; c < C; c++) {
            ImaRowJob &j = jobs[(size_t)s * C + c];
            j.src_off = in->off[s]; j.nbytes = nb; j.out_off = tot; j.nib_total = L; j.c = c; j.pad = 0;
            row_off[(size_t)s * C + c] = tot; row_len[(size_t)s * C + c] = L;
            tot += stride;
        }
    }
    int rc = ctx->tmp_buf.ensure((size_t)tot * 2 + 64);
    if (rc) return rc;
    const size_t jbytes = jobs.size() * sizeof(ImaRowJob);
    if ((rc = ctx->tmp_buf2.ensure(jbytes + 16))) return rc;
    if (jbytes) { int hrc = h2d_table(ctx, ctx->tmp_buf2.p, jobs.data(), jbytes); if (hrc) return hrc; }
    int *err = reinterpret_cast<int *>(reinterpret_cast<char *>(ctx->tmp_buf2.p) + jbytes);
    AUKIT_HIP_CHECK(hipMemsetAsync(err, 0, 8, ctx->stream));
    if (!jobs.empty()) {
        ImaParams P{};
        P.src = in->data(); P.jobs = reinterpret_cast<const ImaRowJob *>(ctx->tmp_buf2.p); P.njobs = (unsigned)jobs.size();
        P.C = C; P.block_align = d->block_align; P.mode = wav ? 0 : 1; P.top_first = d->top_first; P.interleaved = d->interleaved;
        for (int c = 0; c < C; c++) { P.init_pred[c] = d->predictor[c]; P.init_idx[c] = d->step_index[c]; }
        P.mask_mono_index = 1;
        P.out = reinterpret_cast<short *>(ctx->tmp_buf.p);
        P.err = err;
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        uint64_t max_blocks = 0;
        if (wav) for (uint32_t s = 0; s < in->n; s++) max_blocks = std::max<uint64_t>(max_blocks, (in->off[s + 1] - in->off[s] + (uint64_t)d->block_align - 1) / (uint64_t)d->block_align);
        if (wav && C == 1 && max_blocks >= 1 && !getenv("AUKIT_IMA_ROWS_WAVE") && !getenv("AUKIT_IMA_ROWS_SERIAL")) {   // a lane per block (round 6)
            std::vector<uint64_t> blk0(jobs.size() + 1, 0);
            for (size_t j = 0; j < jobs.size(); j++) blk0[j + 1] = blk0[j] + (jobs[j].nbytes + (uint64_t)d->block_align - 1) / (uint64_t)d->block_align;
            const size_t nwaves = (size_t)((blk0.back() + 63) / 64);
            std::vector<uint64_t> tab(blk0);
            tab.resize(blk0.size() + (nwaves + 1) / 2, 0);
            unsigned *wj = reinterpret_cast<unsigned *>(tab.data() + blk0.size());
            for (size_t w = 0, j = 0; w < nwaves; w++) { while (j + 1 < jobs.size() && blk0[j + 1] <= 64ull * w) j++; wj[w] = (unsigned)j; }
            if ((rc = upload_table(ctx, ctx->misc_buf, tab.data(), tab.size() * 8))) return rc;
            ImaLaneParams Q{};
            Q.src = P.src; Q.jobs = P.jobs; Q.blk0 = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p); Q.njobs = P.njobs; Q.nblocks = blk0.back();
            Q.wave_j0 = reinterpret_cast<const unsigned *>(Q.blk0 + blk0.size());
            Q.block_align = d->block_align; Q.mask_mono_index = P.mask_mono_index; Q.out = P.out; Q.err = P.err;
            if (Q.nblocks) hipLaunchKernelGGL(k_ima_lanes, dim3((unsigned)((Q.nblocks + 63) / 64)), dim3(64), 0, ctx->stream, Q);
        } else
        if (wav && max_blocks > 1 && (uint64_t)jobs.size() * max_blocks < (1ull << 31) && !getenv("AUKIT_IMA_ROWS_SERIAL"))
            hipLaunchKernelGGL(k_ima_rows_blocks, dim3((unsigned)(jobs.size() * (C == 1 ? (max_blocks + AUKIT_IMA_BPW - 1) / AUKIT_IMA_BPW : max_blocks))), dim3(64), 0, ctx->stream, P, (unsigned)max_blocks);
        else hipLaunchKernelGGL(k_ima_rows, dim3((unsigned)jobs.size()), dim3(64), 0, ctx->stream, P);
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_ima_rows", in->total() + tot * 2))) return rc;
        if (!(wav && C == 1)) {   // (aukit.wav masks a one-channel block's step index with 0x0F, :1544: it cannot leave 0 .. 88 — no flag to wait for)
            int herr = 0;
            AUKIT_HIP_CHECK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
            AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (herr) return fail(AUKIT_E_ARG, "bad argument #7 (number outside of range)");
        }
    }
    return audio_from_int_rows(ctx, SRC_I16, ctx->tmp_buf.p, row_off, row_len, in->n, C, d->sample_rate, new_rate, interp, do_resample, dtype, 32767, 32768, out);
}

template <typename OUT_T>
static int launch_ima_stream(aukit_ctx *ctx, int interp, const ImaStreamParams &P, unsigned grid, size_t lds, unsigned threads) {
    // (blocks beyond 64 KiB of decoded samples — blockAlign x channels above ~4 KiB — take the CU's whole LDS: the kernel is told so, round 6)
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) AUKIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, ctx->stream, P);
        AUKIT_HIP_CHECK(hipGetLastError());
        return AUKIT_OK;
    };
    switch (interp) {
    case AUKIT_INTERP_NONE: return go(k_ima_stream<AUKIT_INTERP_NONE, OUT_T>);
    case AUKIT_INTERP_LINEAR: return go(k_ima_stream<AUKIT_INTERP_LINEAR, OUT_T>);
    case AUKIT_INTERP_CUBIC: return go(k_ima_stream<AUKIT_INTERP_CUBIC, OUT_T>);
    default: return go(k_ima_stream<AUKIT_INTERP_SINC, OUT_T>);
    }
}

// aukit.stream.adpcm(input, blockAlign, channels, sampleRate, mono)  aukit.lua:2753-2835
// First block of every stream whose header carries a step index above 88 and that decodes at least one word: the reference
// indexes ima_step_table with it unmasked (:2799, :2807) and the iterator call that reaches the block dies with "attempt to perform
// arithmetic on a nil value" — the chunks of earlier calls were delivered, that call's is not.  One wave per stream.
__global__ __launch_bounds__(64) void k_ima_scan_headers(const unsigned char *src, const unsigned long long *off, unsigned n, int C, unsigned long long ba,
                                                        unsigned *first_bad) {
    const unsigned s = blockIdx.x;
    if (s >= n) return;
    const unsigned long long nbytes = off[s + 1] - off[s];
    const unsigned long long nblk = nbytes >= 4ull * C + 1 ? (nbytes - 4ull * C - 1) / ba + 1 : 0;
    unsigned best = 0xFFFFFFFFu;
    for (unsigned long long bi = threadIdx.x; bi < nblk && bi < 0xFFFFFFFFull; bi += 64) {
        const unsigned long long b0 = bi * ba, rem = nbytes - b0;
        const long long ng = (long long)((rem - 1) / (4ull * C)) - 1;  // word groups decoded (:2800-2802), capped elsewhere; > 0 is all that matters here
        if (ng <= 0) continue;
        const unsigned char *blk = src + off[s] + b0;
        bool bad = false;
        for (int c = 0; c < C; c++) bad = bad || blk[4 * c + 2] > 88;
        if (bad) best = min(best, (unsigned)bi);
    }
    for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
    if (threadIdx.x == 0) first_bad[s] = best;
}

static int ima_stream(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out,
                      aukit_chunks **chunks_out) {
    const int C = d->channels;
    if (d->sample_rate < 1) return fail(AUKIT_E_ARG, "bad argument #4 (number outside of range)");
    if (C < 1 || C > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_ARG, "channels out of range");
    if (d->block_align <= 4 * C || (d->block_align - 4 * C) % (4 * C) != 0) return fail(AUKIT_E_UNSUPPORTED, "blockAlign must be 4*channels*(k+1)");
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "invalid interpolation");
    if (dtype != AUKIT_I8 && dtype != AUKIT_F64) return fail(AUKIT_E_ARG, "stream.adpcm output must be AUKIT_I8 or AUKIT_F64");
    const uint64_t ba = (uint64_t)d->block_align;
    const double ratio = 48000 / d->sample_rate;                                   // :2761
    const double spb = (double)(ba - 4ull * C) * 2 / C;                            // :2765
    const double iterPerSecond = std::ceil(d->sample_rate / spb);                  // :2766
    const double bytesPerSecond = (double)ba * iterPerSecond;                      // :2767
    const uint32_t newlen_full = (uint32_t)std::floor(spb * ratio);                // :2768
    const uint64_t ips = (uint64_t)iterPerSecond;
    const int nd = mono ? 1 : C;
    static const bool TT = getenv("AUKIT_HOST_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) { if (TT) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[ima host] %-12s %7.1f us\n", w, std::chrono::duration<double, std::micro>(t - T0).count()); T0 = t; } };
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0);
    ck->status.assign(in->n, 0);
    ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0), blk0(in->n + 1, 0);
    std::vector<unsigned> first_bad(in->n, 0xFFFFFFFFu);
    hipStream_t scan_stream = ctx->stream;
    if (in->n) {  // which streams die on a header index above 88, and where (the answer is awaited after the plans below were made)
        // (round 6, late: on the look-ahead stream — the scan reads the input batch and nothing else, and its host wait was a wait for everything the call
        // BEFORE had left on ctx->stream: back-to-back calls now plan while the kernel of the call before runs.  AUKIT_IMA_SCAN_MAIN=1: as before)
        if (!getenv("AUKIT_IMA_SCAN_MAIN")) {
            int prc = ctx_pre_stream(ctx, &scan_stream);
            if (prc) { delete ck; return prc; }
            if (in->ready && hipStreamWaitEvent(scan_stream, in->ready, 0) != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "hipStreamWaitEvent failed"); }
        }
        DevBuf &sb = scan_stream == ctx->stream ? ctx->misc_buf : ctx->scan_buf;
        int rc0 = sb.ensure((size_t)in->n * 4 + 16);
        if (rc0) { delete ck; return rc0; }
        unsigned *dfb = reinterpret_cast<unsigned *>(sb.p);
        hipLaunchKernelGGL(k_ima_scan_headers, dim3(in->n), dim3(64), 0, scan_stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off), in->n, C,
                           (unsigned long long)ba, dfb);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(first_bad.data(), dfb, (size_t)in->n * 4, hipMemcpyDeviceToHost, scan_stream) != hipSuccess) {
            delete ck; return fail(AUKIT_E_HIP, "stream.adpcm header scan failed");
        }
    }
    // The iterator calls of one stream depend on its byte count and on where a bad header stops it: one plan per distinct pair.
    struct Plan { uint64_t blocks = 0, total = 0; std::vector<uint32_t> len; std::vector<double> pos; };
    auto make_plan = [&](uint64_t nb, uint64_t calls_ok) {
        Plan pl;
        // blocks processed: while n + 4C <= #data (1-based n)  :2794
        const uint64_t nblk = nb >= 4ull * C + 1 ? (nb - 4ull * C - 1) / ba + 1 : 0;
        pl.blocks = calls_ok == ~0ull ? nblk : std::min<uint64_t>(nblk, calls_ok * ips);
        uint64_t done = 0;
        for (uint64_t call = 0;; call++) {  // one iterator call = up to iterPerSecond blocks
            if (call == calls_ok) break;
            const uint64_t take = std::min<uint64_t>(ips, nblk - done);
            // every block but the stream's last sees at least blockAlign + 4C more bytes — all its word groups plus the junk word (Q6) —
            // and yields newlen_full outputs; only the last block needs the arithmetic of :2800-2802 / :2817
            uint64_t full = take, produced = 0;
            if (take && done + take == nblk) {
                full = take - 1;
                const uint64_t rem = nb - (nblk - 1) * ba;
                long long ng = (long long)((rem - 1) / (4ull * C)) - 1;
                const long long nwr = (long long)((ba - 4ull * C) / (4ull * C));
                if (ng > nwr + 1) ng = nwr + 1;
                if (ng < 0) ng = 0;
                const uint64_t nbn = (uint64_t)ng * 8;
                produced += ((double)nbn < spb) ? (uint64_t)std::floor((double)nbn * ratio) : newlen_full;
            }
            produced += full * newlen_full;
            // a short block is always the last one (n advances past #data), so block b's outputs start at b * newlen_full
            done += take;
            if (produced == 0) break;  // #retval[1] == 0 → nil
            pl.len.push_back((uint32_t)produced);
            pl.pos.push_back(((double)(done * ba + ctx->sb_bytes + 1)) / bytesPerSecond);  // (n + pos) / bytesPerSecond :2833 (sb_bytes: what a stream handle dropped)
            pl.total += produced;
            if (done >= nblk) break;   // the next call sees n + 4C > #data and returns nil
        }
        return pl;
    };
    std::map<std::pair<uint64_t, uint64_t>, Plan> plans;
    std::vector<const Plan *> plan_of(in->n, nullptr);
    const Plan *last = nullptr;
    uint64_t last_nb = ~0ull;
    for (uint32_t s = 0; s < in->n; s++) {   // the good-header plans while the scan runs
        const uint64_t nb = in->off[s + 1] - in->off[s];
        if (nb != last_nb) {
            auto it = plans.find({nb, ~0ull});
            if (it == plans.end()) it = plans.emplace(std::make_pair(nb, ~0ull), make_plan(nb, ~0ull)).first;
            last = &it->second; last_nb = nb;
        }
        plan_of[s] = last;
    }
    if (in->n && hipStreamSynchronize(scan_stream) != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "stream.adpcm header scan failed"); }
    lap("scan+plans");
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        ck->length_seconds[s] = (double)(nb + ctx->sb_bytes) / (double)ba * spb / d->sample_rate;   // :2834
        if (first_bad[s] != 0xFFFFFFFFu) {
            // a bad header kills the iterator call that reaches it: only the calls before that one deliver
            const uint64_t calls_ok = (uint64_t)first_bad[s] / std::max<uint64_t>(ips, 1);
            ck->status[s] = AUKIT_E_LUA;  // "attempt to perform arithmetic on a nil value (field '?')"
            auto it = plans.find({nb, calls_ok});
            if (it == plans.end()) it = plans.emplace(std::make_pair(nb, calls_ok), make_plan(nb, calls_ok)).first;
            plan_of[s] = &it->second;
        }
        const Plan &pl = *plan_of[s];
        blk0[s + 1] = blk0[s] + pl.blocks;
        lens[s] = pl.total;
        ck->nchunks[s] = (uint32_t)pl.len.size();
        ck->max_chunks = std::max<uint32_t>(ck->max_chunks, ck->nchunks[s]);
    }
    const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
    ck->lens.assign((size_t)ck->n * mc, 0);
    ck->pos.assign((size_t)ck->n * mc, 0);
    for (uint32_t s = 0; s < in->n; s++) {
        const Plan &pl = *plan_of[s];
        if (!pl.len.empty()) {
            memcpy(&ck->lens[(size_t)s * mc], pl.len.data(), pl.len.size() * sizeof(uint32_t));
            memcpy(&ck->pos[(size_t)s * mc], pl.pos.data(), pl.pos.size() * sizeof(double));
        }
    }
    int rc;
    lap("plan");
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, nd, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    lap("prepare");
    const uint64_t nblocks = blk0[in->n];
    if (nblocks) {
        std::vector<uint64_t> tab(blk0);
        tab.insert(tab.end(), a->row_off.begin(), a->row_off.end());
        tab.insert(tab.end(), a->row_stride.begin(), a->row_stride.end());
        tab.push_back(0);
        if ((rc = upload_table(ctx, ctx->tmp_buf2, tab.data(), tab.size() * 8))) { delete ck; return rc; }
        const unsigned long long *t = reinterpret_cast<const unsigned long long *>(ctx->tmp_buf2.p);
        int *err = reinterpret_cast<int *>(const_cast<unsigned long long *>(t + tab.size() - 1));
        ImaStreamParams P{};
        P.src = in->data(); P.off = reinterpret_cast<const unsigned long long *>(in->d_off);
        P.blk0 = t; P.out_off = t + in->n + 1; P.out_stride = t + 2 * (size_t)in->n + 1;
        P.nstreams = in->n; P.nblocks = nblocks; P.C = C; P.block_align = d->block_align; P.mono = mono ? 1 : 0;
        P.newlen_full = newlen_full; P.ratio = ratio; P.rcp = 1.0 / ratio;
        P.exact_rcp = exact_div_verified(ctx, ratio, (uint64_t)newlen_full + 2) ? 1 : 0;
        P.cap = (int)((ba - 4ull * C) * 2 / C + 8 + 8);
        P.cap += P.cap / 16 + 2;  // skewed LDS slots
        P.out = a->dev; P.err = err;
        if (C == 1 && !ctx->exact_math && d->sample_rate == std::floor(d->sample_rate) && d->sample_rate <= 4e9) {
            unsigned long long x = 48000, y = (unsigned long long)d->sample_rate;
            while (y) { const unsigned long long tq = x % y; x = y; y = tq; }
            unsigned long long fa = (unsigned long long)d->sample_rate / x, fb = 48000 / x;  // x - 1 = (i - 1) / ratio = (i - 1) * fa / fb
            if (fb == 1) { fa *= 2; fb = 2; }  // a 48 kHz file: every position is an integer (rem is always 0, the weights of phase 0 are 0, 1, 0, 0)
            if (fb >= 2 && ((double)newlen_full * (double)fa + (double)fb) * (double)fb < 4294967296.0) {
                P.fast = 1; P.fa = (unsigned)fa; P.fb = (unsigned)fb; P.fmagic = (unsigned)((4294967296ull + fb - 1) / fb); P.inv_fb = 1.0 / (double)fb;
                P.fdq = (unsigned)((64ull * fa) / fb); P.fdr = (unsigned)((64ull * fa) % fb);
                P.mid_lo = (unsigned)(((((3ull * fb + fa - 1) / fa) + 63) / 64) * 64);
                P.qstep = 0;
                {   // a full block: 1024 table entries (the junk word included), newlen_full outputs
                    const unsigned long long jm = ((1024ull - 3) * fb) / fa, lim = std::min<unsigned long long>(jm, P.newlen_full);
                    P.mid_end_full = P.newlen_full >= 64 ? (unsigned)(lim & ~63ull) : 0u;
                }
            }
        }
        // one channel, at most 512 phases: the three-tier kernel on an f32 table (k_ima_stream_f32)
        if (P.fast && P.fb <= 512 && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC) && !getenv("AUKIT_IMA_F64")) {
            const int wf = interp == AUKIT_INTERP_CUBIC ? 4 : 1;
            std::vector<float> w((size_t)P.fb * wf);
            for (unsigned r = 0; r < P.fb; r++) {
                const long double f = (long double)r / (long double)P.fb, f2 = f * f, f3 = f2 * f;
                if (wf == 1) w[r] = (float)f;
                else {
                    w[4 * r] = (float)(-0.5L * f3 + f2 - 0.5L * f); w[4 * r + 1] = (float)(1.5L * f3 - 2.5L * f2 + 1.0L);
                    w[4 * r + 2] = (float)(-1.5L * f3 + 2.0L * f2 + 0.5L * f); w[4 * r + 3] = (float)(0.5L * f3 - 0.5L * f2);
                }
            }
            if ((rc = upload_table(ctx, ctx->tmp_buf3, w.data(), w.size() * 4))) { delete ck; return rc; }
            P.bps = 0;
            if (in->n) {
                const uint64_t b1 = blk0[1] - blk0[0];
                bool uni = b1 > 0;
                for (uint32_t s = 1; s < in->n && uni; s++) uni = blk0[s + 1] - blk0[s] == b1;
                if (uni) P.bps = b1;
            }
            const int capf = ((int)((ba - 4ull) * 2 + 8 + 8 + 15) & ~15) + (int)IMA_DL_CAP;  // floats per wave: the block's nibbles + the junk word's + slack, unskewed, + the deferred-output list
            const size_t lds = ((((size_t)P.fb * wf + 3) & ~(size_t)3) + 96) * 4 + (size_t)capf * 4 * 4;
            if (lds <= 64 * 1024) {
                P.cap = capf;
                const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
                unsigned gmul = 8;   // 1 / 2 / 4 / 8 / 16 / 32 measured 2.98 / 2.83 / 2.74 / 2.71 / 2.72 / 2.74 ms on config 3a
                if (const char *e = getenv("AUKIT_IMA_GRID_MUL")) { const int v = atoi(e); if (v >= 1) gmul = (unsigned)v; }   // tuning knob
                const unsigned grid = (unsigned)std::min<uint64_t>((nblocks + 3) / 4, (uint64_t)ctx->num_cus * per_cu * gmul);
                lap("tables");
                if ((rc = ctx_begin_kernel(ctx))) { delete ck; return rc; }
                const float *wgp = reinterpret_cast<const float *>(ctx->tmp_buf3.p);
                // the phases in registers when a lane meets 3 or 5 of them (fb = 3 · 2^i or 5 · 2^i, i <= 6: 22 050 and 44 100 Hz have 5)
                unsigned g64 = P.fb, h64 = 64;
                while (h64) { const unsigned r = g64 % h64; g64 = h64; h64 = r; }
                unsigned ph = P.fb / g64;
                const char *er = getenv("AUKIT_IMA_REGS");
                if ((ph != 3 && ph != 5) || (er && atoi(er) == 0)) ph = 0;
                P.audit = nullptr;
                if (ctx->collect_stats) {   // the audited instantiation (floor_wave.hip): tier 1's values against tier 2's, every output
                    if ((rc = ctx->fmt_flag.ensure(64))) { delete ck; return rc; }
                    P.audit = reinterpret_cast<unsigned *>(ctx->fmt_flag.p) + 8;
                    AUKIT_HIP_CHECK(hipMemsetAsync(P.audit, 0, 8, ctx->stream));
                    ph = 0;
                }
                P.qstep = ph ? (unsigned)((64ull * ph * P.fa) / P.fb) : 0u;
#define AUKIT_IMA_F32(I, T) do { if (P.audit) hipLaunchKernelGGL((k_ima_stream_f32<I, T, 0, true>), dim3(grid), dim3(256), lds, ctx->stream, P, wgp); \
                                 else if (ph == 5) hipLaunchKernelGGL((k_ima_stream_f32<I, T, 5>), dim3(grid), dim3(256), lds, ctx->stream, P, wgp); \
                                 else if (ph == 3) hipLaunchKernelGGL((k_ima_stream_f32<I, T, 3>), dim3(grid), dim3(256), lds, ctx->stream, P, wgp); \
                                 else hipLaunchKernelGGL((k_ima_stream_f32<I, T, 0>), dim3(grid), dim3(256), lds, ctx->stream, P, wgp); } while (0)
                if (dtype == AUKIT_I8) { if (interp == AUKIT_INTERP_LINEAR) AUKIT_IMA_F32(AUKIT_INTERP_LINEAR, signed char); else AUKIT_IMA_F32(AUKIT_INTERP_CUBIC, signed char); }
                else { if (interp == AUKIT_INTERP_LINEAR) AUKIT_IMA_F32(AUKIT_INTERP_LINEAR, double); else AUKIT_IMA_F32(AUKIT_INTERP_CUBIC, double); }
#undef AUKIT_IMA_F32
                if (hipGetLastError() != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "k_ima_stream_f32 launch failed"); }
                uint64_t out_elems = 0;
                for (uint64_t l : lens) out_elems += l * nd;
                if ((rc = ctx_end_kernel(ctx, "k_ima_stream_f32", in->total() + out_elems * dtype_size(dtype)))) { delete ck; return rc; }
                // (the plans end in front of the first block whose header index is above 88 — k_ima_scan_headers —: the kernel's own flag for such a block cannot
                // be raised by a planned block.  Waiting for it kept the host behind every call's kernel; AUKIT_IMA_ASSERT=1 waits and looks, as do the audited runs)
                int herr = 0;
                if (P.audit || getenv("AUKIT_IMA_ASSERT")) {
                    AUKIT_HIP_CHECK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
                    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                }
                lap("kernel+sync");
                if (P.audit) {
                    unsigned h[2] = {0, 0};
                    AUKIT_HIP_CHECK(hipMemcpy(h, P.audit, 8, hipMemcpyDeviceToHost));
                    float e; memcpy(&e, &h[0], 4);
                    ctx->counters[AUKIT_COUNTER_TIER1_ERR_NANO] = (uint64_t)std::llround((double)e * 1e9);
                    ctx->counters[AUKIT_COUNTER_TIER1_OUTPUTS] = h[1];
                }
                if (herr) { delete ck; return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')"); }  // ima_step_table[idx > 88]
                if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
                lap("chunks out");
                return AUKIT_OK;
            }
        }
        unsigned nwv = 4;
        while (nwv > 1 && (size_t)P.cap * C * 8 * nwv > 64 * 1024) nwv >>= 1;
        const size_t lds = (size_t)P.cap * C * 8 * nwv;
        // one wave per block and the CU's whole 160 KiB (less what the kernel declares itself): blockAlign x channels up to ~10 KiB — every WAV
        // encoder's blocks; the reference has no limit at all, a block-at-a-time path for more than that does not exist here
        if (lds > 156 * 1024) { delete ck; return fail(AUKIT_E_UNSUPPORTED, "blockAlign %d with %d channels needs more than 156 KiB of LDS", d->block_align, C); }
        unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(32 / nwv, (160 * 1024) / lds));
        unsigned grid = (unsigned)std::min<uint64_t>((nblocks + nwv - 1) / nwv, (uint64_t)ctx->num_cus * per_cu * 2);
        if ((rc = ctx_begin_kernel(ctx))) { delete ck; return rc; }
        rc = dtype == AUKIT_I8 ? launch_ima_stream<signed char>(ctx, interp, P, grid, lds, nwv * 64) : launch_ima_stream<double>(ctx, interp, P, grid, lds, nwv * 64);
        if (rc) { delete ck; return rc; }
        uint64_t out_elems = 0;
        for (uint64_t l : lens) out_elems += l * nd;
        if ((rc = ctx_end_kernel(ctx, "k_ima_stream", in->total() + out_elems * dtype_size(dtype)))) { delete ck; return rc; }
        int herr = 0;
        AUKIT_HIP_CHECK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (herr) { delete ck; return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')"); }  // ima_step_table[idx > 88]
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

// ================================================================= dispatch
int decode_msadpcm_audio(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, double, int, bool, int, aukit_audio **);
int decode_qoa_audio(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, double, int, bool, int, aukit_audio **);
int decode_flac_audio(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, double, int, bool, int, aukit_audio **);
int decode_mdfpwm_audio(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, double, int, bool, int, aukit_audio **);

int decode_block_codec(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, double new_rate, int interp, bool do_resample, int dtype,
                       aukit_audio **out) {
    switch (d->codec) {
    case AUKIT_CODEC_DFPWM: return dfpwm_decode_audio(ctx, in, d, new_rate, interp, do_resample, dtype, out);
    case AUKIT_CODEC_ADPCM:
    case AUKIT_CODEC_ADPCM_WAV: return ima_decode_audio(ctx, in, d, new_rate, interp, do_resample, dtype, out);
    case AUKIT_CODEC_MSADPCM: return decode_msadpcm_audio(ctx, in, d, new_rate, interp, do_resample, dtype, out);
    case AUKIT_CODEC_QOA: return decode_qoa_audio(ctx, in, d, new_rate, interp, do_resample, dtype, out);
    case AUKIT_CODEC_FLAC: return decode_flac_audio(ctx, in, d, new_rate, interp, do_resample, dtype, out);
    case AUKIT_CODEC_MDFPWM: return decode_mdfpwm_audio(ctx, in, d, new_rate, interp, do_resample, dtype, out);
    }
    return fail(AUKIT_E_ARG, "unknown codec %d", d->codec);
}

int stream_more_codecs(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, int, int, int, aukit_audio **, aukit_chunks **);

int stream_block_codec(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out,
                       aukit_chunks **chunks) {
    if (d->codec == AUKIT_CODEC_ADPCM_WAV) return ima_stream(ctx, in, d, interp, mono, dtype, out, chunks);
    return stream_more_codecs(ctx, in, d, interp, mono, dtype, out, chunks);
}

}  // namespace aukit

using namespace aukit;

extern "C" {

// aukit.adpcm(data, ...) with `data` a TABLE of nibbles (aukit.lua:1183-1184, :1232-1238: `read()` hands out data[pos], len = #data / channels)
int aukit_decode_nibbles(aukit_ctx *ctx, const uint8_t *nibbles, const uint64_t *offsets, uint32_t n, const aukit_codec_desc *d, int dtype, aukit_audio **out) {
    if (!ctx || !d || !out || (n && (!offsets || (offsets[n] && !nibbles)))) return fail(AUKIT_E_ARG, "null argument");
    if (d->codec != AUKIT_CODEC_ADPCM) return fail(AUKIT_E_UNSUPPORTED, "table of nibbles: aukit.adpcm only");
    if (d->channels < 1) return fail(AUKIT_E_ARG, "bad argument #2 (number outside of range)");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<uint64_t> boff((size_t)n + 1, 0), counts(n, 0);
    std::vector<uint8_t> packed;
    for (uint32_t s = 0; s < n; s++) {
        if (offsets[s + 1] < offsets[s]) return fail(AUKIT_E_ARG, "offsets must not decrease");
        const uint64_t cnt = offsets[s + 1] - offsets[s];
        counts[s] = cnt;   // `for i = 1, #data / channels`: whole rounds of `channels` nibbles (ima_decode_audio divides)
        for (uint64_t i = 0; i < cnt; i += 2) {
            const unsigned hi = nibbles[offsets[s] + i], lo = i + 1 < cnt ? nibbles[offsets[s] + i + 1] : 0;
            if (hi > 15 || lo > 15) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");  // ima_index_table[nibble]
            packed.push_back((uint8_t)(hi << 4 | lo));
        }
        boff[s + 1] = packed.size();
    }
    aukit_codec_desc dd = *d;
    dd.top_first = 1;   // the packing above
    aukit_batch *b = nullptr;
    static const uint8_t none = 0;
    int rc = aukit_batch_upload(ctx, &b, packed.empty() ? &none : packed.data(), boff.data(), n);
    if (rc) return rc;
    rc = ima_decode_audio(ctx, b, &dd, 0, 0, false, dtype, out, counts.data());
    if (!rc) rc = aukit_ctx_sync(ctx);
    aukit_batch_free(b);
    return rc;
}

int aukit_dfpwm_encode(aukit_ctx *ctx, const aukit_audio *in, int interleaved, aukit_batch **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    if (in->dtype != AUKIT_F64 && in->dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "dfpwm encode needs a float audio");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    AUKIT_FLUSH(ctx, in);
    std::vector<uint64_t> off(in->n + 1, 0);
    for (uint32_t s = 0; s < in->n; s++) off[s + 1] = off[s] + (in->len[s] * (uint64_t)in->channels + 7) / 8;
    aukit_batch *b = *out;
    if (b && (!b->own || b->cap < off[in->n] + 128)) { aukit_batch_free(b); b = nullptr; }
    if (!b) {
        b = new aukit_batch();
        b->front_pad = 64;
        b->cap = (size_t)off[in->n] + 128;
        b->own = true;
        hipError_t e = hipMalloc((void **)&b->base, b->cap);
        if (e != hipSuccess) { delete b; return fail(AUKIT_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    }
    b->n = in->n;
    b->off = off;
    if (b->d_off) (void)hipFree(b->d_off);
    b->d_off = nullptr;
    AUKIT_HIP_CHECK(hipMalloc((void **)&b->d_off, ((size_t)in->n + 2) * 8));
    { int hrc = h2d_table(ctx, b->d_off, off.data(), ((size_t)in->n + 1) * 8); if (hrc) return hrc; }
    int *err = reinterpret_cast<int *>(b->d_off + in->n + 1);
    AUKIT_HIP_CHECK(hipMemsetAsync(err, 0, 8, ctx->stream));
    b->version++;
    *out = b;
    if (in->n == 0) return AUKIT_OK;
    // int8 rows in encoding order (16-byte aligned), their offsets and sample counts
    std::vector<uint64_t> tab((size_t)in->n * 2);
    uint64_t qtot = 0, maxtot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t tot = in->len[s] * (uint64_t)in->channels;
        tab[s] = qtot; tab[in->n + s] = tot;
        qtot += round_up(tot, 16) + 16;
        maxtot = std::max(maxtot, tot);
    }
    int rc = ctx->tmp_buf.ensure((size_t)qtot + 64);
    if (rc) return rc;
    if ((rc = upload_table(ctx, ctx->misc_buf, tab.data(), tab.size() * 8))) return rc;
    const unsigned long long *t = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
    signed char *q = reinterpret_cast<signed char *>(ctx->tmp_buf.p);
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    const unsigned long long *m = reinterpret_cast<const unsigned long long *>(in->d_meta);
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(((in->channels == 1 ? maxtot / 16 : maxtot) + 255) / 256, 4096)), in->n);
    if (in->dtype == AUKIT_F64)
        hipLaunchKernelGGL((k_dfpwm_quantize<double>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const double *>(in->dev), m, m + in->n, m + 2 * (size_t)in->n,
                           in->channels, interleaved, q, t, err);
    else
        hipLaunchKernelGGL((k_dfpwm_quantize<float>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const float *>(in->dev), m, m + in->n, m + 2 * (size_t)in->n,
                           in->channels, interleaved, q, t, err);
    AUKIT_HIP_CHECK(hipGetLastError());
    int erc = AUKIT_OK;
    // every stream cut into time chunks that a lane each encodes from a guessed state, verified afterwards (dfpwm_spec.hip): a batch of any size
    // fills the chip — the lane-per-stream encoder takes 27 ns per sample whatever the batch, the candidate search of k_dfe_* 500 x the work
    bool spec = false;
    if ((rc = dfpwm_encode_spec(ctx, q, tab.data(), tab.data() + in->n, in->n, b->data(), reinterpret_cast<const unsigned long long *>(b->d_off), off.data(), &spec))) return rc;
    const bool small = !spec && dfpwm_encode_i8_small(ctx, q, tab.data(), tab.data() + in->n, in->n, b->data(), off.data(), &erc);
    if (small && erc) return erc;
    if (!spec && !small && (rc = dfpwm_encode_i8(ctx, q, t, t + in->n, in->n, b->data(), reinterpret_cast<const unsigned long long *>(b->d_off)))) return rc;
    uint64_t elems = 0;
    for (uint64_t l : in->len) elems += l * in->channels;
    if ((rc = ctx_end_kernel(ctx, spec ? "k_dfpwm_quantize+k_dfx_chunks<rows>" : (small ? "k_dfpwm_quantize+k_dfe_*" : "k_dfpwm_quantize+k_dfpwm_encode_i8"), elems * dtype_size(in->dtype) + off[in->n]))) return rc;
    int herr = 0;
    AUKIT_HIP_CHECK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (herr) return fail(AUKIT_E_LUA, "Amplitude was outside the range -128..127 (cc.audio.dfpwm encoder)");
    return AUKIT_OK;
}

int aukit_dfpwm_transcode_mono(aukit_ctx *ctx, const aukit_batch *in, int channels, aukit_batch **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    if (channels < 1) return fail(AUKIT_E_ARG, "bad argument #2 (number outside of range)");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<uint64_t> off(in->n + 1, 0);
    uint64_t samples_total = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t samples = dfpwm_fed_bytes(in->off[s + 1] - in->off[s]) * 8;
        if (samples % (uint64_t)channels != 0) return fail(AUKIT_E_ARG, "bad argument #1 (uneven amount of data per channel)");
        off[s + 1] = off[s] + (samples / channels + 7) / 8;
        samples_total += samples / channels;
    }
    aukit_batch *b = *out;
    if (b && (!b->own || b->cap < off[in->n] + 128)) { aukit_batch_free(b); b = nullptr; }
    if (!b) {
        b = new aukit_batch();
        b->front_pad = 64;
        b->cap = (size_t)off[in->n] + 128;
        b->own = true;
        hipError_t e = hipMalloc((void **)&b->base, b->cap);
        if (e != hipSuccess) { delete b; return fail(AUKIT_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    }
    const bool same = b->n == in->n && b->off == off && b->d_off;
    b->n = in->n;
    if (!same) {
        b->off = off;
        if (b->d_off) (void)hipFree(b->d_off);
        b->d_off = nullptr;
        AUKIT_HIP_CHECK(hipMalloc((void **)&b->d_off, ((size_t)in->n + 2) * 8));
        { int hrc = h2d_table(ctx, b->d_off, off.data(), ((size_t)in->n + 1) * 8); if (hrc) return hrc; }
    }
    b->version++;
    *out = b;
    if (in->n == 0) return AUKIT_OK;
    int rc = ctx_begin_kernel(ctx);
    if (rc) return rc;
    bool aligned = (reinterpret_cast<uintptr_t>(in->data()) & 15) == 0;
    for (uint32_t s = 0; s < in->n && aligned; s++) aligned = (in->off[s] & 15) == 0;
    const bool stereo = channels == 2 && aligned && !getenv("AUKIT_DFPWM_GENERIC");
    if (channels == 2) {
        // every lane decodes, mixes and encodes its own time chunk of a stream; verified and patched afterwards (dfpwm_spec.hip)
        bool spec = false;
        if ((rc = dfpwm_transcode_spec(ctx, in, b->data(), reinterpret_cast<const unsigned long long *>(b->d_off), off.data(), &spec))) return rc;
        if (spec) return ctx_end_kernel(ctx, "k_dfx_chunks", in->total() + off[in->n]);
    }
    if (channels == 2) {  // decode + mono mix in parallel chunks (exact), then one encoder lane per stream (dfpwm_par.hip)
        std::vector<uint64_t> tab(2 * (size_t)in->n);
        uint64_t mtot = 0;
        for (uint32_t s = 0; s < in->n; s++) {
            const uint64_t pairs = dfpwm_fed_bytes(in->off[s + 1] - in->off[s]) * 4;
            tab[s] = mtot;
            tab[in->n + s] = pairs;
            mtot += round_up(std::max<uint64_t>(pairs, 1), 16);
        }
        if ((rc = ctx->tmp_buf.ensure((size_t)mtot + 64))) return rc;
        if ((rc = upload_table(ctx, ctx->misc_buf, tab.data(), tab.size() * 8))) return rc;
        const unsigned long long *t = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
        // large batches: decoder and encoder pipelined over time slices on two HIP streams (dfpwm_par.hip)
        const uint64_t lanes_full = (uint64_t)ctx->num_cus * 512;  // chunk lanes that fill the chip in mix mode
        // every slice must still give the decoder a chip-filling launch (lanes_full / n chunks per stream) of chunks long enough to
        // keep the 256-byte warm-up at 1/15 of their work
        uint64_t max_bytes = 0;
        for (uint32_t s = 0; s < in->n; s++) max_bytes = std::max<uint64_t>(max_bytes, in->off[s + 1] - in->off[s]);
        const uint64_t want = std::max<uint64_t>(1, lanes_full / in->n);
        int slices = in->n >= 2048 ? (int)std::min<uint64_t>(8, max_bytes / 1920 / want) : 1;  // measured on config 4: 1 / 4 / 8 slices = 27.0 / 25.7 / 24.4 ms per step
        if (const char *e = getenv("AUKIT_DFPWM_SLICES")) slices = atoi(e);
        // batches of up to one 64-stream group per CU: decoder and encoder in one persistent launch (AUKIT_DFPWM_FUSED=1: whatever the size, 0: never)
        const char *fe = getenv("AUKIT_DFPWM_FUSED");
        if (fe ? atoi(fe) != 0 : (in->n >= 2048 && !getenv("AUKIT_DFPWM_SLICES"))) {
            bool taken = false;
            rc = dfpwm_transcode_fused(ctx, in, reinterpret_cast<signed char *>(ctx->tmp_buf.p), t, t + in->n, b->data(), reinterpret_cast<const unsigned long long *>(b->d_off), off.data(), &taken);
            if (rc) return rc;
            if (taken) return ctx_end_kernel(ctx, "k_df_fused", in->total() + off[in->n]);
        }
        if (slices > 1) {
            bool taken = false;
            rc = dfpwm_transcode_sliced(ctx, in, reinterpret_cast<signed char *>(ctx->tmp_buf.p), t, t + in->n, b->data(), reinterpret_cast<const unsigned long long *>(b->d_off), off.data(), slices, &taken);
            if (rc) return rc;
            if (taken) return ctx_end_kernel(ctx, "k_df_chunks|k_dfpwm_encode_i8_slice", in->total() + off[in->n]);
        }
        int prc = AUKIT_OK;
        if (dfpwm_decode_parallel(ctx, in, 1, 2, reinterpret_cast<signed char *>(ctx->tmp_buf.p), t, nullptr, &prc)) {
            if (prc) return prc;
            int erc = AUKIT_OK;
            if (dfpwm_encode_i8_small(ctx, reinterpret_cast<const signed char *>(ctx->tmp_buf.p), tab.data(), tab.data() + in->n, in->n, b->data(), off.data(), &erc)) {
                if (erc) return erc;
                return ctx_end_kernel(ctx, "k_df_chunks+k_dfe_*", in->total() + off[in->n]);
            }
            if ((rc = dfpwm_encode_i8(ctx, reinterpret_cast<const signed char *>(ctx->tmp_buf.p), t, t + in->n, in->n, b->data(), reinterpret_cast<const unsigned long long *>(b->d_off)))) return rc;
            return ctx_end_kernel(ctx, "k_df_chunks+k_dfpwm_encode_i8", in->total() + off[in->n]);
        }
    }
    if (stereo)
        hipLaunchKernelGGL(k_dfpwm_transcode_stereo, dim3((in->n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off),
                           in->n, b->data(), reinterpret_cast<const unsigned long long *>(b->d_off));
    else
        hipLaunchKernelGGL(k_dfpwm_transcode_mono, dim3((in->n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off),
                           in->n, channels, b->data(), reinterpret_cast<const unsigned long long *>(b->d_off));
    AUKIT_HIP_CHECK(hipGetLastError());
    (void)samples_total;
    return ctx_end_kernel(ctx, stereo ? "k_dfpwm_transcode_stereo" : "k_dfpwm_transcode_mono", in->total() + off[in->n]);
}

}  // extern "C"
