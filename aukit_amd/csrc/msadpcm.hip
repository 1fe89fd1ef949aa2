// msadpcm.hip — MS-ADPCM: aukit.msadpcm (aukit.lua:1283-1353) and aukit.stream.msadpcm (aukit.lua:2588-2736).
//
// The recurrence has a floor inside (`math_floor((s1*c1 + s2*c2) / 256)`, :1321) and a `delta` that is multiplied and floored per
// sample (:1324), so a block is sequential: one lane per block, parallelism from blocks × streams.  What the wave kernel below does
// about it (k_ms_wave):
//   * a wave owns 64 consecutive blocks.  Their bytes arrive by 16-byte loads, 16 · RB/16 lanes per block side by side (coalesced over
//     the RB bytes a block contributes per round), one round ahead in registers, and are laid into LDS so that every lane then reads
//     its own block's bytes with conflict-free ds_read_b128;
//   * the recurrence runs in int32: every product has 24-bit operands (v_mul_i32_i24 / v_mad_i32_i24, full rate), `/ 256` of the Lua
//     is an arithmetic shift, the clamp one v_med3_i32.  That is the Lua's arithmetic as long as delta stays below 2^21; a lane whose
//     delta leaves that range (garbage input: the reference has no upper clamp, its doubles run to inf and nan) redoes the 16 bytes at
//     hand in fp64, op for op the Lua, and stays in fp64 for the rest of its block — decided per wave with one ballot, so encoder-made
//     input never executes that code;
//   * decoded predictors go to a per-wave LDS table as exact floats (a nan stays a nan), R samples per block and round behind a
//     four-sample tail of the round before, and leave the kernel coalesced along a block's samples:
//       - rows mode (aukit.msadpcm): normalised `p / (p < 0 and 32768 or 32767)` straight into the Audio's rows (F64 / F32), or raw
//         int16 rows when a resample follows (the PCM wave kernels read those as 16-bit mono strings);
//       - stream mode (aukit.stream.msadpcm): every block is resampled on its own (no history reaches across blocks for linear /
//         cubic: index 0 is nil, :2640-2643), `clamp(floor(v))` decided in the three tiers of k_ima_stream_f32 (codecs.hip) on that
//         table — no decoded sample ever touches HBM.
// Anything the wave kernel does not take (sinc, non-integer sample rates, > 512 output phases, coefficient tables whose products
// could leave 32 bits) runs the lane-per-block fp64 kernel of round 1/2 (k_msadpcm) and the generic resampler behind it.
#include <algorithm>
#include <type_traits>
#include "resample.h"
#include "resample_dev.h"

namespace aukit {

// tier 1's guard (k_ms_wave): its error is below 3e-4 (see tier1 below); 1e-3 until late round 3, 6e-4 keeps a factor of two (cf. floor_wave.hip, codecs.hip)
#ifndef AUKIT_MS_TIER1_GUARD
#define AUKIT_MS_TIER1_GUARD 6e-4f
#endif
constexpr float MS_TIER1_GUARD = AUKIT_MS_TIER1_GUARD;

int audio_from_int_rows(aukit_ctx *ctx, int src_kind, const void *rows_dev, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len,
                        uint32_t n, int channels, double rate, double new_rate, int interp, bool do_resample, int dtype, double norm_pos,
                        double norm_neg, aukit_audio **out);

static AUKIT_DEV double ms_lclamp(double n, double mn, double mx) { return n < mn ? mn : (n > mx ? mx : n); }

__constant__ int c_ms_adapt[16] = {230, 230, 230, 230, 307, 409, 512, 614, 768, 614, 512, 409, 307, 230, 230, 230};  // [0..7], [-8..-1]  :173-176

// ================================================================= generic fallback: one lane per block, fp64, rows of doubles
struct MsJob {
    unsigned long long blk_off;   // byte offset of the block
    unsigned long long hdr_off;   // byte offset of the header to use (mono: always the stream's first block, Q9)
    unsigned long long out_off;   // element offset (doubles) of this block's first sample, channel 0
    unsigned long long out_off_r; // channel 1
};
struct MsParams {
    const unsigned char *src;
    const MsJob *jobs;
    unsigned long long njobs;
    int C, block_align, ncoef;
    int coef1[32], coef2[32];
    double div_neg, div_pos;      // 32768/32767 (aukit.msadpcm) or 128/127 (stream.msadpcm)
    int floor_all;                // stream stereo: every sample floored (:2648-2662); stream mono / Audio path: not
    double *out;
    int *err;
};
static AUKIT_DEV double ms_step(double &s1, double &s2, double &delta, double c1, double c2, int nib, const int *adapt) {
    const double predictor = ms_lclamp(floor((s1 * c1 + s2 * c2) / 256) + nib * delta, -32768, 32767);  // :1321
    s2 = s1; s1 = predictor;
    const double nd = floor(adapt[nib & 15] * delta / 256);                                           // :1324
    delta = nd < 16 ? 16 : nd;  // math.max(nd, 16): PUC math.max keeps the first argument unless the next is greater
    return predictor;
}
static AUKIT_DEV int rd16(const unsigned char *p) { return (short)(p[0] | p[1] << 8); }

__global__ __launch_bounds__(64) void k_msadpcm(const MsParams P) {
    const unsigned long long j = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= P.njobs) return;
    const MsJob job = P.jobs[j];
    const unsigned char *blk = P.src + job.blk_off, *h = P.src + job.hdr_off;
    auto norm = [&](double p) {
        const double v = p / (p < 0 ? P.div_neg : P.div_pos);
        return P.floor_all ? floor(v) : v;
    };
    auto emit = [&](double *o, unsigned long long i, double p) { o[i] = norm(p); };
    typedef unsigned u32u __attribute__((aligned(1)));
    typedef double dbl2a __attribute__((ext_vector_type(2), aligned(8)));
    auto nibs = [](int b, int &hi, int &lo) { hi = b >> 4; lo = b & 15; if (hi >= 8) hi -= 16; if (lo >= 8) lo -= 16; };
    if (P.C == 2) {
        const int piL = h[0], piR = h[1];
        if (piL >= P.ncoef || piR >= P.ncoef) { atomicCAS(P.err, 0, 1); return; }
        double dL = rd16(h + 2), dR = rd16(h + 4), s1L = rd16(h + 6), s1R = rd16(h + 8), s2L = rd16(h + 10), s2R = rd16(h + 12);
        const double c1L = P.coef1[piL], c2L = P.coef2[piL], c1R = P.coef1[piR], c2R = P.coef2[piR];
        double *oL = P.out + job.out_off, *oR = P.out + job.out_off_r;
        emit(oL, 0, s2L); emit(oL, 1, s1L); emit(oR, 0, s2R); emit(oR, 1, s1R);
        unsigned long long w = 2;
        int i = 14;
        for (; i + 4 <= P.block_align; i += 4) {
            const unsigned word = *reinterpret_cast<const u32u *>(blk + i);
            double l[4], r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int hi, lo;
                nibs((int)((word >> (8 * k)) & 0xFF), hi, lo);
                l[k] = norm(ms_step(s1L, s2L, dL, c1L, c2L, hi, c_ms_adapt));
                r[k] = norm(ms_step(s1R, s2R, dR, c1R, c2R, lo, c_ms_adapt));
            }
            dbl2a v;
            v.x = l[0]; v.y = l[1]; *reinterpret_cast<dbl2a *>(oL + w) = v;
            v.x = l[2]; v.y = l[3]; *reinterpret_cast<dbl2a *>(oL + w + 2) = v;
            v.x = r[0]; v.y = r[1]; *reinterpret_cast<dbl2a *>(oR + w) = v;
            v.x = r[2]; v.y = r[3]; *reinterpret_cast<dbl2a *>(oR + w + 2) = v;
            w += 4;
        }
        for (; i < P.block_align; i++) {
            int hi, lo;
            nibs(blk[i], hi, lo);
            emit(oL, w, ms_step(s1L, s2L, dL, c1L, c2L, hi, c_ms_adapt));
            emit(oR, w, ms_step(s1R, s2R, dR, c1R, c2R, lo, c_ms_adapt));
            w++;
        }
    } else {
        const int pi = h[0];
        if (pi >= P.ncoef) { atomicCAS(P.err, 0, 1); return; }
        double d = rd16(h + 1), s1 = rd16(h + 3), s2 = rd16(h + 5);
        const double c1 = P.coef1[pi], c2 = P.coef2[pi];
        double *o = P.out + job.out_off;
        // stream mono leaves the two header samples unfloored too (:2708-2709)
        emit(o, 0, s2); emit(o, 1, s1);
        unsigned long long w = 2;
        int i = 7;
        for (; i + 4 <= P.block_align; i += 4) {
            const unsigned word = *reinterpret_cast<const u32u *>(blk + i);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int hi, lo;
                nibs((int)((word >> (8 * k)) & 0xFF), hi, lo);
                dbl2a v;
                v.x = norm(ms_step(s1, s2, d, c1, c2, hi, c_ms_adapt));
                v.y = norm(ms_step(s1, s2, d, c1, c2, lo, c_ms_adapt));
                *reinterpret_cast<dbl2a *>(o + w) = v;
                w += 2;
            }
        }
        for (; i < P.block_align; i++) {
            int hi, lo;
            nibs(blk[i], hi, lo);
            emit(o, w++, ms_step(s1, s2, d, c1, c2, hi, c_ms_adapt));
            emit(o, w++, ms_step(s1, s2, d, c1, c2, lo, c_ms_adapt));
        }
    }
}

static const int ms_c1_default[7] = {256, 512, 0, 192, 240, 460, 392}, ms_c2_default[7] = {0, -256, 0, 64, 0, -208, -232};  // :1304

// decodes every block of every stream into fp64 rows in ctx->tmp_buf; rows are per (stream, channel), blocks back to back
static int msadpcm_rows(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, bool stream_mode, std::vector<uint64_t> &row_off,
                        std::vector<uint64_t> &row_len, std::vector<uint64_t> &nblocks, uint64_t *spb_out) {
    const int C = d->channels;
    if (C != 1 && C != 2) return fail(AUKIT_E_LUA, "Unsupported number of channels: %d", C);
    const uint64_t ba = (uint64_t)d->block_align;
    if (d->block_align < (C == 2 ? 15 : 8)) return fail(AUKIT_E_ARG, "bad blockAlign");
    const uint64_t spb = C == 2 ? (ba - 14) + 2 : (ba - 7) * 2 + 2;  // samples decoded per block per channel
    *spb_out = spb;
    std::vector<MsJob> jobs;
    row_off.assign((size_t)in->n * C, 0);
    row_len.assign((size_t)in->n * C, 0);
    nblocks.assign(in->n, 0);
    uint64_t tot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        const uint64_t nblk = (nb + ba - 1) / ba;  // for n = 1, #data, blockAlign
        if (nblk && nb % ba != 0) return fail(AUKIT_E_LUA, "bad argument #1 to 'rshift' (number expected, got nil)");
        nblocks[s] = nblk;
        const uint64_t L = nblk * spb, stride = round_up(std::max<uint64_t>(L, 1), 2);
        for (int c = 0; c < C; c++) { row_off[(size_t)s * C + c] = tot + (uint64_t)c * stride; row_len[(size_t)s * C + c] = L; }
        for (uint64_t b = 0; b < nblk; b++) {
            MsJob j;
            j.blk_off = in->off[s] + b * ba;
            j.hdr_off = C == 1 ? in->off[s] : j.blk_off;
            j.out_off = tot + b * spb;
            j.out_off_r = tot + stride + b * spb;
            jobs.push_back(j);
        }
        tot += stride * C;
    }
    int rc = ctx->tmp_buf.ensure((size_t)tot * 8 + 64);
    if (rc) return rc;
    const size_t jbytes = jobs.size() * sizeof(MsJob);
    if ((rc = ctx->tmp_buf2.ensure(jbytes + 16))) return rc;
    if (jbytes) { int hrc = h2d_table(ctx, ctx->tmp_buf2.p, jobs.data(), jbytes); if (hrc) return hrc; }
    int *err = reinterpret_cast<int *>(reinterpret_cast<char *>(ctx->tmp_buf2.p) + jbytes);
    AUKIT_HIP_CHECK(hipMemsetAsync(err, 0, 8, ctx->stream));
    if (jobs.empty()) return AUKIT_OK;
    MsParams P{};
    P.src = in->data(); P.jobs = reinterpret_cast<const MsJob *>(ctx->tmp_buf2.p); P.njobs = jobs.size();
    P.C = C; P.block_align = d->block_align;
    if (d->ncoef > 0) { P.ncoef = std::min(d->ncoef, 32); for (int i = 0; i < P.ncoef; i++) { P.coef1[i] = d->coef1[i]; P.coef2[i] = d->coef2[i]; } }
    else { P.ncoef = 7; for (int i = 0; i < 7; i++) { P.coef1[i] = ms_c1_default[i]; P.coef2[i] = ms_c2_default[i]; } }
    P.div_neg = stream_mode ? 128 : 32768; P.div_pos = stream_mode ? 127 : 32767;
    P.floor_all = (stream_mode && C == 2) ? 1 : 0;
    P.out = reinterpret_cast<double *>(ctx->tmp_buf.p);
    P.err = err;
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    hipLaunchKernelGGL(k_msadpcm, dim3((unsigned)((jobs.size() + 63) / 64)), dim3(64), 0, ctx->stream, P);
    AUKIT_HIP_CHECK(hipGetLastError());
    if ((rc = ctx_end_kernel(ctx, "k_msadpcm", in->total() + tot * 8))) return rc;
    int herr = 0;
    AUKIT_HIP_CHECK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (herr) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (local 'c1')");  // predictor index beyond the coefficient table
    return AUKIT_OK;
}

// ================================================================= the wave kernel
// one round of the stream kernel: outputs jl .. jl + nj - 1 of every block have their floor(x) in the round's table window.  Lanes walk the
// (block, output) pairs of the wave 64 at a time: 64 = dq nj + dr; dr outputs further the position grows by dr fa = aq fb + ar, nj outputs
// back it shrinks by nj fa = nq fb + nrm.  b0q: nj < 64 (a lane's first pair needs a division)
struct MsRound { unsigned jl, nj, dq, dr, aq, ar, nq, nrm, b0q, pad[3]; };
struct MsWaveParams {
    const unsigned char *src, *safe_hi;      // a 16-byte load at p needs p + 16 <= safe_hi
    const unsigned long long *off;           // per stream byte offsets (n + 1)
    const unsigned long long *blk0;          // per stream: index of its first block in the batch's block list (n + 1)
    unsigned nstreams;
    unsigned long long nblocks, bps;         // bps: blocks per stream when that is the same for every stream, else 0
    int block_align, ncoef;
    short coef1[32], coef2[32];
    int *err;                                // [0] predictor index beyond the coefficient table, [1] a nan met an int16 row
    unsigned spb_dec;                        // samples decoded per block and channel (header samples included)
    // rows mode
    void *rows;
    const unsigned long long *row_off, *row_stride;  // per stream: element offset of channel 0's row, elements between channel rows
    // stream mode
    void *out;
    const unsigned long long *out_off, *out_stride;
    const MsRound *rounds;                   // per round: its outputs and the walk's constants
    const float *wg;                         // phase weights (cubic: w0..w3, linear: fx), fb entries
    unsigned newlen, spb, fa, fb, fmagic;
    double inv_fb, ratio, rcp;
    int exact_rcp;
    int unit;                                // sample rate 48000: every position is an integer, exactly (x = (i - 1) / 1 + 1)
    unsigned sst;                            // int8 stream output: bytes per block and output row in the LDS staging area (an odd number of dwords), 0: outputs go out pair by pair
    unsigned *audit;                         // AUDIT instantiation: [0] bits of the largest |tier 1 - tier 2| (a non-negative float), [1] outputs compared (floor_wave.hip)
};

struct MsLane { int s1, s2, d, c1, c2; double ws1, ws2, wd; };

static AUKIT_DEV int ms_mul24(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
static AUKIT_DEV int ms_mad24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
static AUKIT_DEV int ms_med3(int a, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi)); return r; }

// floor(p / 127) for 0 <= p < 2^15 (checked exhaustively by ms_magic_ok at library load)
static inline constexpr unsigned ms_div127(unsigned p) { return (p * 33027u) >> 22; }
static bool ms_magic_ok() { for (unsigned p = 0; p < 32768; p++) if (ms_div127(p) != p / 127) return false; return true; }

// one int32 step (:1321-1324); `ovf` collects the bits of every delta met
static AUKIT_DEV int ms_step_i(MsLane &L, int nib, int adapt, unsigned &ovf, int lo, int hi) {
    const int lin = ms_mad24(L.s2, L.c2, ms_mul24(L.s1, L.c1));
    const int p = ms_med3(ms_mad24(nib, L.d, lin >> 8), lo, hi);
    L.s2 = L.s1; L.s1 = p;
    const int nd = ms_mul24(adapt, L.d) >> 8;
    L.d = nd < 16 ? 16 : nd;
    ovf |= (unsigned)L.d;
    return p;
}
// what a table entry holds — EK 0: the predictor p itself (rows mode); 1: stream.msadpcm stereo's math_floor(p / (p < 0 and 128 or 127)) (:2652);
// 2: stream.msadpcm mono's sample p / (p < 0 and 128 or 127) (:2708-2722) ROUNDED to f32 — what the f32 tier interpolates; the fp64 tiers
// get p back from it exactly (ms_sample_p: |v * 127 - p| < 0.01) and divide in fp64 like the Lua.  A nan stays a nan in all three.
template <int EK> static AUKIT_DEV float ms_entry_i(int p) {
    if constexpr (EK == 1) return (float)(p < 0 ? (p >> 7) : (int)ms_div127((unsigned)p));  // the double quotient cannot round across an integer
    else if constexpr (EK == 2) return (float)p * (p < 0 ? 1.0f / 128.0f : 1.0f / 127.0f);
    else return (float)p;
}
template <int EK> static AUKIT_DEV float ms_entry_d(double p) {
    if constexpr (EK == 1) return (float)floor(p / (p < 0 ? 128.0 : 127.0));
    else if constexpr (EK == 2) return (float)p * (p < 0 ? 1.0f / 128.0f : 1.0f / 127.0f);
    else return (float)p;
}
struct CvMsSample {   // table entry (EK 2) -> the reference's double
    static AUKIT_DEV double cv(double v) {
        const double p = rint(v * (v < 0 ? 128.0 : 127.0));
        return p < 0 ? p * (1.0 / 128) : div_rcp(p, 127.0, 1.0 / 127.0);
    }
};

// the 16 bytes `w` (nbytes of them count) of a block in fp64, op for op the Lua — lanes whose delta left the int32 range
template <int C, int EK>
static AUKIT_DEV void ms_vec_wide(unsigned w0, unsigned w1, unsigned w2, unsigned w3, int nbytes, MsLane (&L)[C], const int *adl, float *r0, float *r1, int slot) {
    for (int b = 0; b < nbytes; b++) {
        const int byte = (int)(w0 & 0xFF);
        w0 = (w0 >> 8) | (w1 << 24); w1 = (w1 >> 8) | (w2 << 24); w2 = (w2 >> 8) | (w3 << 24); w3 >>= 8;
        int hi = byte >> 4, lo = byte & 15;
        if (hi >= 8) hi -= 16;
        if (lo >= 8) lo -= 16;
        if constexpr (C == 1) {
            r0[slot++] = ms_entry_d<EK>(ms_step(L[0].ws1, L[0].ws2, L[0].wd, (double)L[0].c1, (double)L[0].c2, hi, adl));
            r0[slot++] = ms_entry_d<EK>(ms_step(L[0].ws1, L[0].ws2, L[0].wd, (double)L[0].c1, (double)L[0].c2, lo, adl));
        } else {
            r0[slot] = ms_entry_d<EK>(ms_step(L[0].ws1, L[0].ws2, L[0].wd, (double)L[0].c1, (double)L[0].c2, hi, adl));
            r1[slot] = ms_entry_d<EK>(ms_step(L[1].ws1, L[1].ws2, L[1].wd, (double)L[1].c1, (double)L[1].c2, lo, adl));
            slot++;
        }
    }
}

// the same bytes in int32 for every lane of the wave; FULL: all 16, straight-line, 16-byte LDS stores
template <int C, int EK, bool FULL>
static AUKIT_DEV void ms_vec_int(const unsigned (&w)[4], int nbytes, MsLane (&L)[C], const int *adl, float *r0, float *r1, int slot, unsigned &ovf) {
    const int lo16 = -32768, hi16 = 32767;
    if constexpr (FULL) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned word = w[q];
            float e0[8], e1[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int hi = __builtin_amdgcn_sbfe((int)word, 8 * k + 4, 4), lo = __builtin_amdgcn_sbfe((int)word, 8 * k, 4);
                const int ah = adl[__builtin_amdgcn_ubfe(word, 8 * k + 4, 4)], al = adl[__builtin_amdgcn_ubfe(word, 8 * k, 4)];
                if constexpr (C == 1) {
                    e0[2 * k] = ms_entry_i<EK>(ms_step_i(L[0], hi, ah, ovf, lo16, hi16));
                    e0[2 * k + 1] = ms_entry_i<EK>(ms_step_i(L[0], lo, al, ovf, lo16, hi16));
                } else {
                    e0[k] = ms_entry_i<EK>(ms_step_i(L[0], hi, ah, ovf, lo16, hi16));
                    e1[k] = ms_entry_i<EK>(ms_step_i(L[1], lo, al, ovf, lo16, hi16));
                }
            }
            if constexpr (C == 1) {
                *reinterpret_cast<float4 *>(r0 + slot + 8 * q) = make_float4(e0[0], e0[1], e0[2], e0[3]);
                *reinterpret_cast<float4 *>(r0 + slot + 8 * q + 4) = make_float4(e0[4], e0[5], e0[6], e0[7]);
            } else {
                *reinterpret_cast<float4 *>(r0 + slot + 4 * q) = make_float4(e0[0], e0[1], e0[2], e0[3]);
                *reinterpret_cast<float4 *>(r1 + slot + 4 * q) = make_float4(e1[0], e1[1], e1[2], e1[3]);
            }
        }
    } else {
        unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        for (int b = 0; b < nbytes; b++) {
            const unsigned byte = w0 & 0xFF;
            w0 = (w0 >> 8) | (w1 << 24); w1 = (w1 >> 8) | (w2 << 24); w2 = (w2 >> 8) | (w3 << 24); w3 >>= 8;
            const int hi = __builtin_amdgcn_sbfe((int)byte, 4, 4), lo = __builtin_amdgcn_sbfe((int)byte, 0, 4);
            const int ah = adl[byte >> 4], al = adl[byte & 15];
            if constexpr (C == 1) {
                r0[slot++] = ms_entry_i<EK>(ms_step_i(L[0], hi, ah, ovf, lo16, hi16));
                r0[slot++] = ms_entry_i<EK>(ms_step_i(L[0], lo, al, ovf, lo16, hi16));
            } else {
                r0[slot] = ms_entry_i<EK>(ms_step_i(L[0], hi, ah, ovf, lo16, hi16));
                r1[slot] = ms_entry_i<EK>(ms_step_i(L[1], lo, al, ovf, lo16, hi16));
                slot++;
            }
        }
    }
}

enum { MS_ROWS_I16 = 0, MS_ROWS_F32 = 1, MS_ROWS_F64 = 2, MS_STREAM = 3 };
// the stream kernels' list of deferred output lines (flushed when the next lines would not fit; a call of defer() adds at most 64).  With the register
// window (REGS) the list lies where the staged fetch keeps its 64 block offsets: the kernel's LDS — what bounds its resident waves — does not grow
constexpr unsigned MS_DL_CAP = 64;

// C channels; RB bytes of a block per round; MODE one of the enums above; stream mode: INTERP, MIX (stereo: l + r / 2, :2672), OUT_T
// FB bytes of a block per FETCH (the staging area holds them; FB / RB rounds consume it): the stream kernels fetch 64 bytes at a time and decode
// them in four rounds of 16 — with 16-byte fetches every 128-byte line of the input came from HBM eight times (PMC FETCH_SIZE, profiles/)
template <int C, int RB, int MODE, int INTERP, bool MIX, typename OUT_T, int FB = RB, bool AUDIT = false>
__global__ __launch_bounds__(64) void k_ms_wave(const MsWaveParams P) {
    [[maybe_unused]] float amax = 0.f;
    [[maybe_unused]] unsigned acnt = 0;
    constexpr int R = RB * 2 / C;         // samples per channel and round
    constexpr int ROW = 4 + R + 1;        // floats per (block, channel) row: slots 0..3 = table indices R r - 1 .. R r + 2, slot(t) = t - R r + 1
                                          // (+ 1: an odd stride — a lane's row starts in its own bank; with 36 or 20 every 8th lane shared one)
    constexpr int SEGS = FB / 16;         // 16-byte vectors of a block per fetch
    constexpr int RSEG = RB / 16;         // ... per round
    constexpr int RPF = FB / RB;          // rounds per fetch
    // REGS (FB == 128, round 4): no staging area at all — every lane reads ITS block's bytes itself, a whole 128-byte line at a time (eight aligned
    // 16-byte loads into registers, one line ahead), and cuts a round's 16 bytes out of two neighbouring vectors (per-lane byte offset: blocks start
    // 7 or 14 bytes into a line).  With 16 bytes per lane and round from 64 lines 1 KiB apart every line came from HBM eight times over (PMC: 6.5 GB
    // fetched for 0.9 GB of blocks — the power-of-two stride also lands the wave's 64 lines in a few L2 sets)
    constexpr bool REGS = FB == 128;
    static_assert(!REGS || RB == 16, "the register window serves 16-byte rounds");
    constexpr int INS = REGS ? 0 : (FB == 16 ? 4 : FB / 4 + 4);  // dwords between two blocks in the staging area (conflict-free ds_read_b128 per lane)
    constexpr int HB = 7 * C;
    constexpr bool STREAM = MODE == MS_STREAM;
    constexpr bool FLOORED = STREAM && C == 2;
    constexpr int EK = !STREAM ? 0 : (C == 2 ? 1 : 2);   // what the table holds (ms_entry_i)
    constexpr int WF = INTERP == AUKIT_INTERP_CUBIC ? 4 : 1;
    extern __shared__ float smf[];
    const int lane = threadIdx.x;
    unsigned *const inl = reinterpret_cast<unsigned *>(smf);                      // 64 × INS dwords
    float *const tab = smf + 64 * INS;                                             // C × 64 × ROW floats
    unsigned long long *const bp = reinterpret_cast<unsigned long long *>(tab + C * 64 * ROW);  // 64 block byte offsets (0: none)
    unsigned long long *const ob = bp + 64;                                        // 64 output element offsets (channel 0, block's first element)
    unsigned long long *const os = ob + 64;                                        // 64 channel strides
    int *const adl = reinterpret_cast<int *>(os + 64);                             // 16
    float *const wt = reinterpret_cast<float *>(adl + 16);                         // fb × WF (stream mode)
    constexpr bool LANEOUT = MODE == MS_STREAM && sizeof(OUT_T) == 1;               // int8 chunks: every lane makes its own block's outputs (below)
    [[maybe_unused]] signed char *const stage = reinterpret_cast<signed char *>(wt + ((MODE == MS_STREAM ? P.fb * WF : 0u) + 3u & ~3u));   // (C == 2 && !MIX ? 2 : 1) × 64 × P.sst bytes
    [[maybe_unused]] unsigned *const dl = REGS ? reinterpret_cast<unsigned *>(bp) : reinterpret_cast<unsigned *>(stage + (((STREAM ? (size_t)(C == 2 && !MIX ? 2 : 1) * 64 * P.sst : 0) + 7) & ~(size_t)7));   // MS_DL_CAP entries of two dwords: the deferred lines (stream mode)
    if (lane < 16) adl[lane] = c_ms_adapt[lane];
    if constexpr (STREAM) for (unsigned i = lane; i < P.fb * WF; i += 64) wt[i] = P.wg[i];
    const unsigned long long ba = (unsigned long long)P.block_align;
    const unsigned long long gb = (unsigned long long)blockIdx.x * 64 + lane;
    const bool valid = gb < P.nblocks;
    const unsigned nvalid = (unsigned)(P.nblocks - (unsigned long long)blockIdx.x * 64 < 64 ? P.nblocks - (unsigned long long)blockIdx.x * 64 : 64);
    // this lane's block
    unsigned s = 0;
    unsigned long long bi = 0;
    if (valid) {
        if (P.bps) s = (unsigned)(gb / P.bps);
        else { unsigned lo = 0, hi = P.nstreams; while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (P.blk0[mid] <= gb) lo = mid; else hi = mid; } s = lo; }
        bi = gb - P.blk0[s];
    }
    const unsigned long long soff = valid ? P.off[s] : 0, boff = soff + bi * ba;
    bp[lane] = boff;
    if constexpr (STREAM) { ob[lane] = valid ? P.out_off[s] + bi * (unsigned long long)P.newlen : 0; os[lane] = valid ? P.out_stride[s] : 0; }
    else { ob[lane] = valid ? P.row_off[s] + bi * (unsigned long long)P.spb_dec : 0; os[lane] = valid ? P.row_stride[s] : 0; }
    // header (mono: the stream's first block for every block, Q9 — `str_unpack("<!1Bhhh", data)` has no position)
    MsLane L[C];
    float *const r0 = tab + lane * ROW, *const r1 = tab + (64 + lane) * ROW;
    bool wide = false;
    {
        const unsigned char *h = P.src + (C == 1 ? soff : boff);
        bool bad = false;
        if (valid) {
            if constexpr (C == 2) {
                const int piL = h[0], piR = h[1];
                bad = piL >= P.ncoef || piR >= P.ncoef;
                L[0].d = rd16(h + 2); L[1].d = rd16(h + 4); L[0].s1 = rd16(h + 6); L[1].s1 = rd16(h + 8); L[0].s2 = rd16(h + 10); L[1].s2 = rd16(h + 12);
                L[0].c1 = P.coef1[piL & 31]; L[0].c2 = P.coef2[piL & 31]; L[1].c1 = P.coef1[piR & 31]; L[1].c2 = P.coef2[piR & 31];
            } else {
                const int pi = h[0];
                bad = pi >= P.ncoef;
                L[0].d = rd16(h + 1); L[0].s1 = rd16(h + 3); L[0].s2 = rd16(h + 5);
                L[0].c1 = P.coef1[pi & 31]; L[0].c2 = P.coef2[pi & 31];
            }
        } else {
            for (int c = 0; c < C; c++) { L[c].d = 16; L[c].s1 = L[c].s2 = 0; L[c].c1 = L[c].c2 = 0; }
        }
        for (int c = 0; c < C; c++) { L[c].ws1 = L[c].ws2 = L[c].wd = 0; }
        if (bad) atomicCAS(P.err, 0, 1);
        // table indices 1, 2 = sample2, sample1 (:2708-2709, :2648-2651)
        r0[0] = 0; r0[1] = 0; r0[2] = ms_entry_i<EK>(L[0].s2); r0[3] = ms_entry_i<EK>(L[0].s1);
        if constexpr (C == 2) { r1[0] = 0; r1[1] = 0; r1[2] = ms_entry_i<EK>(L[1].s2); r1[3] = ms_entry_i<EK>(L[1].s1); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int ndata = P.block_align - HB;                 // data bytes per block
    const int nr = (ndata + RB - 1) / RB;                  // rounds
    typedef unsigned u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    // REGS: the lane's aligned 16-byte vectors, eight (a line) held and eight requested
    [[maybe_unused]] const unsigned char *cb = nullptr;
    [[maybe_unused]] unsigned sh = 0;
    [[maybe_unused]] uint4 ra[REGS ? 8 : 1], rn[REGS ? 8 : 1];
    [[maybe_unused]] auto ldc = [&](int j) -> uint4 {   // vector j of this lane's block data (zeros beyond the allocation / for a lane without a block)
        const unsigned char *p = cb + 16 * (long long)j;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (valid && p + 16 <= P.safe_hi && p + 16 > P.src) {
            typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
            const u32x4a t = *(const __attribute__((address_space(1))) u32x4a *)(uintptr_t)p;   // (aligned by construction; global_load_dwordx4 — as a generic pointer's FLAT load it sat on lgkmcnt as well, and every LDS wait of the round waited for the line requested for eight rounds later)
            v = make_uint4(t.x, t.y, t.z, t.w);
        } else if (valid) {
            unsigned t[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; k++) if (p + k >= P.src && p + k < P.safe_hi) t[k >> 2] |= (unsigned)p[k] << (8 * (k & 3));
            v = make_uint4(t[0], t[1], t[2], t[3]);
        }
        return v;
    };
    // staging: vector i of this lane belongs to block i * (64 / SEGS) + lane / SEGS, bytes (lane % SEGS) * 16 .. + 16 of the round
    const int sblk0 = lane / SEGS, sseg = lane % SEGS;
    auto fetch = [&](int r, u32x4u (&pf)[SEGS]) {
#pragma unroll
        for (int i = 0; i < SEGS; i++) {
            const int b = i * (64 / SEGS) + sblk0;
            const unsigned char *p = P.src + bp[b] + HB + (unsigned long long)r * FB + sseg * 16;
            u32x4u v = {0, 0, 0, 0};
            if ((unsigned)b < nvalid && p + 16 <= P.safe_hi) v = *reinterpret_cast<const u32x4u *>(p);
            else if ((unsigned)b < nvalid) {   // the last bytes of the allocation
                unsigned t[4] = {0, 0, 0, 0};
                for (int k = 0; k < 16 && p + k < P.safe_hi; k++) t[k >> 2] |= (unsigned)p[k] << (8 * (k & 3));
                v.x = t[0]; v.y = t[1]; v.z = t[2]; v.w = t[3];
            }
            pf[i] = v;
        }
    };
    u32x4u pf[SEGS];
    if constexpr (!REGS) fetch(0, pf);
    else {
        const unsigned char *a0 = P.src + boff + HB;
        cb = reinterpret_cast<const unsigned char *>((uintptr_t)a0 & ~(uintptr_t)15);
        sh = (unsigned)(a0 - cb);
#pragma unroll
        for (int k = 0; k < 8; k++) rn[k] = ldc(k);
    }
    [[maybe_unused]] ResampleParams RP;
    if constexpr (STREAM) { RP.ratio = P.ratio; RP.rcp = P.rcp; RP.exact_rcp = P.exact_rcp; RP.sinc_w = 10; RP.pos_mul = 0; }
    const int nf = (ndata + FB - 1) / FB;                  // fetches
    for (int r = 0; r < nr; r++) {
        const int sub = r % RPF;   // wave-uniform
        if constexpr (REGS) {
            if (sub == 0) {   // the requested line becomes the held one; the line after it is requested (round r needs vectors r and r + 1: up to vector nr)
#pragma unroll
                for (int k = 0; k < 8; k++) ra[k] = rn[k];
                if (r + 8 <= nr) {
#pragma unroll
                    for (int k = 0; k < 8; k++) rn[k] = ldc(r + 8 + k);
                }
            }
        } else
        if (sub == 0) {
            // this fetch's bytes into the staging area, the next fetch's on their way
#pragma unroll
            for (int i = 0; i < SEGS; i++) *reinterpret_cast<uint4 *>(inl + (i * (64 / SEGS) + sblk0) * INS + sseg * 4) = make_uint4(pf[i].x, pf[i].y, pf[i].z, pf[i].w);
            if (r / RPF + 1 < nf) fetch(r / RPF + 1, pf);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const int nby = ndata - r * RB < RB ? ndata - r * RB : RB;   // bytes of this round
#pragma unroll
        for (int v = 0; v < RSEG; v++) {
            const int nb = nby - 16 * v;   // wave-uniform
            if (nb <= 0) break;
            uint4 q;
            if constexpr (REGS) {
                uint4 c0, c1;
                switch (sub) {   // wave-uniform
                case 0: c0 = ra[0]; c1 = ra[1]; break;
                case 1: c0 = ra[1]; c1 = ra[2]; break;
                case 2: c0 = ra[2]; c1 = ra[3]; break;
                case 3: c0 = ra[3]; c1 = ra[4]; break;
                case 4: c0 = ra[4]; c1 = ra[5]; break;
                case 5: c0 = ra[5]; c1 = ra[6]; break;
                case 6: c0 = ra[6]; c1 = ra[7]; break;
                default: c0 = ra[7]; c1 = rn[0]; break;
                }
                // bytes sh .. sh + 15 of the 32: a per-lane dword offset (selects) and byte offset (v_alignbyte)
                const unsigned dsh = sh >> 2, bsh = sh & 3u;
                const unsigned d8[9] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, 0u};
                unsigned e[5];
#pragma unroll
                for (int k = 0; k < 5; k++) e[k] = dsh == 0 ? d8[k] : (dsh == 1 ? d8[k + 1] : (dsh == 2 ? d8[k + 2] : d8[k + 3]));
                q = make_uint4(__builtin_amdgcn_alignbyte(e[1], e[0], bsh), __builtin_amdgcn_alignbyte(e[2], e[1], bsh), __builtin_amdgcn_alignbyte(e[3], e[2], bsh),
                               __builtin_amdgcn_alignbyte(e[4], e[3], bsh));
            } else q = *reinterpret_cast<const uint4 *>(inl + lane * INS + 4 * (sub * RSEG + v));
            const unsigned w[4] = {q.x, q.y, q.z, q.w};
            const int slot = 4 + v * (32 / C);
            const int s1s[2] = {L[0].s1, L[C - 1].s1}, s2s[2] = {L[0].s2, L[C - 1].s2}, ds[2] = {L[0].d, L[C - 1].d};
            unsigned ovf = 0;
            if (nb >= 16) ms_vec_int<C, EK, true>(w, 16, L, adl, r0, r1, slot, ovf);
            else ms_vec_int<C, EK, false>(w, nb, L, adl, r0, r1, slot, ovf);
            const bool need = valid && (wide || (ovf >> 21) != 0);
            if (__builtin_amdgcn_ballot_w64(need)) {
                if (need) {
                    if (!wide) {
                        wide = true;
                        L[0].ws1 = s1s[0]; L[0].ws2 = s2s[0]; L[0].wd = ds[0];
                        if constexpr (C == 2) { L[1].ws1 = s1s[1]; L[1].ws2 = s2s[1]; L[1].wd = ds[1]; }
                    }
                    ms_vec_wide<C, EK>(w[0], w[1], w[2], w[3], nb >= 16 ? 16 : nb, L, adl, r0, r1, slot);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nsamp = nby * 2 / C;   // samples per channel decoded in this round
        if constexpr (!STREAM) {
            // block b's samples of this round: table indices R r + 3 .., row elements 2 + R r ..
            for (unsigned idx = lane; idx < nvalid * (unsigned)R; idx += 64) {
                const unsigned b = idx / R, i = idx % R;
                if ((int)i >= nsamp) continue;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float pv = tab[(c * 64 + b) * ROW + 4 + i];
                    const unsigned long long e = ob[b] + (unsigned long long)c * os[b] + 2 + (unsigned long long)r * R + i;
                    if constexpr (MODE == MS_ROWS_I16) {
                        if (pv != pv) atomicCAS(P.err + 1, 0, 1);
                        reinterpret_cast<short *>(P.rows)[e] = (short)(int)pv;
                    } else {
                        const double pd = (double)pv;
                        const double nv = pd < 0 ? pd * (1.0 / 32768) : div_rcp(pd, 32767.0, 1.0 / 32767.0);   // predictor / (predictor < 0 and 32768 or 32767)  :1322
                        if constexpr (MODE == MS_ROWS_F64) reinterpret_cast<double *>(P.rows)[e] = nv;
                        else reinterpret_cast<float *>(P.rows)[e] = (float)nv;
                    }
                }
            }
            if (r == 0 && valid) {   // the two header samples
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float *rw = c ? r1 : r0;
                    for (int i = 0; i < 2; i++) {
                        const float pv = rw[2 + i];
                        const unsigned long long e = ob[lane] + (unsigned long long)c * os[lane] + i;
                        if constexpr (MODE == MS_ROWS_I16) reinterpret_cast<short *>(P.rows)[e] = (short)(int)pv;
                        else {
                            const double pd = (double)pv;
                            const double nv = pd < 0 ? pd * (1.0 / 32768) : div_rcp(pd, 32767.0, 1.0 / 32767.0);
                            if constexpr (MODE == MS_ROWS_F64) reinterpret_cast<double *>(P.rows)[e] = nv;
                            else reinterpret_cast<float *>(P.rows)[e] = (float)nv;
                        }
                    }
                }
            }
        } else {
            // outputs whose floor(x) = k lies in R r + 1 .. R r + R: every tap k - 2 .. k + 2 is in the table now
            // (through the constant address space: with a uniform index that is an s_load, counted by lgkmcnt — as a plain global load it sits behind
            // vmcnt(0), i.e. behind the previous round's output stores and this round's prefetch)
            MsRound rd;
            {
                const __attribute__((address_space(4))) unsigned *rq = (const __attribute__((address_space(4))) unsigned *)(P.rounds + r);
                rd.jl = rq[0]; rd.nj = rq[1]; rd.dq = rq[2]; rd.dr = rq[3]; rd.aq = rq[4]; rd.ar = rq[5]; rd.nq = rq[6]; rd.nrm = rq[7]; rd.b0q = rq[8];
            }
            const unsigned nj = rd.nj;
            const int kbase = R * r;            // table index of slot 1
            const int w_hi = (int)P.spb_dec;    // #left
            Seg sg; sg.w_lo = 1; sg.w_hi = w_hi;
            const unsigned total = nvalid * nj;
            // (wave-uniform) every output of the round has floor(x) = k in kbase + 1 .. kbase + R: are all their taps inside the table?
            const bool clean_round = INTERP == AUKIT_INTERP_CUBIC ? (kbase + 1 >= 3 && kbase + R + 2 <= w_hi) : (INTERP == AUKIT_INTERP_LINEAR ? (kbase + 1 >= 2 && kbase + R + 1 <= w_hi) : false);
            // one output: block-relative index j at position (q0, remc) [x - 1 = q0 + remc / fb], from the block's table rows t0 (t1: the second channel)
            // SEL 0: tier 1 for every line of the output; what it does not vouch for is NOT computed here — `fails` says which lines (bit 0, bit 1) and
            // the caller puts them on the wave's list (defer / flush below).  SEL 1 / 2: tiers 2 and 3 for line 0 / 1 of an output from that list.
            auto one_output = [&](auto selc, unsigned j, unsigned q0, unsigned remc, const float *t0, const float *t1, bool active, OUT_T &v0, OUT_T &v1, unsigned &fails) {
                constexpr int SEL = decltype(selc)::value;
                const int k = (int)q0 + 1;  // floor(x), exact
                // (SEL 3 = SEL 0 in a round all of whose outputs have their taps inside the table — every round of a block but its first and its last one or
                // two: `inside` is a constant there, five instructions per output less)
                const bool inside = SEL == 3 || (INTERP == AUKIT_INTERP_CUBIC ? (k >= 3 && k + 2 <= w_hi) : (INTERP == AUKIT_INTERP_LINEAR ? (k >= 2 && k + 1 <= w_hi) : (k >= 2 && remc != 0)));
                const int s1 = inside ? k - kbase + 1 : 3;   // slot of table index k
                // tier 1: the interpolated value in f32 — taps of magnitude < 259 with <= 2^-23 of relative rounding, weights rounded once, four
                // roundings below 1024: < 3e-4 in all (cf. k_ima_stream_f32).  Written around p1 (the weights sum to 1): equal taps — silence, a
                // clipped stretch — give p1 itself, exactly, and `spread` (the largest |tap - p1|, one v_max3) says so; see `put`
                auto tier1 = [&](const float *tp, float &spread) -> float {
                    const float p1 = tp[s1], p2 = tp[s1 + 1];
                    if constexpr (INTERP == AUKIT_INTERP_NONE) { spread = 1.0f; return p1; }
                    else if constexpr (INTERP == AUKIT_INTERP_LINEAR) { spread = C == 1 ? fabsf(p2 - p1) : 1.0f; return __builtin_fmaf(p2 - p1, wt[remc], p1); }
                    else if constexpr (C == 1) {
                        const float4 w = *reinterpret_cast<const float4 *>(wt + 4 * remc);
                        const float d0 = tp[s1 - 1] - p1, d2 = p2 - p1, d3 = tp[s1 + 2] - p1;
                        spread = fmaxf(fmaxf(fabsf(d0), fabsf(d2)), fabsf(d3));
                        return __builtin_fmaf(w.w, d3, __builtin_fmaf(w.z, d2, __builtin_fmaf(w.x, d0, p1)));
                    } else {   // (stereo: entries are floored integers and tier 2 is cheap there — the plain form measured faster: 2.81 against 3.00 ms)
                        const float4 w = *reinterpret_cast<const float4 *>(wt + 4 * remc);
                        const float p0 = tp[s1 - 1], p3 = tp[s1 + 2];
                        spread = 1.0f;
                        return __builtin_fmaf(w.w, p3, __builtin_fmaf(w.z, p2, __builtin_fmaf(w.y, p1, w.x * p0)));
                    }
                };
                auto tier2 = [&](const float *tp) -> double {   // fp64, exact rational position, FMA Horner
                    auto cv = [](float p) -> double { if constexpr (FLOORED) return (double)p; else return CvMsSample::cv((double)p); };
                    const double p1 = cv(tp[s1]), p2 = cv(tp[s1 + 1]);
                    const double fx = (double)remc * P.inv_fb;
                    double v;
                    if constexpr (INTERP == AUKIT_INTERP_NONE) v = p1;
                    else if constexpr (INTERP == AUKIT_INTERP_LINEAR) v = __builtin_fma(p2 - p1, fx, p1);
                    else {
                        const double p0 = cv(tp[s1 - 1]), p3 = cv(tp[s1 + 2]);
                        const double c3 = __builtin_fma(1.5, p1 - p2, 0.5 * (p3 - p0));
                        const double c2 = __builtin_fma(-2.5, p1, p0) + __builtin_fma(2.0, p2, -0.5 * p3);
                        const double c1 = 0.5 * (p2 - p0);
                        v = __builtin_fma(__builtin_fma(__builtin_fma(c3, fx, c2), fx, c1), fx, p1);
                    }
                    return v;
                };
                auto tier3 = [&](const float *tp) -> double {   // the reference's own operation order on the same table
                    bool isint;
                    if constexpr (FLOORED) return eval_at<INTERP, false, float, CvIdentity>(RP, sg, tp + 1, kbase, j, &isint);
                    else return eval_at<INTERP, false, float, CvMsSample>(RP, sg, tp + 1, kbase, j, &isint);
                };
                // tiers 2 and 3 for one output line: ta alone, or ta + tb / 2 (:2672).  Taps that are all equal need no margin: the FMA form
                // returns cv(p1) itself (every coefficient is an exact zero) and the reference's sum is cv(p1) + a few ulps, on the same side
                // of every integer (cv(p1) is an integer only when p1's arithmetic is exact) — digital silence stays off tier 3
                auto flat = [&](const float *tp) {
                    if constexpr (INTERP == AUKIT_INTERP_LINEAR) return tp[s1] == tp[s1 + 1];
                    else if constexpr (INTERP == AUKIT_INTERP_CUBIC) return tp[s1 - 1] == tp[s1] && tp[s1] == tp[s1 + 1] && tp[s1 + 1] == tp[s1 + 2];
                    else return true;
                };
                auto slow = [&](const float *ta, const float *tb) -> double {
                    double d = 0;
                    bool ok = inside;
                    if (inside) {
                        d = tier2(ta);
                        bool fl = P.unit || (remc != 0 && flat(ta));   // unit: the value is d[x] itself
                        if (tb) { d = d + tier2(tb) / 2; fl = P.unit || (fl && flat(tb)); }
                        if (!fl) { const double fr = d - floor(d); ok = fr > 1e-6 && fr < 1 - 1e-6; }
                    }
                    if (!ok) { d = tier3(ta); if (tb) d = d + tier3(tb) / 2; }
                    return lua_clamp(floor(d), -128, 127);   // :2673 / :2674 / :2727
                };
                auto slow_value = [&](const float *ta, const float *tb) -> OUT_T {
                    const double d = slow(ta, tb);
                    if constexpr (sizeof(OUT_T) == 8) return (OUT_T)d;
                    else return (OUT_T)(int)__builtin_amdgcn_fmed3f((float)d, -128.0f, 127.0f);
                };
                if constexpr (SEL == 1) { v0 = (C == 2 && MIX) ? slow_value(t0, t1) : slow_value(t0, nullptr); return; }
                if constexpr (SEL == 2) { v1 = slow_value(t1, nullptr); return; }
                auto put = [&](float v, float spread, const float *ta, const float *tb, unsigned failbit) -> OUT_T {   // the value (the caller stores it: a pointer that may be LDS or HBM would make every store a flat one)
                    if constexpr (AUDIT) {
                        if (active && inside) { double d2 = tier2(ta); if (tb) d2 = d2 + tier2(tb) / 2; amax = fmaxf(amax, (float)fabs((double)v - d2)); acnt++; }
                    }
                    float fl = floorf(v);
                    const float fr = v - fl;
                    bool accept = inside && fr > MS_TIER1_GUARD && fr < 1 - MS_TIER1_GUARD;
                    if constexpr (FLOORED) { if (INTERP == AUKIT_INTERP_NONE || P.unit) accept = inside; }   // the entry itself (or l + r / 2 of two entries): exact
                    // Equal taps: the difference forms of tier 1 return the entry itself (or l + r / 2), and an f32 entry that is an integer IS the
                    // reference's value (an entry is the predictor times 1/128 — exact — or times RN(1/127): integral only for multiples of 127,
                    // which it then represents exactly; anything else is 1/127 away from every integer).  Measured on the bench's encoder-made
                    // streams: 5 % of the outputs sit in such stretches (the encoder clips) and went through tier 2 one or two lanes at a time —
                    // half of the wave's turns, 1.2 of the kernel's 2.6 ms.
                    if constexpr (INTERP != AUKIT_INTERP_NONE && C == 1) accept = accept || (inside && fr == 0.0f && spread == 0.0f && remc != 0);
                    // (not vouched for: tiers 2 and 3 LATER, for the wave's failures together — computed here, under a branch the whole wave takes when
                    // one lane in 64 needs it (one output in ~400 does: one turn in seven), they were a sixth of the kernel's instructions)
                    if (active && !accept) fails |= failbit;
                    if constexpr (sizeof(OUT_T) == 8) return (OUT_T)fminf(fmaxf(fl, -128.0f), 127.0f);   // (a nan — garbage input on the fp64 path — stays what fmin / fmax make of it)
                    else return (OUT_T)(int)__builtin_amdgcn_fmed3f(fl, -128.0f, 127.0f);
                };
                float sa = 1.0f, sb = 1.0f;
                if constexpr (C == 1) { const float a = tier1(t0, sa); v0 = put(a, sa, t0, nullptr, 1u); }
                else if constexpr (MIX) { const float a = tier1(t0, sa), b2 = tier1(t1, sb); v0 = put(__builtin_fmaf(b2, 0.5f, a), fmaxf(sa, sb), t0, t1, 1u); }   // l + r / 2
                else { const float a = tier1(t0, sa), b2 = tier1(t1, sb); v0 = put(a, sa, t0, nullptr, 1u); v1 = put(b2, sb, t1, nullptr, 2u); }
            };
            // the wave's list of output lines tier 1 did not vouch for: (block, line, j, q0, rem) in two dwords; flushed — tiers 2 / 3 with a lane per
            // entry — when the next lines would not fit and before the round's table rows move on.  `to_stage`: the value goes to the LDS row it was left
            // out of (the int8 chunk rows), else to the audio
            unsigned dcnt = 0;
            auto flush = [&](bool to_stage) {
                if (!dcnt) return;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (unsigned i0 = 0; i0 < dcnt; i0 += 64) {
                    const bool on = i0 + lane < dcnt;
                    const unsigned e0 = on ? dl[2 * (i0 + lane)] : 0u, e1 = on ? dl[2 * (i0 + lane) + 1] : 0u;
                    const unsigned j = e0 & 0x3FFFFFFu, bb = e0 >> 26, q0 = e1 & 0x1FFFFu, remc = (e1 >> 17) & 0x1FFu, lines = e1 >> 26;
                    const float *t0 = tab + __umul24(bb, ROW), *t1 = t0 + 64 * ROW;
                    OUT_T v0 = 0, v1 = 0;
                    unsigned dummy = 0;
                    if (lines & 1u) {
                        one_output(std::integral_constant<int, 1>{}, j, q0, remc, t0, t1, true, v0, v1, dummy);
                        if (to_stage) { if constexpr (sizeof(OUT_T) == 1) stage[bb * P.sst + (j - rd.jl)] = (signed char)v0; }
                        else reinterpret_cast<OUT_T *>(P.out)[ob[bb] + j] = v0;
                    }
                    if constexpr (C == 2 && !MIX) {
                        if (lines & 2u) {
                            one_output(std::integral_constant<int, 2>{}, j, q0, remc, t0, t1, true, v0, v1, dummy);
                            if (to_stage) { if constexpr (sizeof(OUT_T) == 1) stage[(64u + bb) * P.sst + (j - rd.jl)] = (signed char)v1; }
                            else reinterpret_cast<OUT_T *>(P.out)[ob[bb] + os[bb] + j] = v1;
                        }
                    }
                }
                dcnt = 0;
                __builtin_amdgcn_wave_barrier();
            };
            auto defer = [&](bool to_stage, unsigned lines, unsigned bb, unsigned j, unsigned q0, unsigned remc) {   // lines: bit 0 / bit 1 = the output's first / second line
                const bool need = lines != 0;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(need);
                if (m) {   // (wave-uniform)
                    if (dcnt + (unsigned)__builtin_popcountll(m) > MS_DL_CAP) flush(to_stage);
                    const unsigned pos = dcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (need) { dl[2 * pos] = j | bb << 26; dl[2 * pos + 1] = q0 | remc << 17 | lines << 26; }   // (j < 2^26, q0 < 2^17, rem < 2^9: checked by the host)
                    dcnt += (unsigned)__builtin_popcountll(m);
                }
            };
            if (nj && LANEOUT && P.sst) {
                // int8 chunks (round 3).  Every block of the wave has the same geometry, so output j of the round sits at the same (q, rem) in all
                // of them: lane = block, the position walks in scalar registers, the four taps come from the lane's own row (odd stride: its own
                // bank) and the weights from one broadcast read.  The results wait in an LDS row per block and leave as before — pairs (block,
                // output) 64 at a time, consecutive lanes on consecutive bytes — in a loop that only copies.  (The pair-mapped loop below spent
                // ~90 VALU instructions per 64 outputs on finding out which block, row, position and address a lane's pair has: 3150 of a
                // round's 3700.)
                const float *const t0 = tab + lane * ROW, *const t1 = t0 + 64 * ROW;
                signed char *const st0 = stage + (unsigned)lane * P.sst, *const st1 = st0 + 64u * P.sst;
                // (the round's constants come from memory, i.e. in vector registers: say that they are the same in every lane, so that the position,
                // the slot of its taps and the address of its weights are computed once per output by the scalar unit)
                const unsigned jl_s = (unsigned)__builtin_amdgcn_readfirstlane((int)rd.jl), nj_s = (unsigned)__builtin_amdgcn_readfirstlane((int)nj);
                unsigned q0 = (unsigned)(((unsigned long long)jl_s * P.fa) / P.fb), rem = jl_s * P.fa - q0 * P.fb;   // scalar arithmetic (jl fa < 2^32: checked by the host)
                const unsigned dq1 = P.fa / P.fb, dr1 = P.fa - dq1 * P.fb;
                auto fill = [&](auto fastc) {
                for (unsigned jj = 0; jj < nj_s; jj++) {
                    OUT_T v0 = 0, v1 = 0;
                    unsigned fm = 0;
                    one_output(fastc, jl_s + jj, q0, rem, t0, t1, valid, v0, v1, fm);
                    st0[jj] = (signed char)v0;
                    if constexpr (C == 2 && !MIX) st1[jj] = (signed char)v1;
                    defer(true, fm, (unsigned)lane, jl_s + jj, q0, rem);
                    q0 += dq1; rem += dr1;
                    if (rem >= P.fb) { rem -= P.fb; q0++; }
                }
                };
                if (clean_round) fill(std::integral_constant<int, 3>{}); else fill(std::integral_constant<int, 0>{});
                flush(true);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                unsigned b = rd.b0q ? (unsigned)lane / nj : 0, jj = rd.b0q ? (unsigned)lane % nj : (unsigned)lane;
                for (unsigned base = 0; base < total; base += 64) {
                    if (base + lane < total) {
                        OUT_T *const o0 = reinterpret_cast<OUT_T *>(P.out) + ob[b] + rd.jl + jj;
                        *o0 = (OUT_T)stage[b * P.sst + jj];
                        if constexpr (C == 2 && !MIX) o0[os[b]] = (OUT_T)stage[(64u + b) * P.sst + jj];
                    }
                    b += rd.dq; jj += rd.dr;
                    if (jj >= nj) { jj -= nj; b++; }
                }
                __builtin_amdgcn_wave_barrier();
            } else
            if (nj) {
                // lanes walk the round's (block, output) pairs 64 at a time; a lane's pair, its floor(x) - 1 = q and the phase rem advance by
                // additions with one carry each (host-made constants of the round: no multiply or divide per output)
                unsigned b = rd.b0q ? (unsigned)lane / nj : 0, jj = rd.b0q ? (unsigned)lane % nj : (unsigned)lane;
                unsigned q0, rem;
                { const unsigned j = rd.jl + jj; q0 = __umulhi(j * P.fa, P.fmagic); rem = j * P.fa - q0 * P.fb; }
                auto pairs = [&](auto fastc) {
                for (unsigned base = 0; base < total; base += 64) {
                    const bool active = base + lane < total;
                    const unsigned bb = active ? b : 0;
                    const unsigned j = rd.jl + jj, q0c = q0, remc = rem;
                    const float *t0 = tab + __umul24(bb, ROW), *t1 = t0 + 64 * ROW;
                    // the next pair of this lane
                    b += rd.dq; jj += rd.dr; q0 += rd.aq; rem += rd.ar;
                    if (rem >= P.fb) { rem -= P.fb; q0++; }
                    if (jj >= nj) { jj -= nj; b++; q0 -= rd.nq; if (rem < rd.nrm) { rem += P.fb; q0--; } rem -= rd.nrm; }
                    OUT_T *const o0 = reinterpret_cast<OUT_T *>(P.out) + ob[bb] + j;
                    OUT_T v0 = 0, v1 = 0;
                    unsigned fm = 0;
                    one_output(fastc, j, q0c, remc, t0, t1, active, v0, v1, fm);
                    if (active) { if (!(fm & 1u)) *o0 = v0; if constexpr (C == 2 && !MIX) { if (!(fm & 2u)) o0[os[bb]] = v1; } }
                    defer(false, fm, bb, j, q0c, remc);
                }
                };
                if (clean_round) pairs(std::integral_constant<int, 3>{}); else pairs(std::integral_constant<int, 0>{});
                flush(false);
            }
        }
        // the round's last four samples are the next round's slots 0..3
        if (r + 1 < nr) {
            __builtin_amdgcn_wave_barrier();
            { const float a0 = r0[R], a1 = r0[R + 1], a2 = r0[R + 2], a3 = r0[R + 3]; r0[0] = a0; r0[1] = a1; r0[2] = a2; r0[3] = a3; }
            if constexpr (C == 2) { const float a0 = r1[R], a1 = r1[R + 1], a2 = r1[R + 2], a3 = r1[R + 3]; r1[0] = a0; r1[1] = a1; r1[2] = a2; r1[3] = a3; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (AUDIT) {
        for (int o = 32; o; o >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, o)); acnt += __shfl_xor(acnt, o); }
        if (threadIdx.x == 0 && P.audit) { atomicMax(&P.audit[0], __float_as_uint(amax)); atomicAdd(&P.audit[1], acnt); }
    }
}

// ----------------------------------------------------------------- host side of the wave kernel
// bytes of a block per round: 32 for the rows kernels, 16 for the stream kernels (same-box A/B, 1024 × 10 s mono: rows 0.99 / 1.26 / 1.30 ms
// for 32 / 16 / 64, stream 3.16 / 2.53 / 4.88 ms — the stream kernel is bound by how many waves a CU holds, i.e. by its LDS tables).
// AUKIT_MS_RB=16|32 overrides both (A/B knob).
static const int MS_RB_ENV = getenv("AUKIT_MS_RB") ? atoi(getenv("AUKIT_MS_RB")) : 0;
static const int MS_RB_ROWS = MS_RB_ENV == 16 ? 16 : 32, MS_RB_STREAM = MS_RB_ENV == 32 ? 32 : 16;

// fills the coefficient table; false when a product sum could leave 32 bits (|s1 c1 + s2 c2| <= 32768 (|c1| + |c2|))
static bool ms_fill_coefs(const aukit_codec_desc *d, MsWaveParams &P) {
    if (d->ncoef > 0) { P.ncoef = std::min(d->ncoef, 32); for (int i = 0; i < P.ncoef; i++) { P.coef1[i] = d->coef1[i]; P.coef2[i] = d->coef2[i]; } }
    else { P.ncoef = 7; for (int i = 0; i < 7; i++) { P.coef1[i] = (short)ms_c1_default[i]; P.coef2[i] = (short)ms_c2_default[i]; } }
    for (int i = 0; i < P.ncoef; i++) if (std::abs((int)P.coef1[i]) + std::abs((int)P.coef2[i]) >= 65536) return false;
    return true;
}

template <int C, int RB, int MODE>
static void ms_launch_rows(const MsWaveParams &P, unsigned grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_ms_wave<C, RB, MODE, AUKIT_INTERP_NONE, false, float>), dim3(grid), dim3(64), lds, st, P);
}
// bytes of a block per fetch in the stream kernels.  Measured (1024 x 10 s mono, same box): fetching 64 bytes at a time and decoding them in four
// rounds of 16 halves the input re-fetch (PMC FETCH x 2: 1.29 -> 0.60 GB for 0.23 GB of blocks — with 16-byte fetches every 128-byte line comes from
// HBM several times) but costs 4 KB of LDS per wave, i.e. resident waves, and the kernel is bound by those: 2.52 -> 3.04 ms.  Time won: 16.
// Round 4: 128 = the register window (REGS in k_ms_wave): a line per lane at a time, no staging area — the re-fetch is gone (PMC FETCH x 2: 6.5 -> ?
// GB per 4096 x 216 blocks) without the LDS that made 64 lose.  -DAUKIT_MS_FB=16 for the A/B.
#ifndef AUKIT_MS_FB
#define AUKIT_MS_FB 128
#endif
constexpr int MS_FB_STREAM = AUKIT_MS_FB;
template <int C, int RB, bool MIX, typename OUT_T>
static void ms_launch_stream(int interp, const MsWaveParams &P, unsigned grid, size_t lds, hipStream_t st) {
    constexpr int FB = RB == 16 ? MS_FB_STREAM : RB;
    if constexpr (RB == 16 && sizeof(OUT_T) == 1) {   // the audited instantiations (AUKIT_OPT_COLLECT_STATS)
        if (P.audit) {
            if (interp == AUKIT_INTERP_LINEAR) hipLaunchKernelGGL((k_ms_wave<C, RB, MS_STREAM, AUKIT_INTERP_LINEAR, MIX, OUT_T, FB, true>), dim3(grid), dim3(64), lds, st, P);
            else if (interp == AUKIT_INTERP_CUBIC) hipLaunchKernelGGL((k_ms_wave<C, RB, MS_STREAM, AUKIT_INTERP_CUBIC, MIX, OUT_T, FB, true>), dim3(grid), dim3(64), lds, st, P);
            if (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC) return;
        }
    }
    if (interp == AUKIT_INTERP_NONE) hipLaunchKernelGGL((k_ms_wave<C, RB, MS_STREAM, AUKIT_INTERP_NONE, MIX, OUT_T, FB>), dim3(grid), dim3(64), lds, st, P);
    else if (interp == AUKIT_INTERP_LINEAR) hipLaunchKernelGGL((k_ms_wave<C, RB, MS_STREAM, AUKIT_INTERP_LINEAR, MIX, OUT_T, FB>), dim3(grid), dim3(64), lds, st, P);
    else hipLaunchKernelGGL((k_ms_wave<C, RB, MS_STREAM, AUKIT_INTERP_CUBIC, MIX, OUT_T, FB>), dim3(grid), dim3(64), lds, st, P);
}
static size_t ms_lds_bytes(int C, int rb, unsigned wt_floats, int fb = 0, size_t stage_bytes = 0) {
    if (fb < rb) fb = rb;
    if (rb != 16 && fb == 128) fb = rb;
    const int R = rb * 2 / C, ROW = 4 + R + 1, INS = fb == 128 ? 0 : (fb == 16 ? 4 : fb / 4 + 4);
    return (size_t)64 * INS * 4 + (size_t)C * 64 * ROW * 4 + 3 * 64 * 8 + 16 * 4 + (((size_t)wt_floats + 3) & ~(size_t)3) * 4 + stage_bytes;
}

struct MsBlocks { std::vector<uint64_t> blk0; uint64_t bps = 0; };
// per stream block counts (a partial last block is the reference's `str_byte` returning nil: an error, as before)
static int ms_count_blocks(const aukit_batch *in, const aukit_codec_desc *d, MsBlocks &B) {
    const int C = d->channels;
    if (C != 1 && C != 2) return fail(AUKIT_E_LUA, "Unsupported number of channels: %d", C);
    const uint64_t ba = (uint64_t)d->block_align;
    if (d->block_align < (C == 2 ? 15 : 8)) return fail(AUKIT_E_ARG, "bad blockAlign");
    B.blk0.assign((size_t)in->n + 1, 0);
    bool uni = in->n > 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s], nblk = (nb + ba - 1) / ba;
        if (nblk && nb % ba != 0) return fail(AUKIT_E_LUA, "bad argument #1 to 'rshift' (number expected, got nil)");
        B.blk0[s + 1] = B.blk0[s] + nblk;
        uni = uni && nblk > 0 && nblk == B.blk0[1];
    }
    B.bps = uni ? B.blk0[1] : 0;
    return AUKIT_OK;
}

int decode_msadpcm_audio(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, double new_rate, int interp, bool do_resample, int dtype,
                         aukit_audio **out) {
    if (d->sample_rate < 1) return fail(AUKIT_E_ARG, "bad argument #4 (number outside of range)");
    static const bool magic_ok = ms_magic_ok();
    MsWaveParams P{};
    const int C = d->channels;
    const bool wave = magic_ok && ms_fill_coefs(d, P) && (dtype == AUKIT_F64 || dtype == AUKIT_F32) && !getenv("AUKIT_MS_GENERIC") && (C == 1 || C == 2);
    if (wave) {
        MsBlocks B;
        int rc = ms_count_blocks(in, d, B);
        if (rc) return rc;
        const uint64_t ba = (uint64_t)d->block_align, nblocks = B.blk0[in->n];
        const uint64_t spb = C == 2 ? (ba - 14) + 2 : (ba - 7) * 2 + 2;
        std::vector<uint64_t> lens(in->n), row_off, row_len;
        for (uint32_t s = 0; s < in->n; s++) lens[s] = (B.blk0[s + 1] - B.blk0[s]) * spb;
        const size_t lds = ms_lds_bytes(C, MS_RB_ROWS, 0);
        std::vector<uint64_t> tab(B.blk0);
        uint64_t tot = 0;
        aukit_audio *a = nullptr;
        if (!do_resample) {
            a = *out;
            if ((rc = audio_prepare(ctx, &a, in->n, C, d->sample_rate, dtype, lens.data()))) return rc;
            *out = a;
        } else {   // int16 rows, one per (stream, channel), for the resampler behind
            row_off.assign((size_t)in->n * C, 0); row_len.assign((size_t)in->n * C, 0);
            std::vector<uint64_t> ro(in->n), rs(in->n);
            for (uint32_t s = 0; s < in->n; s++) {
                const uint64_t stride = round_up(std::max<uint64_t>(lens[s], 1), 8);
                ro[s] = tot; rs[s] = stride;
                for (int c = 0; c < C; c++) { row_off[(size_t)s * C + c] = tot + (uint64_t)c * stride; row_len[(size_t)s * C + c] = lens[s]; }
                tot += stride * C;
            }
            tab.insert(tab.end(), ro.begin(), ro.end());
            tab.insert(tab.end(), rs.begin(), rs.end());
            if ((rc = ctx->tmp_buf.ensure((size_t)tot * 2 + 64))) return rc;
        }
        tab.push_back(0);   // err[0], err[1]
        if (nblocks) {
            if ((rc = upload_table(ctx, ctx->tmp_buf2, tab.data(), tab.size() * 8))) return rc;
            const unsigned long long *t = reinterpret_cast<const unsigned long long *>(ctx->tmp_buf2.p);
            P.src = in->data(); P.safe_hi = in->base + in->cap;
            P.off = reinterpret_cast<const unsigned long long *>(in->d_off);
            P.blk0 = t; P.nstreams = in->n; P.nblocks = nblocks; P.bps = B.bps;
            P.block_align = d->block_align; P.spb_dec = (unsigned)spb;
            P.err = reinterpret_cast<int *>(const_cast<unsigned long long *>(t + tab.size() - 1));
            int mode;
            if (!do_resample) {
                const unsigned long long *m = reinterpret_cast<const unsigned long long *>(a->d_meta);
                P.rows = a->dev; P.row_off = m + in->n; P.row_stride = m + 2 * (size_t)in->n;
                mode = dtype == AUKIT_F64 ? MS_ROWS_F64 : MS_ROWS_F32;
            } else {
                P.rows = ctx->tmp_buf.p; P.row_off = t + in->n + 1; P.row_stride = t + 2 * (size_t)in->n + 1;
                mode = MS_ROWS_I16;
            }
            const unsigned grid = (unsigned)((nblocks + 63) / 64);
            if ((rc = ctx_begin_kernel(ctx))) return rc;
#define AUKIT_MS_ROWS(CC, RBB)                                                                                          \
            do {                                                                                                        \
                if (mode == MS_ROWS_F64) ms_launch_rows<CC, RBB, MS_ROWS_F64>(P, grid, lds, ctx->stream);               \
                else if (mode == MS_ROWS_F32) ms_launch_rows<CC, RBB, MS_ROWS_F32>(P, grid, lds, ctx->stream);          \
                else ms_launch_rows<CC, RBB, MS_ROWS_I16>(P, grid, lds, ctx->stream);                                   \
            } while (0)
            if (C == 1) { if (MS_RB_ROWS == 16) AUKIT_MS_ROWS(1, 16); else AUKIT_MS_ROWS(1, 32); }
            else { if (MS_RB_ROWS == 16) AUKIT_MS_ROWS(2, 16); else AUKIT_MS_ROWS(2, 32); }
#undef AUKIT_MS_ROWS
            AUKIT_HIP_CHECK(hipGetLastError());
            uint64_t out_elems = 0;
            for (uint64_t l : lens) out_elems += l * C;
            if ((rc = ctx_end_kernel(ctx, "k_ms_wave", in->total() + out_elems * (do_resample ? 2 : dtype_size(dtype))))) return rc;
            int herr[2] = {0, 0};
            AUKIT_HIP_CHECK(hipMemcpyAsync(herr, P.err, 8, hipMemcpyDeviceToHost, ctx->stream));
            AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (herr[0]) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (local 'c1')");  // predictor index beyond the coefficient table
            if (!do_resample) return AUKIT_OK;
            if (!herr[1]) return audio_from_int_rows(ctx, SRC_I16, ctx->tmp_buf.p, row_off, row_len, in->n, C, d->sample_rate, new_rate, interp, true, dtype, 32767, 32768, out);
            // a nan (delta overflowed to inf on garbage input) cannot sit in an int16 row: the rows of doubles below
        } else if (!do_resample) return AUKIT_OK;
    }
    std::vector<uint64_t> row_off, row_len, nblocks;
    uint64_t spb;
    int rc = msadpcm_rows(ctx, in, d, false, row_off, row_len, nblocks, &spb);
    if (rc) return rc;
    return audio_from_int_rows(ctx, SRC_AUDIO_F64, ctx->tmp_buf.p, row_off, row_len, in->n, d->channels, d->sample_rate, new_rate, interp, do_resample, dtype, 1, 1, out);
}

// One channel: every block of a stream decodes with the predictor index of the stream's FIRST block (Q9), so "is an index beyond the coefficient table" is one
// byte per stream.  The stream's first byte, to the host — on the look-ahead stream (round 6, late): the call then need not wait for k_ms_wave's own flag, and
// a host that issues call after call plans while the kernel of the call before runs.
__global__ __launch_bounds__(256) void k_ms_first_bytes(const unsigned char *src, const unsigned long long *off, unsigned n, unsigned char *out) {
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s < n) out[s] = off[s + 1] > off[s] ? src[off[s]] : (unsigned char)0;
}

// aukit.stream.msadpcm  aukit.lua:2588-2736
int stream_msadpcm(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out,
                   aukit_chunks **chunks_out) {
    const int C = d->channels;
    if (d->sample_rate < 1) return fail(AUKIT_E_ARG, "bad argument #4 (number outside of range)");
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "invalid interpolation");
    if (dtype != AUKIT_I8 && dtype != AUKIT_F64) return fail(AUKIT_E_ARG, "stream.msadpcm output must be AUKIT_I8 or AUKIT_F64");
    MsBlocks B;
    int rc = ms_count_blocks(in, d, B);
    if (rc) return rc;
    std::vector<unsigned char> first_byte;
    hipStream_t scan_stream = nullptr;
    if (C == 1 && in->n && !getenv("AUKIT_MS_SCAN_OFF")) {
        if ((rc = ctx_pre_stream(ctx, &scan_stream))) return rc;
        if (in->ready) AUKIT_HIP_CHECK(hipStreamWaitEvent(scan_stream, in->ready, 0));
        if ((rc = ctx->scan_buf.ensure((size_t)in->n + 64))) return rc;
        first_byte.resize(in->n);
        hipLaunchKernelGGL(k_ms_first_bytes, dim3((in->n + 255) / 256), dim3(256), 0, scan_stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off), in->n,
                           reinterpret_cast<unsigned char *>(ctx->scan_buf.p));
        AUKIT_HIP_CHECK(hipGetLastError());
        AUKIT_HIP_CHECK(hipMemcpyAsync(first_byte.data(), ctx->scan_buf.p, in->n, hipMemcpyDeviceToHost, scan_stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(scan_stream));   // (the look-ahead stream: not a wait for what the call before left on ctx->stream)
    }
    const uint64_t ba = (uint64_t)d->block_align;
    const uint64_t spb_dec = C == 2 ? (ba - 14) + 2 : (ba - 7) * 2 + 2;
    const double ratio = 48000 / d->sample_rate;
    const double samplesPerBlock = C == 2 ? (double)(ba - 14) : (double)(ba - 7) * 2;  // :2617 / :2682 (2 short, Q9)
    const double ips_d = std::ceil(d->sample_rate / samplesPerBlock);
    const double bytesPerSecond = (double)ba * ips_d;
    const uint32_t newlen = (uint32_t)std::max(0.0, std::floor(samplesPerBlock * ratio));
    const uint64_t ips = (uint64_t)ips_d;
    const int nd = (C == 2 && !mono) ? 2 : 1;
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0); ck->status.assign(in->n, 0); ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0);
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s], nblk = B.blk0[s + 1] - B.blk0[s];
        ck->length_seconds[s] = (double)(nb + ctx->sb_bytes) / (double)ba * samplesPerBlock / d->sample_rate;
        ck->nchunks[s] = newlen ? (uint32_t)((nblk + ips - 1) / ips) : 0;
        ck->max_chunks = std::max(ck->max_chunks, ck->nchunks[s]);
        lens[s] = nblk * newlen;
    }
    const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
    ck->lens.assign((size_t)ck->n * mc, 0);
    ck->pos.assign((size_t)ck->n * mc, 0);
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nblk = B.blk0[s + 1] - B.blk0[s];
        for (uint32_t k = 0; k < ck->nchunks[s]; k++) {
            const uint64_t done = std::min<uint64_t>((uint64_t)(k + 1) * ips, nblk), first = (uint64_t)k * ips;
            ck->lens[(size_t)s * mc + k] = (uint32_t)((done - first) * newlen);
            ck->pos[(size_t)s * mc + k] = ((double)(done * ba + ctx->sb_bytes + 1)) / bytesPerSecond;  // (n + pos) / bytesPerSecond
        }
    }
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, nd, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    auto done = [&]() { if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck; return AUKIT_OK; };
    const uint64_t nblocks = B.blk0[in->n];
    if (!nblocks || !newlen) return done();
    uint64_t out_elems = 0;
    for (uint64_t l : lens) out_elems += l * nd;
    // ---- the wave kernel: integer sample rates whose positions have at most 512 phases
    static const bool magic_ok = ms_magic_ok();
    MsWaveParams P{};
    bool wave = magic_ok && ms_fill_coefs(d, P) && interp != AUKIT_INTERP_SINC && !ctx->exact_math && !getenv("AUKIT_MS_GENERIC") &&
                d->sample_rate == std::floor(d->sample_rate) && d->sample_rate <= 4e9;
    unsigned long long fa = 0, fb = 0;
    if (wave) {
        unsigned long long x = 48000, y = (unsigned long long)d->sample_rate;
        while (y) { const unsigned long long tq = x % y; x = y; y = tq; }
        fa = (unsigned long long)d->sample_rate / x; fb = 48000 / x;   // x - 1 = (i - 1) / ratio = (i - 1) * fa / fb
        if (fb == 1) { fa *= 2; fb = 2; }
        wave = fb <= 512 && ((double)newlen * (double)fa + (double)fb) * (double)fb < 4294967296.0;
    }
    const int wf = interp == AUKIT_INTERP_CUBIC ? 4 : 1;
    const int MS_RB = MS_RB_STREAM;
    // int8 chunks: the round's outputs of every block wait in an LDS row (sst bytes, an odd number of dwords) before they leave
    unsigned sst = 0;
    // (stereo only.  Same-box A/B, 1024 ten-second streams, pairs / lanes: stereo 2.98 / 2.81 ms, stereo mixed to mono 2.40 / 2.11, mono 2.55 / 3.26 —
    // PMC for mono: 828 M / 963 M VALU instructions, LDS bank conflicts 78 M / 0: the copy loop that brings the staged bytes out costs a
    // mono round more than the bookkeeping it saves, AUKIT_MS_LANES=1 forces it)
    if (wave && dtype == AUKIT_I8 && !getenv("AUKIT_MS_PAIRS") && (C == 2 || getenv("AUKIT_MS_LANES"))) {
        const unsigned long long Rr = (unsigned long long)(MS_RB * 2 / C);
        const unsigned long long njmax = (Rr * fb + fa - 1) / fa + 2;   // outputs whose floor(x) falls into one round's samples
        sst = (unsigned)((((njmax + 3) / 4) | 1) * 4);
        if ((size_t)nd * 64 * sst > 24 * 1024) sst = 0;                // strong up-sampling: the pair-mapped loop
    }
    wave = wave && spb_dec < (1u << 17) && newlen < (1u << 26);   // (the deferred-line records of k_ms_wave pack q0 and j into 17 and 26 bits)
    size_t lds = wave ? ms_lds_bytes(C, MS_RB, (unsigned)fb * wf, MS_FB_STREAM, (((size_t)nd * 64 * sst + 7) & ~(size_t)7) + ((MS_RB == 16 && MS_FB_STREAM == 128) ? 0 : (size_t)MS_DL_CAP * 8)) : 0;
    // (A/B only: what a per-block staging row for whole output lines would cost in resident waves — unused bytes of LDS per workgroup,
    // profiles/r05_msadpcm_lds_ab.txt)
    if (const char *e = getenv("AUKIT_MS_EXTRA_LDS")) lds += (size_t)std::max(0, atoi(e));
    if (wave && lds <= 64 * 1024) {
        const int R = MS_RB * 2 / C, ndata = d->block_align - 7 * C, nr = (ndata + MS_RB - 1) / MS_RB;
        // tables: blk0 (n + 1) | out_off (n) | out_stride (n) | err | rounds (nr) | weights (fb * wf, f32)
        std::vector<uint64_t> tab(B.blk0);
        tab.insert(tab.end(), a->row_off.begin(), a->row_off.end());
        tab.insert(tab.end(), a->row_stride.begin(), a->row_stride.end());
        tab.push_back(0);
        const size_t jl_at = tab.size();
        std::vector<MsRound> rt((size_t)nr);
        for (int r = 0; r < nr; r++) {
            const unsigned long long lo = std::min<unsigned long long>(((unsigned long long)R * r * fb + fa - 1) / fa, newlen);
            const unsigned long long hi = r + 1 == nr ? newlen : std::min<unsigned long long>(((unsigned long long)R * (r + 1) * fb + fa - 1) / fa, newlen);
            MsRound &q = rt[r];
            memset(&q, 0, sizeof q);
            q.jl = (unsigned)lo; q.nj = (unsigned)(hi - lo);
            if (q.nj) {
                q.dq = 64u / q.nj; q.dr = 64u % q.nj;
                q.aq = (unsigned)(((unsigned long long)q.dr * fa) / fb); q.ar = (unsigned)(((unsigned long long)q.dr * fa) % fb);
                q.nq = (unsigned)(((unsigned long long)q.nj * fa) / fb); q.nrm = (unsigned)(((unsigned long long)q.nj * fa) % fb);
                q.b0q = q.nj < 64 ? 1 : 0;
            }
        }
        std::vector<float> w((size_t)fb * wf);
        for (unsigned r = 0; r < fb; r++) {
            const long double f = (long double)r / (long double)fb, f2 = f * f, f3 = f2 * f;
            if (wf == 1) w[r] = (float)f;
            else {
                w[4 * r] = (float)(-0.5L * f3 + f2 - 0.5L * f); w[4 * r + 1] = (float)(1.5L * f3 - 2.5L * f2 + 1.0L);
                w[4 * r + 2] = (float)(-1.5L * f3 + 2.0L * f2 + 0.5L * f); w[4 * r + 3] = (float)(0.5L * f3 - 0.5L * f2);
            }
        }
        const size_t jl_words = (rt.size() * sizeof(MsRound) + 7) / 8, w_words = (w.size() * 4 + 7) / 8;
        tab.resize(jl_at + jl_words + w_words, 0);
        memcpy(&tab[jl_at], rt.data(), rt.size() * sizeof(MsRound));
        memcpy(&tab[jl_at + jl_words], w.data(), w.size() * 4);
        if ((rc = upload_table(ctx, ctx->tmp_buf2, tab.data(), tab.size() * 8))) { delete ck; return rc; }
        const unsigned long long *t = reinterpret_cast<const unsigned long long *>(ctx->tmp_buf2.p);
        P.src = in->data(); P.safe_hi = in->base + in->cap;
        P.off = reinterpret_cast<const unsigned long long *>(in->d_off);
        P.blk0 = t; P.nstreams = in->n; P.nblocks = nblocks; P.bps = B.bps;
        P.block_align = d->block_align; P.spb_dec = (unsigned)spb_dec;
        P.out = a->dev; P.out_off = t + in->n + 1; P.out_stride = t + 2 * (size_t)in->n + 1;
        P.err = reinterpret_cast<int *>(const_cast<unsigned long long *>(t + jl_at - 1));
        P.rounds = reinterpret_cast<const MsRound *>(t + jl_at);
        P.wg = reinterpret_cast<const float *>(t + jl_at + jl_words);
        P.newlen = newlen; P.spb = (unsigned)samplesPerBlock; P.fa = (unsigned)fa; P.fb = (unsigned)fb;
        P.fmagic = (unsigned)((4294967296ull + fb - 1) / fb); P.inv_fb = 1.0 / (double)fb;
        P.ratio = ratio; P.rcp = 1.0 / ratio;
        P.exact_rcp = exact_div_verified(ctx, ratio, (uint64_t)newlen + 2) ? 1 : 0;
        P.unit = d->sample_rate == 48000 ? 1 : 0;
        P.sst = sst;
        P.audit = nullptr;
        if (ctx->collect_stats && dtype == AUKIT_I8 && MS_RB == 16 && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {
            if ((rc = ctx->fmt_flag.ensure(64))) { delete ck; return rc; }
            P.audit = reinterpret_cast<unsigned *>(ctx->fmt_flag.p) + 8;
            if (hipMemsetAsync(P.audit, 0, 8, ctx->stream) != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "hipMemsetAsync failed"); }
        }
        const unsigned grid = (unsigned)((nblocks + 63) / 64);
        bool scanned = false;
        if (scan_stream) {   // the predictor indices of the streams that have a block at all: the reference's error, before anything is decoded
            for (uint32_t s2 = 0; s2 < in->n; s2++)
                if (B.blk0[s2 + 1] > B.blk0[s2] && (int)first_byte[s2] >= P.ncoef) { delete ck; return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (local 'c1')"); }
            scanned = true;
        }
        if ((rc = ctx_begin_kernel(ctx))) { delete ck; return rc; }
#define AUKIT_MS_STREAM(CC, RBB, MIXX)                                                                                                 \
        do {                                                                                                                            \
            if (dtype == AUKIT_I8) ms_launch_stream<CC, RBB, MIXX, signed char>(interp, P, grid, lds, ctx->stream);                    \
            else ms_launch_stream<CC, RBB, MIXX, double>(interp, P, grid, lds, ctx->stream);                                           \
        } while (0)
        if (C == 1) { if (MS_RB == 16) AUKIT_MS_STREAM(1, 16, false); else AUKIT_MS_STREAM(1, 32, false); }
        else if (mono) { if (MS_RB == 16) AUKIT_MS_STREAM(2, 16, true); else AUKIT_MS_STREAM(2, 32, true); }
        else { if (MS_RB == 16) AUKIT_MS_STREAM(2, 16, false); else AUKIT_MS_STREAM(2, 32, false); }
#undef AUKIT_MS_STREAM
        if (hipGetLastError() != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "k_ms_wave launch failed"); }
        if ((rc = ctx_end_kernel(ctx, "k_ms_wave", in->total() + out_elems * dtype_size(dtype)))) { delete ck; return rc; }
        int herr = 0;
        // (scanned: no planned block can raise the kernel's flag — the host looked at every index it will meet; AUKIT_MS_ASSERT=1 waits and looks all the same)
        if (!scanned || P.audit || getenv("AUKIT_MS_ASSERT"))
        if (hipMemcpyAsync(&herr, P.err, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "k_ms_wave failed"); }
        if (P.audit) {
            unsigned h[2] = {0, 0};
            if (hipMemcpy(h, P.audit, 8, hipMemcpyDeviceToHost) != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "audit read-back failed"); }
            float e; memcpy(&e, &h[0], 4);
            ctx->counters[AUKIT_COUNTER_TIER1_ERR_NANO] = (uint64_t)std::llround((double)e * 1e9);
            ctx->counters[AUKIT_COUNTER_TIER1_OUTPUTS] = h[1];
        }
        if (herr) { delete ck; return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (local 'c1')"); }
        return done();
    }
    // ---- generic: rows of doubles, then the segment resampler with its floor epilogue
    std::vector<uint64_t> row_off, row_len, nblk;
    uint64_t spb_chk;
    if ((rc = msadpcm_rows(ctx, in, d, true, row_off, row_len, nblk, &spb_chk))) { delete ck; return rc; }
    // one segment per block; its C source rows are consecutive entries of the row table
    std::vector<Seg> segs;
    std::vector<uint64_t> blkrows;
    for (uint32_t s = 0; s < in->n; s++) {
        for (uint64_t b = 0; b < nblk[s]; b++) {
            Seg g;
            g.src_base = -1; g.w_lo = 1; g.w_hi = (int)spb_dec; g.n_out = newlen;
            // stereo keeps the block before at indices -N .. -1 (`left[i-#lastL-1] = lastL[i]`, :2640-2643; `lastL` outlives the iterator call): only
            // sinc reaches below index 1 — the rows of a stream's blocks are consecutive, so those entries are the row's entries 1 - N .. 0
            if (C == 2 && interp == AUKIT_INTERP_SINC && b > 0) g.w_lo = 1 - (int)spb_dec;
            g.stream = (unsigned)(blkrows.size() / C);
            g.out_off = a->row_off[s] + b * newlen;
            g.out_stride = (unsigned)a->row_stride[s];
            g.pad = 0;
            for (int c = 0; c < C; c++) blkrows.push_back(row_off[(size_t)s * C + c] + b * spb_dec);
            segs.push_back(g);
        }
    }
    if (!segs.empty()) {
        if ((rc = upload_table(ctx, ctx->misc_buf, blkrows.data(), blkrows.size() * 8))) { delete ck; return rc; }
        ResampleParams RP;
        memset(&RP, 0, sizeof RP);
        RP.src = reinterpret_cast<const unsigned char *>(ctx->tmp_buf.p);
        RP.src_off = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
        RP.channels = C;
        RP.mix_mono = (C == 2 && mono) ? 2 : 0;
        RP.sinc_hole = (C == 2 && interp == AUKIT_INTERP_SINC) ? 1 : 0;
        RP.out = a->dev;
        size_t glds;
        if ((rc = plan_tiles(ctx, segs, ratio, interp, C, RP, &glds))) { delete ck; return rc; }
        rc = launch_resample(ctx, SRC_AUDIO_F64, interp, EPI_STREAM_FLOOR, dtype, RP, glds, in->total() + out_elems * dtype_size(dtype), nullptr);
        if (rc) { delete ck; return rc; }
    }
    return done();
}

}  // namespace aukit
