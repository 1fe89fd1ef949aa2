// qoa.hip — QOA: aukit.qoa (aukit.lua:1706-1777) and aukit.stream.qoa (aukit.lua:3202-3337), round 3.
//
// What changed against round 2 (codecs2.hip / qoa_stream.hip, which stay as the fallback for shapes this file does not take):
//   * the frame headers are walked ON THE DEVICE (k_qoa_walk: one lane per stream follows the reference's loops — `read(8)`, the header
//     checks, `frame_size`, the slice count — twice: count, then fill), and only the frame list (16 bytes per 5120-sample frame) travels to
//     the host for the chunk plan.  Round 2 copied the whole batch to the host for that: 360 MB per 1024 stereo streams, most of its time;
//   * the LMS decoder (k_qoa_wave) is a wave kernel in the sense of msadpcm.hip: a wave owns 64 consecutive (frame, channel) jobs, the
//     slices of a round (four per job) arrive by 8-byte loads in slice order (neighbouring lanes read neighbouring slices of the same frame:
//     whole cache lines per instruction) into an LDS staging area, every lane then decodes its own four slices — int32 throughout: only
//     the low 32 bits of the prediction sum reach `signed_rshift` (:1681-1689), weights stay below 2^23 within a frame of <= 8192 samples,
//     history is int16, so four v_mad_i32_i24 are the Lua's sum — and the 80 decoded samples per job leave through LDS as 8-byte stores
//     along each job's row (coalesced rows instead of 40 bytes per lane at a 10 KiB stride);
//   * stream.qoa stores `math_floor(reconstructed / 256)` (:3299) as int8 rows (one byte per sample of intermediate) and runs its
//     resample + recursive low-pass (+ channel mean) tail in ONE launch from them (k_iir_tail, stream_tail.hip).
#include <algorithm>
#include <chrono>
#include "resample.h"
#include "stream_tail.h"

namespace aukit {

int audio_from_int_rows(aukit_ctx *ctx, int src_kind, const void *rows_dev, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len,
                        uint32_t n, int channels, double rate, double new_rate, int interp, bool do_resample, int dtype, double norm_pos,
                        double norm_neg, aukit_audio **out);
// round 2's implementations (host-side header walk, lane-per-job kernels): the fallback
int decode_qoa_audio_host(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, double new_rate, int interp, bool do_resample, int dtype, aukit_audio **out);
int stream_qoa_host(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks);

struct QoaJob { unsigned long long frame_off, out_off; int c, channels, samples, emit; };   // as in codecs2.hip

__constant__ int c_qoa_deq[16][8] = {
    {1, -1, 3, -3, 5, -5, 7, -7}, {5, -5, 18, -18, 32, -32, 49, -49}, {16, -16, 53, -53, 95, -95, 147, -147},
    {34, -34, 113, -113, 203, -203, 315, -315}, {63, -63, 210, -210, 378, -378, 588, -588}, {104, -104, 345, -345, 621, -621, 966, -966},
    {158, -158, 528, -528, 950, -950, 1477, -1477}, {228, -228, 760, -760, 1368, -1368, 2128, -2128},
    {316, -316, 1053, -1053, 1895, -1895, 2947, -2947}, {422, -422, 1405, -1405, 2529, -2529, 3934, -3934},
    {548, -548, 1828, -1828, 3290, -3290, 5117, -5117}, {696, -696, 2320, -2320, 4176, -4176, 6496, -6496},
    {868, -868, 2893, -2893, 5207, -5207, 8099, -8099}, {1064, -1064, 3548, -3548, 6386, -6386, 9933, -9933},
    {1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005}, {1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336}};  // :1662-1679

// ================================================================= the frame walk
// One lane per stream follows the reference's loops — mode 0: aukit.qoa's (:1727-1775); mode 1: stream.qoa's iterator calls (:3256-3308)
// one after the other — statement for statement (cf. qoa_scan in codecs2.hip, the host version of round 2).  Two passes:
//   count (FILL false): per stream the file header's 12 bytes, how many calls deliver a chunk, how many (frame, channel) decode jobs they
//     hold, how many row elements they need, whether the walk raised;
//   fill (FILL true): with the host's prefix sums of those, the decode jobs themselves (they stay on the device) and one record per call
//     (row, stride, #table, samples) for the host's chunk plan.
// A call is walked dry first (its row stride depends on its last frame), then again to emit.
struct QoaWalkOut { unsigned char head[12]; unsigned ncalls, raised, big, pad; unsigned long long njobs, rows_total, L; };
struct QoaFillIn { unsigned long long job_first, call_first, row_base, stride; };
struct QoaCallRec { unsigned long long row0; unsigned stride, n, sample_pos, pad; };
static_assert(sizeof(QoaWalkOut) == 56 && sizeof(QoaFillIn) == 32 && sizeof(QoaCallRec) == 24, "walk records");

struct QoaCallWalk { unsigned long long end_pos, n, sample_pos; unsigned nframes; bool raised, big; };
// one iterator call (mode 1) or the whole file (mode 0) from `pos`; emit(frame_pos, samples, sp) per accepted frame
template <class Emit>
static AUKIT_DEV QoaCallWalk qoa_walk_call(const unsigned char *h, unsigned long long nb, unsigned long long pos, int mode, int fc, unsigned fr, double file_samples, Emit emit) {
    QoaCallWalk w{pos, 0, 0, 0, false, false};
    auto header_ok = [&](const unsigned char *f, int &channels, int &samples, int &frame_size) {
        channels = f[0];
        const unsigned rate = (unsigned)f[1] << 16 | (unsigned)f[2] << 8 | f[3];
        samples = f[4] << 8 | f[5]; frame_size = f[6] << 8 | f[7];
        const int data_size = frame_size - 8 - 16 * channels;
        const int num_slices = data_size >= 0 ? data_size / 8 : -((-data_size + 7) / 8);   // math_floor(data_size / 8)
        return !(channels != fc || rate != fr || samples * channels > num_slices * 20);
    };
    double sample_pos = 0;
    unsigned long long sp = 0;
    for (;;) {
        int channels, samples, frame_size;
        unsigned long long fpos, need;
        if (mode == 0) {
            if (!(pos + 1 + 16ull * fc + 8 <= nb && sample_pos < file_samples)) break;
            if (!header_ok(h + pos, channels, samples, frame_size)) break;
            if ((double)frame_size > (double)nb - (double)(pos + 8)) break;   // frame_size > #data - pos + 1, pos after the header (Q18)
            need = 8 + 16ull * channels + 8ull * (unsigned long long)((samples + 19) / 20) * channels;
            if (pos + need > nb) { w.raised = true; break; }                  // "data string too short"
            fpos = pos;
            pos += need;
        } else {
            if (!(sample_pos < (double)fr)) break;
            if (pos >= nb) break;                                              // read(8) → nil
            if (pos + 8 > nb) { w.raised = true; break; }
            fpos = pos;
            const bool ok = header_ok(h + pos, channels, samples, frame_size);
            pos += 8;
            if (!ok) break;                                                    // :3270-3277 (the header is consumed)
            need = 16ull * channels + 8ull * (unsigned long long)((samples + 19) / 20) * channels;
            if (pos + need > nb) { w.raised = true; break; }                   // assert(read(8), "Invalid QOA data") / short unpack
            pos += need;
        }
        emit(fpos, samples, sp);
        w.nframes++;
        if (samples > 8192) w.big = true;
        const unsigned long long top = sp + (unsigned long long)((samples + 19) / 20) * 20;
        w.n = top > w.n ? top : w.n;
        sp += (unsigned long long)samples;
        sample_pos += samples;
    }
    w.end_pos = pos;
    w.sample_pos = sp;
    return w;
}

template <bool FILL>
__global__ __launch_bounds__(64) void k_qoa_walk(const unsigned char *src, const unsigned long long *off, unsigned n, int mode, QoaWalkOut *wo,
                                                const QoaFillIn *fin, QoaJob *jobs, QoaCallRec *calls) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const unsigned char *h = src + off[s];
    const unsigned long long nb = off[s + 1] - off[s];
    QoaWalkOut o;
    for (int i = 0; i < 12; i++) o.head[i] = (unsigned long long)i < nb ? h[i] : 0;
    o.ncalls = 0; o.raised = 0; o.big = 0; o.pad = 0; o.njobs = 0; o.rows_total = 0; o.L = 0;
    if (nb >= 12 && h[0] == 'q' && h[1] == 'o' && h[2] == 'a' && h[3] == 'f') {
        const double file_samples = (double)((unsigned)h[4] << 24 | (unsigned)h[5] << 16 | (unsigned)h[6] << 8 | h[7]);
        const int fc = h[8];
        const unsigned fr = (unsigned)h[9] << 16 | (unsigned)h[10] << 8 | h[11];
        if (fc >= 1 && fc <= AUKIT_MAX_PLANAR_CHANNELS) {
            unsigned long long pos = 8, jat = 0, rat = 0;
            if (FILL) { jat = fin[s].job_first; rat = fin[s].row_base; }
            for (;;) {   // mode 1: one iterator call per turn
                const QoaCallWalk w = qoa_walk_call(h, nb, pos, mode, fc, fr, file_samples, [](unsigned long long, int, unsigned long long) {});
                if (w.big) o.big = 1;
                if (w.raised) { o.raised = 1; break; }
                if (mode == 1 && w.n == 0) break;                              // #chunk[1] == 0 → nil
                const unsigned long long stride = mode == 0 ? (FILL ? fin[s].stride : 0) : ((w.n + 2 + 15) / 16) * 16;
                if (FILL) {
                    const unsigned nfr = w.nframes;
                    unsigned k = 0;
                    qoa_walk_call(h, nb, pos, mode, fc, fr, file_samples, [&](unsigned long long fpos, int samples, unsigned long long sp) {
                        const bool lastf = ++k == nfr;
                        for (int c = 0; c < fc; c++) {
                            QoaJob j;
                            j.frame_off = off[s] + fpos; j.out_off = rat + (unsigned long long)c * stride + sp;
                            j.c = c; j.channels = fc; j.samples = samples;
                            j.emit = lastf ? ((samples + 19) / 20) * 20 : samples;   // Q15: the ≤19-sample tail survives only after the last frame of a table
                            jobs[jat++] = j;
                        }
                    });
                    QoaCallRec cr;
                    cr.row0 = rat; cr.stride = (unsigned)stride; cr.n = (unsigned)w.n; cr.sample_pos = (unsigned)w.sample_pos; cr.pad = 0;
                    calls[fin[s].call_first + o.ncalls] = cr;
                    rat += stride * (unsigned long long)fc;
                }
                o.njobs += (unsigned long long)w.nframes * fc;
                o.rows_total += stride * (unsigned long long)fc;
                o.L = w.n;
                o.ncalls++;
                pos = w.end_pos;
                if (mode == 0) break;
            }
        }
    }
    if (!FILL) wo[s] = o;
}

// ================================================================= the wave decoder
static AUKIT_DEV int qoa_mul24(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
static AUKIT_DEV int qoa_mad24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
static AUKIT_DEV int qoa_med3(int a, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi)); return r; }

#ifndef AUKIT_QOA_K
#define AUKIT_QOA_K 4
#endif
// S8: stream.qoa's math_floor(reconstructed / 256) as int8 rows (:3299); else the reconstructed int16 (aukit.qoa, :1765 divides later)
template <bool S8>
__global__ __launch_bounds__(64) void k_qoa_wave(const unsigned char *src, const QoaJob *jobs, unsigned long long njobs, void *out) {
    constexpr int K = AUKIT_QOA_K;                // slices per job and round
    constexpr int OB = S8 ? 1 : 2;                // bytes per stored sample
    constexpr int OSTR = K * 20 * OB + 8;         // bytes between two jobs' output areas
    constexpr int UPJ = K * 20 * OB / 8;          // 8-byte units per job and round
    // (the fetched slices and the decoded samples share their LDS: a lane takes its K slices into registers before the first sample is written — 2.6 KB
    // less per wave, and what bounds this kernel's resident waves is its LDS: PMC had its waves waiting to issue 41 % of their time, VALU 69 % busy)
    constexpr int OBYTES = 64 * (K * 20 * OB + 8), SBYTES = 64 * (K + 1) * 8;
    __shared__ __attribute__((aligned(16))) unsigned char io_buf[OBYTES > SBYTES ? OBYTES : SBYTES];
    unsigned long long *const sl_in = reinterpret_cast<unsigned long long *>(io_buf);
    __shared__ unsigned long long j_src[64], j_out[64];
    __shared__ unsigned j_step[64], j_ns[64];
    __shared__ int j_emit[64];
    __shared__ int deq[128];
    unsigned char *const obuf = io_buf;
    const int lane = threadIdx.x;
    const unsigned long long j = (unsigned long long)blockIdx.x * 64 + lane;
    QoaJob job;
    if (j < njobs) job = jobs[j];
    else { job.frame_off = 0; job.out_off = 0; job.c = 0; job.channels = 1; job.samples = 0; job.emit = 0; }
    const unsigned ns = (unsigned)((job.samples + 19) / 20);
    j_src[lane] = job.frame_off + 8 + 16ull * job.channels + 8ull * job.c;
    j_step[lane] = 8u * (unsigned)job.channels;
    j_ns[lane] = ns;
    j_out[lane] = job.out_off;
    j_emit[lane] = job.emit;
    deq[lane] = (&c_qoa_deq[0][0])[lane];
    deq[64 + lane] = (&c_qoa_deq[0][0])[64 + lane];
    // LMS state: 4 × int16 history, 4 × int16 weights, big-endian (:1743-1748)
    int h0 = 0, h1 = 0, h2 = 0, h3 = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (ns) {
        typedef unsigned u32x4u __attribute__((ext_vector_type(4), aligned(1)));
        const u32x4u q = *reinterpret_cast<const u32x4u *>(src + job.frame_off + 8 + 16ull * job.c);
        auto be16 = [](unsigned w, int hi) { const unsigned b = __builtin_bswap32(w); return (int)(short)(hi ? (b >> 16) : (b & 0xFFFF)); };
        h0 = be16(q.x, 1); h1 = be16(q.x, 0); h2 = be16(q.y, 1); h3 = be16(q.y, 0);
        w0 = be16(q.z, 1); w1 = be16(q.z, 0); w2 = be16(q.w, 1); w3 = be16(q.w, 0);
    }
    int g0 = (h0 >> 31) | 1, g1 = (h1 >> 31) | 1, g2 = (h2 >> 31) | 1, g3 = (h3 >> 31) | 1;   // the history samples' signs
    unsigned nrmax = (ns + K - 1) / K;
    for (int o = 32; o > 0; o >>= 1) nrmax = max(nrmax, (unsigned)__shfl_xor((int)nrmax, o));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    typedef unsigned u32x2u __attribute__((ext_vector_type(2), aligned(1)));
    auto fetch = [&](unsigned r, u32x2u (&pf)[K]) {
#pragma unroll
        for (int i = 0; i < K; i++) {
            const unsigned u = (unsigned)i * 64 + lane, jb = u / K, sj = u % K, sl = r * K + sj;
            u32x2u v = {0, 0};
            if (sl < j_ns[jb]) v = *reinterpret_cast<const u32x2u *>(src + j_src[jb] + (unsigned long long)sl * j_step[jb]);
            pf[i] = v;
        }
    };
    u32x2u pf[K];
    if (nrmax) fetch(0, pf);
    const int lo16 = -32768, hi16 = 32767;
    for (unsigned r = 0; r < nrmax; r++) {
#pragma unroll
        for (int i = 0; i < K; i++) {
            const unsigned u = (unsigned)i * 64 + lane, jb = u / K, sj = u % K;
            sl_in[jb * (K + 1) + sj] = (unsigned long long)pf[i].x | (unsigned long long)pf[i].y << 32;
        }
        if (r + 1 < nrmax) fetch(r + 1, pf);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned long long raws[K];
#pragma unroll
        for (int sj = 0; sj < K; sj++) raws[sj] = sl_in[lane * (K + 1) + sj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (every lane has its slices: the samples may overwrite them)
#pragma unroll
        for (int sj = 0; sj < K; sj++) {
            const unsigned long long raw = raws[sj];
            const unsigned hi = __builtin_bswap32((unsigned)raw), lo = __builtin_bswap32((unsigned)(raw >> 32));   // (">I4I4"):unpack
            const int *dq = deq + (hi >> 28) * 8;   // scalefactor = bit32_extract(sliceH, 28, 4)
            unsigned pk[S8 ? 5 : 10];
#pragma unroll
            for (int k = 0; k < 20; k++) {
                const int sum = qoa_mad24(w3, h3, qoa_mad24(w2, h2, qoa_mad24(w1, h1, qoa_mul24(w0, h0))));   // the low 32 bits of the Lua's double sum
                const int predicted = sum >> 13;                                                              // signed_rshift(..., 13)  :1686-1689
                // quantized = bit32_extract(sliceH, 25, 3), then the slice moves three bits up (:1752-1760): residual k sits at bits 57 - 3 k .. 59 - 3 k of
                // the 64, a constant once the loop is unrolled — one v_bfe_u32 instead of a shift, a mask and the two-word shift (k = 9 straddles the words)
                const int bp = 57 - 3 * k;
                unsigned qi;
                if (bp >= 32) qi = __builtin_amdgcn_ubfe(hi, (unsigned)(bp - 32), 3u);
                else if (bp + 3 <= 32) qi = __builtin_amdgcn_ubfe(lo, (unsigned)bp, 3u);
                else qi = __builtin_amdgcn_alignbit(hi, lo, (unsigned)bp) & 7u;
                const int d = dq[qi];
                const int rec = qoa_med3(predicted + d, lo16, hi16);                                         // :1763
                if constexpr (S8) {   // math_floor(reconstructed / 256) = byte 1 of the int32  (:3299): one v_perm_b32 drops it into its place
                    constexpr unsigned SEL[4] = {0x0c0c0c05u, 0x0c0c0500u, 0x0c050100u, 0x05020100u};
                    pk[k >> 2] = __builtin_amdgcn_perm((unsigned)rec, (k & 3) ? pk[k >> 2] : 0u, SEL[k & 3]);
                } else {
                    if (k & 1) pk[k >> 1] = __builtin_amdgcn_perm((unsigned)rec, pk[k >> 1], 0x05040100u); else pk[k >> 1] = (unsigned)rec;   // (the upper half: overwritten by the odd sample)
                }
                const int delta = d >> 4;                                                                     // signed_rshift(residual, 4)
                // weights[i] += history[i] < 0 and -delta or delta  (:1694-1699) as ONE v_mad_i32_i24 each: the history sample's sign (+-1, made once when
                // the sample is, two instructions) times delta (|delta| < 2^11) — a compare, four selects and four adds otherwise: 10 of the 21 per sample
                w0 = qoa_mad24(g0, delta, w0); w1 = qoa_mad24(g1, delta, w1); w2 = qoa_mad24(g2, delta, w2); w3 = qoa_mad24(g3, delta, w3);
                h0 = h1; h1 = h2; h2 = h3; h3 = rec;
                g0 = g1; g1 = g2; g2 = g3; g3 = (rec >> 31) | 1;
            }
            unsigned *ow = reinterpret_cast<unsigned *>(obuf + lane * OSTR + sj * 20 * OB);
#pragma unroll
            for (int q = 0; q < (S8 ? 5 : 10); q++) ow[q] = pk[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the round's K * 20 samples of every job, 8 bytes per lane along each job's row
#pragma unroll 2
        for (int i = 0; i < UPJ; i++) {
            const unsigned u = (unsigned)i * 64 + lane, jb = u / UPJ, v = u % UPJ;
            constexpr int SPU = 8 / OB;   // samples per unit
            const int sb = (int)(r * (K * 20)) + (int)v * SPU, em = j_emit[jb];
            if (sb < em) {
                const unsigned long long bits = *reinterpret_cast<const unsigned long long *>(obuf + jb * OSTR + v * 8);
                unsigned char *dstp = reinterpret_cast<unsigned char *>(out) + (j_out[jb] + (unsigned long long)sb) * OB;
                if (sb + SPU <= em) {
                    typedef unsigned u32x2o __attribute__((ext_vector_type(2), aligned(1)));
                    u32x2o w; w.x = (unsigned)bits; w.y = (unsigned)(bits >> 32);
                    *reinterpret_cast<u32x2o *>(dstp) = w;
                } else {
                    for (int e = 0; e < (em - sb) * OB; e++) dstp[e] = (unsigned char)(bits >> (8 * e));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ================================================================= host side
struct QoaLaps {   // AUKIT_HOST_TIMING=1: host laps on stderr
    bool on = getenv("AUKIT_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *w) { if (on) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[qoa host] %-14s %8.1f us\n", w, std::chrono::duration<double, std::micro>(t - t0).count()); t0 = t; } }
};
struct QoaStreamInfo { int channels; double rate, file_samples; bool raised, big; unsigned ncalls; uint64_t njobs, rows_total, L, job_first, call_first, row_base, stride; };

// pass 1 of the device walk; validates the file headers like the reference (errors with its strings)
// (wb: where the walks keep their words — tmp_buf3, or the call's set of ctx->qoa_set when they run on the look-ahead stream: stream_qoa)
static int qoa_walk_count(aukit_ctx *ctx, const aukit_batch *in, int mode, std::vector<QoaStreamInfo> &S, DevBuf *wb = nullptr) {
    const uint32_t n = in->n;
    S.assign(n, QoaStreamInfo{});
    if (!n) return AUKIT_OK;
    DevBuf &W = wb ? *wb : ctx->tmp_buf3;
    int rc = W.ensure((size_t)n * (sizeof(QoaWalkOut) + sizeof(QoaFillIn)) + 64);
    if (rc) return rc;
    QoaWalkOut *dwo = reinterpret_cast<QoaWalkOut *>(W.p);
    const unsigned long long *doff = reinterpret_cast<const unsigned long long *>(in->d_off);
    hipLaunchKernelGGL((k_qoa_walk<false>), dim3((n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), doff, n, mode, dwo, static_cast<const QoaFillIn *>(nullptr),
                       static_cast<QoaJob *>(nullptr), static_cast<QoaCallRec *>(nullptr));
    AUKIT_HIP_CHECK(hipGetLastError());
    // (read-backs go through the context's pinned staging buffer: a pageable destination makes the runtime pin it on the fly — 26 ms per
    // call for a megabyte, measured, against 0.4 ms)
    std::vector<QoaWalkOut> wo(n);
    {
        void *st = ctx_host_stage(ctx, (size_t)n * sizeof(QoaWalkOut));
        AUKIT_HIP_CHECK(hipMemcpyAsync(st ? st : wo.data(), dwo, (size_t)n * sizeof(QoaWalkOut), hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (st) memcpy(wo.data(), st, (size_t)n * sizeof(QoaWalkOut));
    }
    uint64_t jat = 0, cat = 0, rat = 0;
    for (uint32_t s = 0; s < n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        const unsigned char *h = wo[s].head;
        // aukit.qoa: (">c4I4"):unpack / (">BI3"):unpack on a short string raise; stream.qoa: assert(read(8), ...) / assert(peek(4), ...)
        if (nb < 8) return fail(AUKIT_E_LUA, mode == 0 ? "data string too short" : "Not a QOA file");
        if (memcmp(h, "qoaf", 4) != 0) return fail(AUKIT_E_ARG, "Not a QOA file");
        if (nb < 12) return fail(AUKIT_E_LUA, nb == 8 && mode == 1 ? "Not a QOA file" : "data string too short");
        QoaStreamInfo &q = S[s];
        q.file_samples = (double)((uint32_t)h[4] << 24 | (uint32_t)h[5] << 16 | (uint32_t)h[6] << 8 | h[7]);
        q.channels = h[8];
        q.rate = (double)((uint32_t)h[9] << 16 | (uint32_t)h[10] << 8 | h[11]);
        if (q.channels < 1 || q.channels > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "QOA channel count %d", q.channels);
        q.raised = wo[s].raised != 0; q.big = wo[s].big != 0; q.ncalls = wo[s].ncalls;
        q.njobs = wo[s].njobs; q.L = wo[s].L;
        q.stride = mode == 0 ? round_up(std::max<uint64_t>(q.L, 1), 8) : 0;
        q.rows_total = mode == 0 ? q.stride * (uint64_t)q.channels : wo[s].rows_total;
        q.job_first = jat; q.call_first = cat; q.row_base = rat;
        jat += q.njobs; cat += q.ncalls; rat += q.rows_total;
    }
    return AUKIT_OK;
}

// pass 2: the decode jobs into `djobs` (device), the call records to the host (stream mode: want_calls)
static int qoa_walk_fill(aukit_ctx *ctx, const aukit_batch *in, int mode, const std::vector<QoaStreamInfo> &S, QoaJob *djobs, uint64_t ncalls, std::vector<QoaCallRec> *calls, DevBuf *wb = nullptr, DevBuf *cb = nullptr) {
    const uint32_t n = in->n;
    std::vector<QoaFillIn> fin(n);
    for (uint32_t s = 0; s < n; s++) fin[s] = QoaFillIn{S[s].job_first, S[s].call_first, S[s].row_base, S[s].stride};
    QoaFillIn *dfin = reinterpret_cast<QoaFillIn *>(reinterpret_cast<QoaWalkOut *>((wb ? *wb : ctx->tmp_buf3).p) + n);
    { int hrc = h2d_table(ctx, dfin, fin.data(), (size_t)n * sizeof(QoaFillIn)); if (hrc) return hrc; }
    DevBuf &CB = cb ? *cb : ctx->misc_buf;
    int rc = CB.ensure((size_t)std::max<uint64_t>(ncalls, 1) * sizeof(QoaCallRec) + 64);
    if (rc) return rc;
    QoaCallRec *dcalls = reinterpret_cast<QoaCallRec *>(CB.p);
    hipLaunchKernelGGL((k_qoa_walk<true>), dim3((n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off), n, mode,
                       static_cast<QoaWalkOut *>(nullptr), dfin, djobs, dcalls);
    AUKIT_HIP_CHECK(hipGetLastError());
    if (calls) {
        calls->resize(ncalls);
        if (ncalls) {
            void *st = ctx_host_stage(ctx, ncalls * sizeof(QoaCallRec));
            AUKIT_HIP_CHECK(hipMemcpyAsync(st ? st : calls->data(), dcalls, ncalls * sizeof(QoaCallRec), hipMemcpyDeviceToHost, ctx->stream));
            AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (st) memcpy(calls->data(), st, ncalls * sizeof(QoaCallRec));
        }
    }
    return AUKIT_OK;
}

template <bool S8>
static int qoa_decode_launch(aukit_ctx *ctx, const aukit_batch *in, const QoaJob *djobs, uint64_t njobs, void *rows, uint64_t row_bytes) {
    if (!njobs) return AUKIT_OK;
    int rc = ctx_begin_kernel(ctx);
    if (rc) return rc;
    hipLaunchKernelGGL((k_qoa_wave<S8>), dim3((unsigned)((njobs + 63) / 64)), dim3(64), 0, ctx->stream, in->data(), djobs, (unsigned long long)njobs, rows);
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, "k_qoa_wave", in->total() + row_bytes);
}

// aukit.qoa(data)  aukit.lua:1706-1777
int decode_qoa_audio(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, double new_rate, int interp, bool do_resample, int dtype, aukit_audio **out) {
    if (getenv("AUKIT_QOA_HOST")) return decode_qoa_audio_host(ctx, in, d, new_rate, interp, do_resample, dtype, out);
    if (in->n == 0) return fail(AUKIT_E_ARG, "empty batch");
    QoaLaps laps;
    std::vector<QoaStreamInfo> S;
    int rc = qoa_walk_count(ctx, in, 0, S);
    if (rc) return rc;
    laps.lap("walk 1");
    const int C = S[0].channels;
    const double rate = S[0].rate;
    std::vector<uint64_t> row_off, row_len;
    uint64_t tot = 0, njobs = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        if (S[s].raised) return fail(AUKIT_E_LUA, "data string too short");
        if (S[s].channels != C || S[s].rate != rate) return fail(AUKIT_E_ARG, "all QOA streams of a batch must share channel count and sample rate");
        if (S[s].big) return decode_qoa_audio_host(ctx, in, d, new_rate, interp, do_resample, dtype, out);   // frames of more than 8192 samples: weights could leave 24 bits
        for (int c = 0; c < C; c++) { row_off.push_back(S[s].row_base + (uint64_t)c * S[s].stride); row_len.push_back(S[s].L); }
        tot += S[s].rows_total; njobs += S[s].njobs;
    }
    if ((rc = ctx->tmp_buf.ensure((size_t)tot * 2 + 64))) return rc;
    if ((rc = ctx->tmp_buf2.ensure((size_t)std::max<uint64_t>(njobs, 1) * sizeof(QoaJob) + 64))) return rc;
    QoaJob *djobs = reinterpret_cast<QoaJob *>(ctx->tmp_buf2.p);
    if ((rc = qoa_walk_fill(ctx, in, 0, S, djobs, in->n, nullptr))) return rc;
    if ((rc = qoa_decode_launch<false>(ctx, in, djobs, njobs, ctx->tmp_buf.p, tot * 2))) return rc;
    laps.lap("walk 2 + decode");
    struct AtExit { QoaLaps &l; ~AtExit() { l.lap("rows -> audio"); } } at_exit{laps};
    return audio_from_int_rows(ctx, SRC_I16, ctx->tmp_buf.p, row_off, row_len, in->n, C, rate, new_rate, interp, do_resample, dtype, 32767, 32768, out);
}

// aukit.stream.qoa(data, mono)  aukit.lua:3202-3337
int stream_qoa(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks_out) {
    if (getenv("AUKIT_QOA_HOST")) return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out);
    if (interp == AUKIT_INTERP_SINC) return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out);   // aukit.defaultInterpolation = "sinc" (:3252): the reference-order kernels
    if (interp < 0 || interp > 2) return fail(AUKIT_E_ARG, "stream.qoa: bad interpolation");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "stream.qoa output must be AUKIT_F64 or AUKIT_F32");
    if (in->n == 0) return fail(AUKIT_E_ARG, "empty batch");
    QoaLaps laps;
    std::vector<QoaStreamInfo> S;
    // Round 6, late: the two walks (they read the input batch and nothing else) run on the look-ahead stream, into words of their own (ctx->qoa_set, alternating):
    // their host waits were waits for everything the call BEFORE had left on ctx->stream — a host that issues call after call now counts, plans and
    // builds the tail's job table while that call's decoder and tail still run.  entry_ev orders a set's next writer behind its last readers (common.h).
    // AUKIT_QOA_WALK_MAIN=1: everything on ctx->stream, as before.
    struct StreamSwap { aukit_ctx *c; hipStream_t saved; bool on; void back() { if (on) { c->stream = saved; on = false; } } ~StreamSwap() { back(); } } sw{ctx, ctx->stream, false};
    DevBuf *qs = nullptr;   // the call's set: [0] walk words, [1] call records, [2] decode jobs
    hipStream_t pre = nullptr;
    if (!getenv("AUKIT_QOA_WALK_MAIN")) {
        int prc = ctx_pre_stream(ctx, &pre);
        if (prc) return prc;
        if (in->ready) AUKIT_HIP_CHECK(hipStreamWaitEvent(pre, in->ready, 0));
        const uint64_t k = ctx->flac_calls++;
        AUKIT_HIP_CHECK(hipEventRecord(ctx->entry_ev[k & 1], ctx->stream));
        if (k >= 1) AUKIT_HIP_CHECK(hipStreamWaitEvent(pre, ctx->entry_ev[(k - 1) & 1], 0));
        ctx->qoa_par ^= 1;
        qs = ctx->qoa_set[ctx->qoa_par];
        ctx->stream = pre; sw.on = true;
    }
    int rc = qoa_walk_count(ctx, in, 1, S, qs ? &qs[0] : nullptr);
    if (rc) return rc;
    laps.lap("walk 1");
    const int C = S[0].channels;
    const double rate = S[0].rate;
    if (!(rate > 0)) { sw.back(); return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out); }
    const double ratio = 48000 / rate;
    uint64_t tot = 0, njobs = 0, ncalls = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        if (S[s].channels != C || S[s].rate != rate) return fail(AUKIT_E_ARG, "all QOA streams of a batch must share channel count and sample rate");
        if (S[s].big) { sw.back(); return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out); }
        tot += S[s].rows_total; njobs += S[s].njobs; ncalls += S[s].ncalls;
    }
    std::vector<QoaCallRec> calls;
    if ((rc = ctx->tmp_buf.ensure((size_t)tot + 64))) return rc;
    DevBuf &JB = qs ? qs[2] : ctx->tmp_buf2;
    if ((rc = JB.ensure((size_t)std::max<uint64_t>(njobs, 1) * sizeof(QoaJob) + 64))) return rc;
    QoaJob *djobs = reinterpret_cast<QoaJob *>(JB.p);
    if ((rc = qoa_walk_fill(ctx, in, 1, S, djobs, ncalls, &calls, qs ? &qs[0] : nullptr, qs ? &qs[1] : nullptr))) return rc;
    if (sw.on) {   // ctx->stream behind the walks (the decoder reads their jobs)
        sw.back();
        AUKIT_HIP_CHECK(hipEventRecord(ctx->pre_ev, pre));
        AUKIT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->pre_ev, 0));
    }
    laps.lap("walk 2");
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0); ck->status.assign(in->n, 0); ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0);
    std::vector<uint64_t> nouts(ncalls, 0);
    for (uint32_t s = 0; s < in->n; s++) {
        ck->length_seconds[s] = S[s].file_samples / rate;
        for (unsigned k = 0; k < S[s].ncalls; k++) {
            const QoaCallRec &cr = calls[S[s].call_first + k];
            const double newlen = (double)cr.n * ratio;                                            // :3312
            const uint64_t nout = newlen >= 1 ? (uint64_t)std::floor(newlen) : 0;
            if (nout > 0x7FFFFFF0ull) { delete ck; return fail(AUKIT_E_UNSUPPORTED, "stream too long"); }
            nouts[S[s].call_first + k] = nout;
            lens[s] += nout;
        }
        if (S[s].raised) ck->status[s] = AUKIT_E_LUA;
        ck->nchunks[s] = S[s].ncalls;
        ck->max_chunks = std::max(ck->max_chunks, S[s].ncalls);
    }
    const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
    ck->lens.assign((size_t)ck->n * mc, 0);
    ck->pos.assign((size_t)ck->n * mc, 0);
    for (uint32_t s = 0; s < in->n; s++) {
        double file_pos = (double)ctx->sb_samples;   // (the rest of a stream behind a bounded reader-function handle: what was dropped counts)
        for (unsigned k = 0; k < S[s].ncalls; k++) {
            ck->lens[(size_t)s * mc + k] = (uint32_t)nouts[S[s].call_first + k];
            ck->pos[(size_t)s * mc + k] = file_pos / rate;                                         // :3332
            file_pos += (double)calls[S[s].call_first + k].sample_pos;
        }
    }
    const bool mix = mono && C > 1;
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, mix ? 1 : C, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    laps.lap("plan");
    if (ncalls) {
        if ((rc = qoa_decode_launch<true>(ctx, in, djobs, njobs, ctx->tmp_buf.p, tot))) { delete ck; return rc; }
        std::vector<TailJob> jobs;
        jobs.reserve((size_t)ncalls * (mix ? 1 : C));
        uint64_t total_out = 0;
        for (uint32_t s = 0; s < in->n; s++) {
            uint64_t outpos = 0;
            for (unsigned k = 0; k < S[s].ncalls; k++) {
                const QoaCallRec &cl = calls[S[s].call_first + k];
                const uint64_t nout = nouts[S[s].call_first + k];
                for (int c = 0; c < (mix ? 1 : C); c++) {
                    TailJob j;
                    memset(&j, 0, sizeof j);
                    j.src_off = cl.row0 + (uint64_t)c * cl.stride;
                    j.last_off = ~0ull; j.m1_off = ~0ull;
                    if (k > 0) {   // chunk[i] = {[-1] = last[i][1], [0] = last[i][2]}  :3255
                        const QoaCallRec &pc = calls[S[s].call_first + k - 1];
                        j.last_off = pc.row0 + (uint64_t)c * pc.stride + pc.n - 1;
                        j.m1_off = j.last_off - 1;
                        j.last_cstride = pc.stride;
                    }
                    j.src_cstride = cl.stride;
                    j.out_off = a->row_off[s] + (mix ? 0 : (uint64_t)c * a->row_stride[s]) + outpos;
                    j.n = (int)cl.n; j.nout = (int)nout;
                    jobs.push_back(j);
                    total_out += nout;
                }
                outpos += nout;
            }
        }
        int trc = AUKIT_OK;
        const double lp_alpha = 1 - std::exp(-(rate / 96000) * 2 * M_PI);   // :3250
        if (dtype == AUKIT_F32 && (!mix || C == 2) &&
            rs_onepole_jobs_try(ctx, ctx->tmp_buf.p, jobs, mix ? C : 1, rate, interp, lp_alpha, reinterpret_cast<float *>(a->dev), tot + total_out * dtype_size(dtype), "k_rs_onepole<qoa>", &trc)) {
            // (the tile chain of flac_tail.hip: the state carried, not warmed up)
        } else
        if (!iir_tail_try(ctx, TAIL_QOA, TAIL_ROWS_I8, ctx->tmp_buf.p, 1.0, jobs, mix ? C : 1, rate, interp, dtype, a->dev, tot + total_out * dtype_size(dtype), "k_iir_tail<qoa>", &trc)) {
            delete ck;
            return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out);   // very low sample rates: the filter's memory outlasts a tile's warm-up
        }
        if (trc) { delete ck; return trc; }
        laps.lap("decode + tail");
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

}  // namespace aukit
