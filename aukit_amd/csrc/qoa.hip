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
struct QoaFrameRec { unsigned long long pos; unsigned samples; unsigned call; };   // pos: byte offset of the frame header within the stream
struct QoaWalkOut { unsigned char head[12]; unsigned nframes; unsigned raised; unsigned raised_call; unsigned ncalls; unsigned pad; };
static_assert(sizeof(QoaWalkOut) == 32 && sizeof(QoaFrameRec) == 16, "walk records");

// One lane per stream.  mode 0: aukit.qoa's loop (:1727-1775); mode 1: stream.qoa's iterator calls (:3256-3308) one after the other.
// FILL false: counts; true: writes the records at recs + first[s].  The two loops below are the reference's, statement for statement
// (cf. qoa_scan in codecs2.hip, the host version of round 2).
template <bool FILL>
__global__ __launch_bounds__(64) void k_qoa_walk(const unsigned char *src, const unsigned long long *off, unsigned n, int mode, QoaWalkOut *wo,
                                                const unsigned long long *first, QoaFrameRec *recs) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const unsigned char *h = src + off[s];
    const unsigned long long nb = off[s + 1] - off[s];
    QoaWalkOut o;
    for (int i = 0; i < 12; i++) o.head[i] = (unsigned long long)i < nb ? h[i] : 0;
    o.nframes = 0; o.raised = 0; o.raised_call = 0; o.ncalls = 0; o.pad = 0;
    QoaFrameRec *dst = FILL ? recs + first[s] : nullptr;
    if (nb >= 12 && h[0] == 'q' && h[1] == 'o' && h[2] == 'a' && h[3] == 'f') {
        const double file_samples = (double)((unsigned)h[4] << 24 | (unsigned)h[5] << 16 | (unsigned)h[6] << 8 | h[7]);
        const int fc = h[8];
        const unsigned fr = (unsigned)h[9] << 16 | (unsigned)h[10] << 8 | h[11];
        unsigned long long pos = 8;
        auto header_ok = [&](const unsigned char *f, int &channels, int &samples, int &frame_size) {
            channels = f[0];
            const unsigned rate = (unsigned)f[1] << 16 | (unsigned)f[2] << 8 | f[3];
            samples = f[4] << 8 | f[5]; frame_size = f[6] << 8 | f[7];
            const int data_size = frame_size - 8 - 16 * channels;
            const int num_slices = data_size >= 0 ? data_size / 8 : -((-data_size + 7) / 8);   // math_floor(data_size / 8)
            return !(channels != fc || rate != fr || samples * channels > num_slices * 20);
        };
        if (fc >= 1 && fc <= AUKIT_MAX_CHANNELS) {
            if (mode == 0) {
                double sample_pos = 0;
                for (;;) {
                    if (!(pos + 1 + 16ull * fc + 8 <= nb && sample_pos < file_samples)) break;
                    int channels, samples, frame_size;
                    if (!header_ok(h + pos, channels, samples, frame_size)) break;
                    if ((double)frame_size > (double)nb - (double)(pos + 8)) break;   // frame_size > #data - pos + 1, pos after the header (Q18)
                    const unsigned long long need = 8 + 16ull * channels + 8ull * (unsigned long long)((samples + 19) / 20) * channels;
                    if (pos + need > nb) { o.raised = 1; break; }                      // "data string too short"
                    if (FILL) { dst[o.nframes].pos = pos; dst[o.nframes].samples = (unsigned)samples; dst[o.nframes].call = 0; }
                    o.nframes++;
                    pos += need;
                    sample_pos += samples;
                }
                o.ncalls = 1;
            } else {
                for (unsigned call = 0;; call++) {   // one iterator call
                    double sample_pos = 0;
                    unsigned long long nmax = 0, sp = 0;
                    bool raised = false;
                    while (sample_pos < (double)fr) {
                        if (pos >= nb) break;                                          // read(8) → nil
                        if (pos + 8 > nb) { raised = true; break; }
                        int channels, samples, frame_size;
                        const unsigned long long fpos = pos;
                        const bool ok = header_ok(h + pos, channels, samples, frame_size);
                        pos += 8;
                        if (!ok) break;                                                // :3270-3277 (the header is consumed)
                        const unsigned long long need = 16ull * channels + 8ull * (unsigned long long)((samples + 19) / 20) * channels;
                        if (pos + need > nb) { raised = true; break; }                 // assert(read(8), "Invalid QOA data") / short unpack
                        if (FILL) { dst[o.nframes].pos = fpos; dst[o.nframes].samples = (unsigned)samples; dst[o.nframes].call = call; }
                        o.nframes++;
                        pos += need;
                        const unsigned long long top = sp + (unsigned long long)((samples + 19) / 20) * 20;
                        nmax = top > nmax ? top : nmax;
                        sp += (unsigned long long)samples;
                        sample_pos += samples;
                    }
                    if (raised) { o.raised = 1; o.raised_call = call; break; }
                    if (nmax == 0) break;                                              // #chunk[1] == 0 → nil
                    o.ncalls = call + 1;
                }
            }
        }
    }
    if (!FILL) wo[s] = o;
}

// ================================================================= the wave decoder
static AUKIT_DEV int qoa_mul24(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
static AUKIT_DEV int qoa_mad24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
static AUKIT_DEV int qoa_med3(int a, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi)); return r; }

// S8: stream.qoa's math_floor(reconstructed / 256) as int8 rows (:3299); else the reconstructed int16 (aukit.qoa, :1765 divides later)
template <bool S8>
__global__ __launch_bounds__(64) void k_qoa_wave(const unsigned char *src, const QoaJob *jobs, unsigned long long njobs, void *out) {
    constexpr int K = 4;                          // slices per job and round
    constexpr int OB = S8 ? 1 : 2;                // bytes per stored sample
    constexpr int OSTR = 80 * OB + 8;             // bytes between two jobs' output areas
    constexpr int UPJ = 80 * OB / 8;              // 8-byte units per job and round
    __shared__ unsigned long long sl_in[64 * (K + 1)];
    __shared__ unsigned long long j_src[64], j_out[64];
    __shared__ unsigned j_step[64], j_ns[64];
    __shared__ int j_emit[64];
    __shared__ int deq[128];
    __shared__ __attribute__((aligned(16))) unsigned char obuf[64 * OSTR];
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
#pragma unroll
        for (int sj = 0; sj < K; sj++) {
            const unsigned long long raw = sl_in[lane * (K + 1) + sj];
            unsigned hi = __builtin_bswap32((unsigned)raw), lo = __builtin_bswap32((unsigned)(raw >> 32));   // (">I4I4"):unpack
            const int *dq = deq + (hi >> 28) * 8;   // scalefactor = bit32_extract(sliceH, 28, 4)
            unsigned pk[S8 ? 5 : 10];
#pragma unroll
            for (int k = 0; k < 20; k++) {
                const int sum = qoa_mad24(w3, h3, qoa_mad24(w2, h2, qoa_mad24(w1, h1, qoa_mul24(w0, h0))));   // the low 32 bits of the Lua's double sum
                const int predicted = sum >> 13;                                                              // signed_rshift(..., 13)  :1686-1689
                const int d = dq[(hi >> 25) & 7];
                const int rec = qoa_med3(predicted + d, lo16, hi16);                                         // :1763
                if constexpr (S8) {
                    const unsigned v8 = (unsigned)(rec >> 8) & 0xFFu;                                        // math_floor(reconstructed / 256)  :3299
                    if ((k & 3) == 0) pk[k >> 2] = v8; else pk[k >> 2] |= v8 << (8 * (k & 3));
                } else {
                    const unsigned v16 = (unsigned)rec & 0xFFFFu;
                    if (k & 1) pk[k >> 1] |= v16 << 16; else pk[k >> 1] = v16;
                }
                hi = (hi << 3) | (lo >> 29);
                lo <<= 3;
                const int delta = d >> 4;                                                                     // signed_rshift(residual, 4)
                w0 += h0 < 0 ? -delta : delta; w1 += h1 < 0 ? -delta : delta; w2 += h2 < 0 ? -delta : delta; w3 += h3 < 0 ? -delta : delta;   // :1694-1699
                h0 = h1; h1 = h2; h2 = h3; h3 = rec;
            }
            unsigned *ow = reinterpret_cast<unsigned *>(obuf + lane * OSTR + sj * 20 * OB);
#pragma unroll
            for (int q = 0; q < (S8 ? 5 : 10); q++) ow[q] = pk[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the round's 80 samples of every job, 8 bytes per lane along each job's row
#pragma unroll 2
        for (int i = 0; i < UPJ; i++) {
            const unsigned u = (unsigned)i * 64 + lane, jb = u / UPJ, v = u % UPJ;
            constexpr int SPU = 8 / OB;   // samples per unit
            const int sb = (int)(r * 80) + (int)v * SPU, em = j_emit[jb];
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
struct QoaStreamFrames { int channels; double rate, file_samples; bool raised; unsigned raised_call, ncalls; size_t first, count; };

// the device walk of every stream; validates the file headers like the reference (errors with its strings)
static int qoa_walk(aukit_ctx *ctx, const aukit_batch *in, int mode, std::vector<QoaStreamFrames> &S, std::vector<QoaFrameRec> &recs) {
    const uint32_t n = in->n;
    S.assign(n, QoaStreamFrames{});
    recs.clear();
    if (!n) return AUKIT_OK;
    int rc = ctx->tmp_buf3.ensure((size_t)n * (sizeof(QoaWalkOut) + 8) + 64);
    if (rc) return rc;
    QoaWalkOut *dwo = reinterpret_cast<QoaWalkOut *>(ctx->tmp_buf3.p);
    unsigned long long *dfirst = reinterpret_cast<unsigned long long *>(dwo + n);
    const unsigned long long *doff = reinterpret_cast<const unsigned long long *>(in->d_off);
    hipLaunchKernelGGL((k_qoa_walk<false>), dim3((n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), doff, n, mode, dwo, dfirst, static_cast<QoaFrameRec *>(nullptr));
    AUKIT_HIP_CHECK(hipGetLastError());
    // (both read-backs go through the context's pinned staging buffer: a pageable destination of a megabyte makes the runtime pin it on the
    // fly — 26 ms per call, measured, against 0.4 ms)
    std::vector<QoaWalkOut> wo(n);
    {
        void *st = ctx_host_stage(ctx, (size_t)n * sizeof(QoaWalkOut));
        AUKIT_HIP_CHECK(hipMemcpyAsync(st ? st : wo.data(), dwo, (size_t)n * sizeof(QoaWalkOut), hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (st) memcpy(wo.data(), st, (size_t)n * sizeof(QoaWalkOut));
    }
    std::vector<unsigned long long> first(n);
    size_t tot = 0;
    for (uint32_t s = 0; s < n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        const unsigned char *h = wo[s].head;
        // aukit.qoa: (">c4I4"):unpack / (">BI3"):unpack on a short string raise; stream.qoa: assert(read(8), ...) / assert(peek(4), ...)
        if (nb < 8) return fail(AUKIT_E_LUA, mode == 0 ? "data string too short" : "Not a QOA file");
        if (memcmp(h, "qoaf", 4) != 0) return fail(AUKIT_E_ARG, "Not a QOA file");
        if (nb < 12) return fail(AUKIT_E_LUA, nb == 8 && mode == 1 ? "Not a QOA file" : "data string too short");
        S[s].file_samples = (double)((uint32_t)h[4] << 24 | (uint32_t)h[5] << 16 | (uint32_t)h[6] << 8 | h[7]);
        S[s].channels = h[8];
        S[s].rate = (double)((uint32_t)h[9] << 16 | (uint32_t)h[10] << 8 | h[11]);
        if (S[s].channels < 1 || S[s].channels > AUKIT_MAX_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "QOA channel count %d", S[s].channels);
        S[s].raised = wo[s].raised != 0; S[s].raised_call = wo[s].raised_call; S[s].ncalls = wo[s].ncalls;
        S[s].first = tot; S[s].count = wo[s].nframes;
        first[s] = tot;
        tot += wo[s].nframes;
    }
    recs.resize(tot);
    if (!tot) return AUKIT_OK;
    if ((rc = ctx->misc_buf.ensure(tot * sizeof(QoaFrameRec) + 64))) return rc;
    { int hrc = h2d_table(ctx, dfirst, first.data(), (size_t)n * 8); if (hrc) return hrc; }
    QoaFrameRec *drecs = reinterpret_cast<QoaFrameRec *>(ctx->misc_buf.p);
    hipLaunchKernelGGL((k_qoa_walk<true>), dim3((n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), doff, n, mode, dwo, dfirst, drecs);
    AUKIT_HIP_CHECK(hipGetLastError());
    {
        void *st = ctx_host_stage(ctx, tot * sizeof(QoaFrameRec));
        AUKIT_HIP_CHECK(hipMemcpyAsync(st ? st : recs.data(), drecs, tot * sizeof(QoaFrameRec), hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (st) memcpy(recs.data(), st, tot * sizeof(QoaFrameRec));
    }
    return AUKIT_OK;
}

template <bool S8>
static int qoa_decode_launch(aukit_ctx *ctx, const aukit_batch *in, const std::vector<QoaJob> &jobs, void *rows, uint64_t row_bytes) {
    if (jobs.empty()) return AUKIT_OK;
    int rc = upload_table(ctx, ctx->tmp_buf2, jobs.data(), jobs.size() * sizeof(QoaJob));
    if (rc) return rc;
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    hipLaunchKernelGGL((k_qoa_wave<S8>), dim3((unsigned)((jobs.size() + 63) / 64)), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const QoaJob *>(ctx->tmp_buf2.p),
                       (unsigned long long)jobs.size(), rows);
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, "k_qoa_wave", in->total() + row_bytes);
}

// aukit.qoa(data)  aukit.lua:1706-1777
int decode_qoa_audio(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, double new_rate, int interp, bool do_resample, int dtype, aukit_audio **out) {
    if (getenv("AUKIT_QOA_HOST")) return decode_qoa_audio_host(ctx, in, d, new_rate, interp, do_resample, dtype, out);
    if (in->n == 0) return fail(AUKIT_E_ARG, "empty batch");
    std::vector<QoaStreamFrames> S;
    std::vector<QoaFrameRec> recs;
    QoaLaps laps;
    int rc = qoa_walk(ctx, in, 0, S, recs);
    if (rc) return rc;
    laps.lap("walk");
    const int C = S[0].channels;
    const double rate = S[0].rate;
    std::vector<QoaJob> jobs;
    std::vector<uint64_t> row_off, row_len;
    uint64_t tot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        if (S[s].raised) return fail(AUKIT_E_LUA, "data string too short");
        if (S[s].channels != C || S[s].rate != rate) return fail(AUKIT_E_ARG, "all QOA streams of a batch must share channel count and sample rate");
        const QoaFrameRec *fr = recs.data() + S[s].first;
        uint64_t L = 0, sp = 0;
        for (size_t k = 0; k < S[s].count; k++) {
            if (fr[k].samples > 8192) return decode_qoa_audio_host(ctx, in, d, new_rate, interp, do_resample, dtype, out);   // weights could leave 24 bits
            L = std::max<uint64_t>(L, sp + (uint64_t)((fr[k].samples + 19) / 20) * 20);
            sp += fr[k].samples;
        }
        const uint64_t stride = round_up(std::max<uint64_t>(L, 1), 8);
        sp = 0;
        for (size_t k = 0; k < S[s].count; k++) {
            const bool lastf = k + 1 == S[s].count;
            for (int c = 0; c < C; c++) {
                QoaJob j;
                j.frame_off = in->off[s] + fr[k].pos; j.out_off = tot + (uint64_t)c * stride + sp;
                j.c = c; j.channels = C; j.samples = (int)fr[k].samples;
                j.emit = lastf ? (int)((fr[k].samples + 19) / 20) * 20 : (int)fr[k].samples;  // Q15: the ≤19-sample tail survives only after the last frame
                jobs.push_back(j);
            }
            sp += fr[k].samples;
        }
        for (int c = 0; c < C; c++) { row_off.push_back(tot + (uint64_t)c * stride); row_len.push_back(L); }
        tot += stride * C;
    }
    laps.lap("jobs");
    if ((rc = ctx->tmp_buf.ensure((size_t)tot * 2 + 64))) return rc;
    if ((rc = qoa_decode_launch<false>(ctx, in, jobs, ctx->tmp_buf.p, tot * 2))) return rc;
    laps.lap("decode launch");
    struct AtExit { QoaLaps &l; ~AtExit() { l.lap("rows -> audio"); } } at_exit{laps};
    return audio_from_int_rows(ctx, SRC_I16, ctx->tmp_buf.p, row_off, row_len, in->n, C, rate, new_rate, interp, do_resample, dtype, 32767, 32768, out);
}

// aukit.stream.qoa(data, mono)  aukit.lua:3202-3337
int stream_qoa(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks_out) {
    if (getenv("AUKIT_QOA_HOST")) return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out);
    if (interp < 0 || interp > 2) return fail(interp == AUKIT_INTERP_SINC ? AUKIT_E_UNSUPPORTED : AUKIT_E_ARG, "stream.qoa: interpolation must be none, linear or cubic");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "stream.qoa output must be AUKIT_F64 or AUKIT_F32");
    if (in->n == 0) return fail(AUKIT_E_ARG, "empty batch");
    std::vector<QoaStreamFrames> S;
    std::vector<QoaFrameRec> recs;
    int rc = qoa_walk(ctx, in, 1, S, recs);
    if (rc) return rc;
    const int C = S[0].channels;
    const double rate = S[0].rate;
    if (!(rate > 0)) return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out);
    const double ratio = 48000 / rate;
    for (uint32_t s = 0; s < in->n; s++) {
        if (S[s].channels != C || S[s].rate != rate) return fail(AUKIT_E_ARG, "all QOA streams of a batch must share channel count and sample rate");
        for (size_t k = 0; k < S[s].count; k++) if (recs[S[s].first + k].samples > 8192) return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out);
    }
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0); ck->status.assign(in->n, 0); ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0);
    std::vector<QoaJob> djobs;
    struct Call { uint32_t stream; uint64_t row0, stride, n, nout; };  // row0: element offset of channel 0's table index 1; channel c at row0 + c * stride
    std::vector<Call> calls;
    std::vector<double> cpos;
    uint64_t tot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        ck->length_seconds[s] = S[s].file_samples / rate;
        const QoaFrameRec *fr = recs.data() + S[s].first;
        double file_pos = 0;
        size_t k = 0;
        uint32_t nch = 0;
        while (k < S[s].count) {
            const unsigned call = fr[k].call;
            size_t e = k;
            while (e < S[s].count && fr[e].call == call) e++;
            if (S[s].raised && call == S[s].raised_call) break;   // the call that raised delivers nothing
            if (call >= S[s].ncalls) break;                          // (a call whose table stayed empty ends the stream)
            uint64_t n = 0, sp = 0;
            double sample_pos = 0;
            for (size_t q = k; q < e; q++) { n = std::max<uint64_t>(n, sp + (uint64_t)((fr[q].samples + 19) / 20) * 20); sp += fr[q].samples; sample_pos += fr[q].samples; }
            const uint64_t stride = round_up(n + 2, 16);
            sp = 0;
            for (size_t q = k; q < e; q++) {
                const bool lastf = q + 1 == e;
                for (int c = 0; c < C; c++) {
                    QoaJob j;
                    j.frame_off = in->off[s] + fr[q].pos; j.out_off = tot + (uint64_t)c * stride + sp;
                    j.c = c; j.channels = C; j.samples = (int)fr[q].samples;
                    j.emit = lastf ? (int)((fr[q].samples + 19) / 20) * 20 : (int)fr[q].samples;
                    djobs.push_back(j);
                }
                sp += fr[q].samples;
            }
            const double newlen = (double)n * ratio;                                               // :3312
            const uint64_t nout = newlen >= 1 ? (uint64_t)std::floor(newlen) : 0;
            if (nout > 0x7FFFFFF0ull) { delete ck; return fail(AUKIT_E_UNSUPPORTED, "stream too long"); }
            calls.push_back(Call{s, tot, stride, n, nout});
            cpos.push_back(file_pos / rate);                                                       // :3332
            file_pos += sample_pos;
            lens[s] += nout;
            tot += stride * C;
            nch++;
            k = e;
        }
        if (S[s].raised) ck->status[s] = AUKIT_E_LUA;
        ck->nchunks[s] = nch;
        ck->max_chunks = std::max(ck->max_chunks, nch);
    }
    const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
    ck->lens.assign((size_t)ck->n * mc, 0);
    ck->pos.assign((size_t)ck->n * mc, 0);
    {
        std::vector<uint32_t> at(in->n, 0);
        for (size_t k = 0; k < calls.size(); k++) {
            const uint32_t s = calls[k].stream;
            ck->lens[(size_t)s * mc + at[s]] = (uint32_t)calls[k].nout;
            ck->pos[(size_t)s * mc + at[s]] = cpos[k];
            at[s]++;
        }
    }
    const bool mix = mono && C > 1;
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, mix ? 1 : C, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    if (!calls.empty()) {
        if ((rc = ctx->tmp_buf.ensure((size_t)tot + 64))) { delete ck; return rc; }
        if ((rc = qoa_decode_launch<true>(ctx, in, djobs, ctx->tmp_buf.p, tot))) { delete ck; return rc; }
        std::vector<TailJob> jobs;
        std::vector<uint64_t> outpos(in->n, 0);
        std::vector<long long> prev(in->n, -1);
        uint64_t nouts = 0;
        for (size_t k = 0; k < calls.size(); k++) {
            const Call &cl = calls[k];
            for (int c = 0; c < (mix ? 1 : C); c++) {
                TailJob j;
                memset(&j, 0, sizeof j);
                j.src_off = cl.row0 + (uint64_t)c * cl.stride;
                j.last_off = ~0ull; j.m1_off = ~0ull;
                if (prev[cl.stream] >= 0) {   // chunk[i] = {[-1] = last[i][1], [0] = last[i][2]}  :3255
                    const Call &pc = calls[(size_t)prev[cl.stream]];
                    j.last_off = pc.row0 + (uint64_t)c * pc.stride + pc.n - 1;
                    j.m1_off = j.last_off - 1;
                    j.last_cstride = (unsigned)pc.stride;
                }
                j.src_cstride = (unsigned)cl.stride;
                j.out_off = a->row_off[cl.stream] + (mix ? 0 : (uint64_t)c * a->row_stride[cl.stream]) + outpos[cl.stream];
                j.n = (int)cl.n; j.nout = (int)cl.nout;
                jobs.push_back(j);
                nouts += cl.nout;
            }
            outpos[cl.stream] += cl.nout;
            prev[cl.stream] = (long long)k;
        }
        int trc = AUKIT_OK;
        if (!iir_tail_try(ctx, TAIL_QOA, TAIL_ROWS_I8, ctx->tmp_buf.p, 1.0, jobs, mix ? C : 1, rate, interp, dtype, a->dev, tot + nouts * dtype_size(dtype), "k_iir_tail<qoa>", &trc)) {
            delete ck;
            return stream_qoa_host(ctx, in, d, interp, mono, dtype, out, chunks_out);   // very low sample rates: the filter's memory outlasts a tile's warm-up
        }
        if (trc) { delete ck; return trc; }
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

}  // namespace aukit
