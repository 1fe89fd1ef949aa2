// qoa_stream.hip — aukit.stream.qoa (aukit.lua:3202-3337) on gfx950.
//
// Frames carry their own LMS state, so (frame, channel) pairs decode independently (k_qoa in codecs2.hip).  The stream
// iterator then resamples each ≈1 s table with a recursive one-pole low-pass whose state runs across the whole call
// (:3316-3325, Q15): one lane per (stream, call, channel) walks its outputs in order, taps come straight from the
// int16 rows (L2-resident), history of the previous call through two extra slots.  Not on a BASELINE config.
#include <algorithm>
#include "resample.h"
#include "resample_dev.h"

namespace aukit {

struct QoaJob { unsigned long long frame_off, out_off; int c, channels, samples, emit; };
__global__ void k_qoa(const unsigned char *src, const QoaJob *jobs, unsigned long long njobs, short *out, int shift8);

struct QsJob {
    unsigned long long src_off;   // element offset (int16) of table index 1
    unsigned long long last_off;  // element offset of the previous call's table index #chunk (its last sample); ~0 → {0, 0}
    unsigned long long out_off;
    int n, nout;
};

template <int INTERP, typename OUT_T>
__global__ __launch_bounds__(64) void k_qoa_stream(const QsJob *jobs, unsigned long long njobs, const short *rows, OUT_T *out, double ratio, double rcp, int exact,
                                                  double lp_alpha, int sinc_w) {
    const unsigned long long j = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= njobs) return;
    const QsJob job = jobs[j];
    const short *src = rows + job.src_off;
    double m1 = 0, z0 = 0;  // chunk[i] = {[-1] = last[i][1], [0] = last[i][2]}  :3255
    if (job.last_off != ~0ull) { z0 = (double)rows[job.last_off]; m1 = (double)rows[job.last_off - 1]; }
    double ls = z0;         // :3316
    const int n = job.n;
    auto tap = [&](int k) -> double { return k >= 1 ? (double)src[k - 1] : (k == 0 ? z0 : m1); };
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
            else if constexpr (INTERP == AUKIT_INTERP_SINC) s = sinc_at(tap, k, -1, n, fx, sinc_w);
            else { const double p1 = tap(k), p0 = tap(k - 1), p2 = (k + 1 <= n) ? tap(k + 1) : p1, p3 = (k + 2 <= n) ? tap(k + 2) : p2; s = cubic_exact(p0, p1, p2, p3, fx); }
            s = lua_clamp(s, -128, 127);  // :3323
        }
        s = ls + lp_alpha * (s - ls);     // :3324
        ls = s;
        out[job.out_off + i] = (OUT_T)s;
    }
}

// The same in two passes, for small batches (one file is the usual batch of austream): with one lane per (call, channel) the four
// dependent table loads of every output are paid in full (17 ms for ten seconds of stereo).  Pass 1: the interpolated, clamped
// sample of every output, all outputs in parallel (fp64, reference order) into scratch; pass 2: the recursive low-pass
// (ls = filtered s, :3324-3325) serially per job over contiguous doubles, 32 per round with the next 32 in flight.
template <int INTERP>
__global__ __launch_bounds__(256) void k_qoa_stream_interp(const QsJob *jobs, const unsigned long long *scr_off, const short *rows, double *scr, double ratio, double rcp,
                                                          int exact, int sinc_w) {
    const QsJob job = jobs[blockIdx.y];
    const short *src = rows + job.src_off;
    double m1 = 0, z0 = 0;
    if (job.last_off != ~0ull) { z0 = (double)rows[job.last_off]; m1 = (double)rows[job.last_off - 1]; }
    const int n = job.n;
    auto tap = [&](int k) -> double { return k >= 1 ? (double)src[k - 1] : (k == 0 ? z0 : m1); };
    double *o = scr + scr_off[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < job.nout; i += gridDim.x * 256) {
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
            else if constexpr (INTERP == AUKIT_INTERP_SINC) s = sinc_at(tap, k, -1, n, fx, sinc_w);
            else { const double p1 = tap(k), p0 = tap(k - 1), p2 = (k + 1 <= n) ? tap(k + 1) : p1, p3 = (k + 2 <= n) ? tap(k + 2) : p2; s = cubic_exact(p0, p1, p2, p3, fx); }
            s = lua_clamp(s, -128, 127);  // :3323
        }
        o[i] = s;
    }
}
template <typename OUT_T>
__global__ __launch_bounds__(64) void k_qoa_stream_iir(const QsJob *jobs, const unsigned long long *scr_off, unsigned long long njobs, const short *rows, const double *scr,
                                                      OUT_T *out, double lp_alpha) {
    const unsigned long long j = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= njobs) return;
    const QsJob job = jobs[j];
    const double *p = scr + scr_off[j];
    OUT_T *o = out + job.out_off;
    double ls = job.last_off != ~0ull ? (double)rows[job.last_off] : 0.0;  // :3316
    constexpr int R = 32;
    const int rounds = job.nout / R;
    double cur[R], nxt[R];
    if (rounds) {
#pragma unroll
        for (int k = 0; k < R; k++) cur[k] = p[k];
    }
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int k = 0; k < R; k++) nxt[k] = cur[k];
        if (r + 1 < rounds) {
#pragma unroll
            for (int k = 0; k < R; k++) nxt[k] = p[(r + 1) * R + k];
        }
        OUT_T res[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const double s = ls + lp_alpha * (cur[k] - ls);  // :3324
            ls = s;
            res[k] = (OUT_T)s;
        }
        {   // 16 bytes per store (a job's outputs start anywhere in the row: element-aligned vectors)
            constexpr int PV = 16 / (int)sizeof(OUT_T);
            typedef OUT_T ovp __attribute__((ext_vector_type(PV), aligned(sizeof(OUT_T))));
#pragma unroll
            for (int v = 0; v < R / PV; v++) {
                ovp w;
#pragma unroll
                for (int e = 0; e < PV; e++) w[e] = res[v * PV + e];
                *reinterpret_cast<ovp *>(o + r * R + v * PV) = w;
            }
        }
#pragma unroll
        for (int k = 0; k < R; k++) cur[k] = nxt[k];
    }
    for (int i = rounds * R; i < job.nout; i++) {
        const double s = ls + lp_alpha * (p[i] - ls);
        ls = s;
        o[i] = (OUT_T)s;
    }
}

struct QFrame { uint64_t off; int samples; };

// round 2's aukit.stream.qoa: the fallback of qoa.hip (frames of more than 8192 samples, sample rates below ≈ 300 Hz)
int stream_qoa_host(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks_out) {
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "stream.qoa: bad interpolation");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "stream.qoa output must be AUKIT_F64 or AUKIT_F32");
    if (in->n == 0) return fail(AUKIT_E_ARG, "empty batch");
    // (through the context's pinned staging buffer: a fresh 360 MB std::vector — zero-filled, then page-faulted in by a pageable copy — was
    // 80 of the 100 ms a 1024-stream call took)
    std::vector<uint8_t> host_pageable;
    uint8_t *host_p = static_cast<uint8_t *>(ctx_host_stage(ctx, (size_t)in->total() + 16));
    if (!host_p) { host_pageable.resize(in->total() + 16); host_p = host_pageable.data(); }
    struct HostView { uint8_t *p; uint8_t *data() const { return p; } } host{host_p};
    if (in->total()) AUKIT_HIP_CHECK(hipMemcpyAsync(host.data(), in->data(), in->total(), hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    int C = 0;
    double rate = 0;
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0); ck->status.assign(in->n, 0); ck->length_seconds.assign(in->n, 0);
    std::vector<std::vector<uint32_t>> clen(in->n);
    std::vector<std::vector<double>> cpos(in->n);
    std::vector<uint64_t> lens(in->n, 0);
    std::vector<QoaJob> djobs;
    struct Call { uint32_t stream; uint64_t row0, n, nout; };  // row0: element offset of channel 0's table index 1; channel c at row0 + c * cstride
    std::vector<Call> calls;
    std::vector<uint64_t> cstride;
    uint64_t tot = 0;
    double ratio = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint8_t *h = host.data() + in->off[s];
        const uint64_t nb = in->off[s + 1] - in->off[s];
        if (nb < 8) { delete ck; return fail(AUKIT_E_LUA, "Not a QOA file"); }                    // assert(read(8), ...)
        if (memcmp(h, "qoaf", 4) != 0) { delete ck; return fail(AUKIT_E_ARG, "Not a QOA file"); }
        const double file_samples = (double)((uint32_t)h[4] << 24 | (uint32_t)h[5] << 16 | (uint32_t)h[6] << 8 | h[7]);
        if (nb == 8) { delete ck; return fail(AUKIT_E_LUA, "Not a QOA file"); }                   // assert(peek(4), ...)
        if (nb < 12) { delete ck; return fail(AUKIT_E_LUA, "data string too short"); }
        const int fc = h[8];
        const double fr = (double)((uint32_t)h[9] << 16 | (uint32_t)h[10] << 8 | h[11]);
        if (fc < 1 || fc > AUKIT_MAX_PLANAR_CHANNELS) { delete ck; return fail(AUKIT_E_UNSUPPORTED, "QOA channel count %d", fc); }
        if (s == 0) { C = fc; rate = fr; ratio = 48000 / rate; }
        else if (fc != C || fr != rate) { delete ck; return fail(AUKIT_E_ARG, "all QOA streams of a batch must share channel count and sample rate"); }
        ck->length_seconds[s] = file_samples / fr;
        uint64_t pos = 8;
        double file_pos = (double)ctx->sb_samples;
        for (;;) {  // one iterator call
            std::vector<QFrame> frames;
            double sample_pos = 0;
            bool raised = false;
            while (sample_pos < fr) {
                if (pos >= nb) break;                                                              // read(8) → nil
                if (pos + 8 > nb) { raised = true; break; }
                const uint8_t *f = h + pos;
                const int channels = f[0];
                const double frate = (double)((uint32_t)f[1] << 16 | (uint32_t)f[2] << 8 | f[3]);
                const int samples = f[4] << 8 | f[5], frame_size = f[6] << 8 | f[7];
                const uint64_t fpos = pos;
                pos += 8;
                const int data_size = frame_size - 8 - 16 * channels;
                const int num_slices = (int)std::floor((double)data_size / 8);
                if (channels != fc || frate != fr || samples * channels > num_slices * 20) break;  // :3270-3277
                const uint64_t need = 16 * (uint64_t)channels + 8 * (uint64_t)((samples + 19) / 20) * channels;
                if (pos + need > nb) { raised = true; break; }                                     // assert(read(8), "Invalid QOA data") / short unpack
                frames.push_back(QFrame{fpos, samples});
                pos += need;
                sample_pos += samples;
            }
            if (raised) { ck->status[s] = AUKIT_E_LUA; break; }
            uint64_t n = 0, sp = 0;
            for (const QFrame &q : frames) { n = std::max<uint64_t>(n, sp + (uint64_t)((q.samples + 19) / 20) * 20); sp += (uint64_t)q.samples; }
            if (n == 0) break;                                                                      // #chunk[1] == 0 → nil
            const uint64_t stride = round_up(n + 2, 8);
            sp = 0;
            for (size_t k = 0; k < frames.size(); k++) {
                const bool lastf = k + 1 == frames.size();
                for (int c = 0; c < C; c++) {
                    QoaJob j;
                    j.frame_off = in->off[s] + frames[k].off; j.out_off = tot + (uint64_t)c * stride + sp;
                    j.c = c; j.channels = C; j.samples = frames[k].samples;
                    j.emit = lastf ? ((frames[k].samples + 19) / 20) * 20 : frames[k].samples;
                    djobs.push_back(j);
                }
                sp += (uint64_t)frames[k].samples;
            }
            const double newlen = (double)n * ratio;                                               // :3312
            const uint64_t nout = newlen >= 1 ? (uint64_t)std::floor(newlen) : 0;
            calls.push_back(Call{s, tot, n, nout});
            cstride.push_back(stride);
            clen[s].push_back((uint32_t)nout);
            cpos[s].push_back(file_pos / fr);                                                      // :3332
            file_pos += sample_pos;
            lens[s] += nout;
            tot += stride * C;
        }
        ck->nchunks[s] = (uint32_t)clen[s].size();
        ck->max_chunks = std::max(ck->max_chunks, ck->nchunks[s]);
    }
    const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
    ck->lens.assign((size_t)ck->n * mc, 0);
    ck->pos.assign((size_t)ck->n * mc, 0);
    for (uint32_t s = 0; s < in->n; s++)
        for (uint32_t k = 0; k < ck->nchunks[s]; k++) { ck->lens[(size_t)s * mc + k] = clen[s][k]; ck->pos[(size_t)s * mc + k] = cpos[s][k]; }
    int rc;
    aukit_audio *&full = ctx->stream_full;  // reused from call to call (audio_prepare keeps buffers that are large enough)
    aukit_audio **dst = mono && C > 1 ? &full : out;
    aukit_audio *a = *dst;
    if ((rc = audio_prepare(ctx, &a, in->n, C, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *dst = a;
    if (!calls.empty()) {
        if ((rc = ctx->tmp_buf.ensure((size_t)tot * 2 + 64))) { delete ck; return rc; }
        if ((rc = upload_table(ctx, ctx->tmp_buf2, djobs.data(), djobs.size() * sizeof(QoaJob)))) { delete ck; return rc; }
        hipLaunchKernelGGL(k_qoa, dim3((unsigned)((djobs.size() + 63) / 64)), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const QoaJob *>(ctx->tmp_buf2.p),
                           (unsigned long long)djobs.size(), reinterpret_cast<short *>(ctx->tmp_buf.p), 1);
        AUKIT_HIP_CHECK(hipGetLastError());
        std::vector<QsJob> jobs;
        std::vector<uint64_t> outpos(in->n, 0);
        std::vector<long long> prev(in->n, -1);
        for (size_t k = 0; k < calls.size(); k++) {
            const Call &cl = calls[k];
            for (int c = 0; c < C; c++) {
                QsJob j;
                j.src_off = cl.row0 + (uint64_t)c * cstride[k];
                j.last_off = ~0ull;
                if (prev[cl.stream] >= 0) { const Call &pc = calls[(size_t)prev[cl.stream]]; j.last_off = pc.row0 + (uint64_t)c * cstride[(size_t)prev[cl.stream]] + pc.n - 1; }
                j.out_off = a->row_off[cl.stream] + (uint64_t)c * a->row_stride[cl.stream] + outpos[cl.stream];
                j.n = (int)cl.n; j.nout = (int)cl.nout;
                jobs.push_back(j);
            }
            outpos[cl.stream] += cl.nout;
            prev[cl.stream] = (long long)k;
        }
        if ((rc = upload_table(ctx, ctx->seg_buf, jobs.data(), jobs.size() * sizeof(QsJob)))) { delete ck; return rc; }
        const double lp_alpha = 1 - std::exp(-(rate / 96000) * 2 * M_PI);  // :3251
        const int exact = exact_div_verified(ctx, ratio, 1ull << 18) ? 1 : 0;
        const unsigned grid = (unsigned)((jobs.size() + 63) / 64);
        const QsJob *dj = reinterpret_cast<const QsJob *>(ctx->seg_buf.p);
        const short *rows = reinterpret_cast<const short *>(ctx->tmp_buf.p);
        if ((rc = ctx_begin_kernel(ctx))) { delete ck; return rc; }
        uint64_t scr_elems = 0, max_nout = 0;
        std::vector<uint64_t> scr_off(jobs.size());
        for (size_t k = 0; k < jobs.size(); k++) { scr_off[k] = scr_elems; scr_elems += ((uint64_t)jobs[k].nout + 1) & ~1ull; max_nout = std::max<uint64_t>(max_nout, (uint64_t)jobs[k].nout); }
        // two passes whenever the scratch (one double per output) is affordable: the one-lane-per-job kernel pays four dependent table loads
        // and ~60 fp64 operations per output on a few hundred waves (1024 stereo streams of ten seconds: 94 ms against 7 ms in two passes)
        if (scr_elems * 8 <= (48ull << 30) && max_nout && !getenv("AUKIT_QOA_ONE_PASS")) {
            if ((rc = ctx->tmp_buf3.ensure((size_t)scr_elems * 8 + 64))) { delete ck; return rc; }
            if ((rc = upload_table(ctx, ctx->misc_buf, scr_off.data(), scr_off.size() * 8))) { delete ck; return rc; }
            const unsigned long long *dso = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
            double *scr = reinterpret_cast<double *>(ctx->tmp_buf3.p);
            for (size_t first = 0; first < jobs.size(); first += 65535) {  // (grid.y holds 65535 jobs)
                const dim3 g1((unsigned)std::min<uint64_t>((max_nout + 255) / 256, 1024), (unsigned)std::min<size_t>(65535, jobs.size() - first));
                if (interp == 0) hipLaunchKernelGGL((k_qoa_stream_interp<0>), g1, dim3(256), 0, ctx->stream, dj + first, dso + first, rows, scr, ratio, 1.0 / ratio, exact, ctx->sinc_w);
                else if (interp == 1) hipLaunchKernelGGL((k_qoa_stream_interp<1>), g1, dim3(256), 0, ctx->stream, dj + first, dso + first, rows, scr, ratio, 1.0 / ratio, exact, ctx->sinc_w);
                else if (interp == 2) hipLaunchKernelGGL((k_qoa_stream_interp<2>), g1, dim3(256), 0, ctx->stream, dj + first, dso + first, rows, scr, ratio, 1.0 / ratio, exact, ctx->sinc_w);
                else hipLaunchKernelGGL((k_qoa_stream_interp<3>), g1, dim3(256), 0, ctx->stream, dj + first, dso + first, rows, scr, ratio, 1.0 / ratio, exact, ctx->sinc_w);
            }
            if (dtype == AUKIT_F64) hipLaunchKernelGGL((k_qoa_stream_iir<double>), dim3(grid), dim3(64), 0, ctx->stream, dj, dso, (unsigned long long)jobs.size(), rows, scr, reinterpret_cast<double *>(a->dev), lp_alpha);
            else hipLaunchKernelGGL((k_qoa_stream_iir<float>), dim3(grid), dim3(64), 0, ctx->stream, dj, dso, (unsigned long long)jobs.size(), rows, scr, reinterpret_cast<float *>(a->dev), lp_alpha);
        } else {
#define AUKIT_QS(I, T) hipLaunchKernelGGL((k_qoa_stream<I, T>), dim3(grid), dim3(64), 0, ctx->stream, dj, (unsigned long long)jobs.size(), rows, reinterpret_cast<T *>(a->dev), ratio, 1.0 / ratio, exact, lp_alpha, ctx->sinc_w)
        if (dtype == AUKIT_F64) { if (interp == 0) AUKIT_QS(0, double); else if (interp == 1) AUKIT_QS(1, double); else if (interp == 2) AUKIT_QS(2, double); else AUKIT_QS(3, double); }
        else { if (interp == 0) AUKIT_QS(0, float); else if (interp == 1) AUKIT_QS(1, float); else if (interp == 2) AUKIT_QS(2, float); else AUKIT_QS(3, float); }
#undef AUKIT_QS
        }
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_qoa_stream", in->total()))) { delete ck; return rc; }
    }
    if (mono && C > 1) {  // lines[1][i] = (0 + s_1 + ... + s_C) / C  :3326-3329 — the same sum order as Audio:mono
        rc = aukit_mono(ctx, full, out);
        if (rc) { delete ck; return rc; }
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

}  // namespace aukit
