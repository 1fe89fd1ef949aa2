// codecs2.hip — MS-ADPCM, QOA, MDFPWM loaders and the remaining stream.* factories (dfpwm, mdfpwm, msadpcm, qoa).
//
// The QOA LMS (aukit.lua:1686-1701) has a wrap inside the recurrence, so frames are sequential: one lane per (frame, channel);
// parallelism comes from frames × streams.  (MS-ADPCM lives in msadpcm.hip.)
#include <algorithm>
#include <chrono>
#include "resample.h"
#include "dfpwm_dev.h"

namespace aukit {

int audio_from_int_rows(aukit_ctx *ctx, int src_kind, const void *rows_dev, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len,
                        uint32_t n, int channels, double rate, double new_rate, int interp, bool do_resample, int dtype, double norm_pos,
                        double norm_neg, aukit_audio **out);

int stream_msadpcm(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks);  // msadpcm.hip

// ================================================================= QOA
__constant__ int c_qoa_dequant[16][8] = {
    {1, -1, 3, -3, 5, -5, 7, -7}, {5, -5, 18, -18, 32, -32, 49, -49}, {16, -16, 53, -53, 95, -95, 147, -147},
    {34, -34, 113, -113, 203, -203, 315, -315}, {63, -63, 210, -210, 378, -378, 588, -588}, {104, -104, 345, -345, 621, -621, 966, -966},
    {158, -158, 528, -528, 950, -950, 1477, -1477}, {228, -228, 760, -760, 1368, -1368, 2128, -2128},
    {316, -316, 1053, -1053, 1895, -1895, 2947, -2947}, {422, -422, 1405, -1405, 2529, -2529, 3934, -3934},
    {548, -548, 1828, -1828, 3290, -3290, 5117, -5117}, {696, -696, 2320, -2320, 4176, -4176, 6496, -6496},
    {868, -868, 2893, -2893, 5207, -5207, 8099, -8099}, {1064, -1064, 3548, -3548, 6386, -6386, 9933, -9933},
    {1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005}, {1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336}};  // :1662-1679

struct QoaJob {
    unsigned long long frame_off;  // byte offset of the frame header
    unsigned long long out_off;    // element offset (int16) of sample_pos + 1 in this channel's row
    int c, channels, samples, emit;  // emit: how many samples to store (20 * slices for the last frame of a table, `samples` otherwise)
};
AUKIT_DEV unsigned rdbe32(const unsigned char *p) { return (unsigned)p[0] << 24 | (unsigned)p[1] << 16 | (unsigned)p[2] << 8 | p[3]; }
AUKIT_DEV int rdbe16s(const unsigned char *p) { return (short)(p[0] << 8 | p[1]); }

__global__ __launch_bounds__(64) void k_qoa(const unsigned char *src, const QoaJob *jobs, unsigned long long njobs, short *out, int shift8) {
    const unsigned long long j = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= njobs) return;
    const QoaJob job = jobs[j];
    const unsigned char *f = src + job.frame_off;
    const unsigned char *l = f + 8 + 16 * job.c;
    long long h[4], w[4];
    for (int k = 0; k < 4; k++) { h[k] = rdbe16s(l + 2 * k); w[k] = rdbe16s(l + 8 + 2 * k); }
    const unsigned char *sl = f + 8 + 16 * job.channels;
    short *o = out + job.out_off;
    // One lane per (frame, channel): a slice is read as one unaligned 8-byte load and its 20 samples leave as 16 + 16 + 8 bytes (the rows
    // start anywhere: 2-byte aligned).  Sample by sample — 2-byte stores, 64 cache lines per store instruction for the 64 frames of a wave,
    // and eight byte loads per slice — the memory pipe of the CU was what the kernel waited for.
    typedef unsigned u32x2u __attribute__((ext_vector_type(2), aligned(1)));
    typedef unsigned u32x4h __attribute__((ext_vector_type(4), aligned(2)));
    typedef unsigned u32x2h __attribute__((ext_vector_type(2), aligned(2)));
    for (int si0 = 0; si0 < job.samples; si0 += 20) {
        const unsigned char *p = sl + 8 * ((size_t)(si0 / 20) * job.channels + job.c);
        const u32x2u raw = *reinterpret_cast<const u32x2u *>(p);
        unsigned hi = __builtin_bswap32(raw.x), lo = __builtin_bswap32(raw.y);
        const int sf = hi >> 28;
        unsigned pk[10];
#pragma unroll
        for (int k = 0; k < 20; k++) {
            const long long sum = w[0] * h[0] + w[1] * h[1] + w[2] * h[2] + w[3] * h[3];
            const int predicted = ((int)(unsigned)(unsigned long long)sum) >> 13;         // signed_rshift: wrap to int32, arithmetic shift :1681-1689
            const int deq = c_qoa_dequant[sf][(hi >> 25) & 7];
            int rec = predicted + deq;
            rec = rec < -32768 ? -32768 : (rec > 32767 ? 32767 : rec);
            const unsigned v16 = (unsigned)(shift8 ? (rec >> 8) : rec) & 0xFFFFu;         // stream.qoa: math.floor(reconstructed / 256) :3299
            if (k & 1) pk[k >> 1] |= v16 << 16; else pk[k >> 1] = v16;
            hi = (hi << 3) | (lo >> 29);
            lo <<= 3;
            const int delta = deq >> 4;                                                    // signed_rshift(residual, 4)
            for (int q = 0; q < 4; q++) w[q] += h[q] < 0 ? -delta : delta;                 // :1694-1699
            h[0] = h[1]; h[1] = h[2]; h[2] = h[3]; h[3] = rec;
        }
        if (si0 + 20 <= job.emit) {
            u32x4h a, b; u32x2h c;
            a.x = pk[0]; a.y = pk[1]; a.z = pk[2]; a.w = pk[3]; b.x = pk[4]; b.y = pk[5]; b.z = pk[6]; b.w = pk[7]; c.x = pk[8]; c.y = pk[9];
            *reinterpret_cast<u32x4h *>(o + si0) = a;
            *reinterpret_cast<u32x4h *>(o + si0 + 8) = b;
            *reinterpret_cast<u32x2h *>(o + si0 + 16) = c;
        } else {
            for (int k = 0; k < 20; k++) if (si0 + k < job.emit) o[si0 + k] = (short)((pk[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
        }
    }
}

struct QoaFrame { uint64_t off; int samples; };
// walks the frame headers exactly like the reference loops; `audio_mode` adds aukit.qoa's extra checks (:1720, :1735)
static int qoa_scan(const uint8_t *hdr, uint64_t nb, bool audio_mode, int *file_channels, double *file_rate, double *file_samples,
                    std::vector<QoaFrame> &frames, bool *raised) {
    *raised = false;
    if (nb < 8) return fail(AUKIT_E_LUA, audio_mode ? "data string too short" : "Not a QOA file");
    if (memcmp(hdr, "qoaf", 4) != 0) return fail(AUKIT_E_ARG, "Not a QOA file");
    *file_samples = (double)((uint32_t)hdr[4] << 24 | (uint32_t)hdr[5] << 16 | (uint32_t)hdr[6] << 8 | hdr[7]);
    if (nb < 12) return fail(AUKIT_E_LUA, nb == 8 && !audio_mode ? "Not a QOA file" : "data string too short");
    *file_channels = hdr[8];
    *file_rate = (double)((uint32_t)hdr[9] << 16 | (uint32_t)hdr[10] << 8 | hdr[11]);
    if (*file_channels < 1 || *file_channels > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "QOA channel count %d", *file_channels);
    uint64_t pos = 8;
    double sample_pos = 0;
    for (;;) {
        if (audio_mode) { if (!(pos + 1 + 16 * (uint64_t)*file_channels + 8 <= nb && sample_pos < *file_samples)) break; }
        else { if (pos >= nb) break; if (pos + 8 > nb) { *raised = true; break; } }
        const uint8_t *f = hdr + pos;
        const int channels = f[0];
        const double rate = (double)((uint32_t)f[1] << 16 | (uint32_t)f[2] << 8 | f[3]);
        const int samples = f[4] << 8 | f[5], frame_size = f[6] << 8 | f[7];
        const int data_size = frame_size - 8 - 16 * channels;
        const int num_slices = (int)std::floor((double)data_size / 8);
        if (channels != *file_channels || rate != *file_rate || samples * channels > num_slices * 20) break;
        if (audio_mode && (double)frame_size > (double)nb - (double)(pos + 8)) break;  // frame_size > #data - pos + 1, pos after the header (drops the last frame)
        const uint64_t need = 8 + 16 * (uint64_t)channels + 8 * (uint64_t)((samples + 19) / 20) * channels;
        if (pos + need > nb) { *raised = true; break; }  // "data string too short" / assert(read(8))
        frames.push_back(QoaFrame{pos, samples});
        pos += need;
        sample_pos += samples;
        (void)sample_pos;
    }
    return AUKIT_OK;
}

// round 2's aukit.qoa (host-side header walk on a copy of the batch, lane-per-job kernel): the fallback of qoa.hip
int decode_qoa_audio_host(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, double new_rate, int interp, bool do_resample, int dtype,
                          aukit_audio **out) {
    // headers are parsed on the host (8 bytes per 5120-sample frame); the batch is read back once for that
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
    std::vector<QoaJob> jobs;
    std::vector<uint64_t> row_off, row_len;
    uint64_t tot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        int fc; double fr, fs; bool raised;
        std::vector<QoaFrame> frames;
        int rc = qoa_scan(host.data() + in->off[s], in->off[s + 1] - in->off[s], true, &fc, &fr, &fs, frames, &raised);
        if (rc) return rc;
        if (raised) return fail(AUKIT_E_LUA, "data string too short");
        if (s == 0) { C = fc; rate = fr; }
        else if (fc != C || fr != rate) return fail(AUKIT_E_ARG, "all QOA streams of a batch must share channel count and sample rate");
        uint64_t L = 0, sp = 0;
        for (size_t k = 0; k < frames.size(); k++) { L = std::max<uint64_t>(L, sp + (uint64_t)((frames[k].samples + 19) / 20) * 20); sp += frames[k].samples; }
        const uint64_t stride = round_up(std::max<uint64_t>(L, 1), 8);
        sp = 0;
        for (size_t k = 0; k < frames.size(); k++) {
            const bool lastf = k + 1 == frames.size();
            for (int c = 0; c < C; c++) {
                QoaJob j;
                j.frame_off = in->off[s] + frames[k].off; j.out_off = tot + (uint64_t)c * stride + sp;
                j.c = c; j.channels = C; j.samples = frames[k].samples;
                j.emit = lastf ? ((frames[k].samples + 19) / 20) * 20 : frames[k].samples;  // Q15: the ≤19-sample tail survives only after the last frame
                jobs.push_back(j);
            }
            sp += frames[k].samples;
        }
        for (int c = 0; c < C; c++) { row_off.push_back(tot + (uint64_t)c * stride); row_len.push_back(L); }
        tot += stride * C;
    }
    if (in->n == 0) return fail(AUKIT_E_ARG, "empty batch");
    int rc = ctx->tmp_buf.ensure((size_t)tot * 2 + 64);
    if (rc) return rc;
    if ((rc = upload_table(ctx, ctx->tmp_buf2, jobs.data(), jobs.size() * sizeof(QoaJob)))) return rc;
    if (!jobs.empty()) {
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        hipLaunchKernelGGL(k_qoa, dim3((unsigned)((jobs.size() + 63) / 64)), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const QoaJob *>(ctx->tmp_buf2.p),
                           (unsigned long long)jobs.size(), reinterpret_cast<short *>(ctx->tmp_buf.p), 0);
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_qoa", in->total() + tot * 2))) return rc;
    }
    return audio_from_int_rows(ctx, SRC_I16, ctx->tmp_buf.p, row_off, row_len, in->n, C, rate, new_rate, interp, do_resample, dtype, 32767, 32768, out);
}

// ================================================================= MDFPWM  aukit.lua:1420-1448
struct DfInit2 { int on; int st[2][6]; };   // decoderL's / decoderR's state where the rest of a stream starts (aukit_ctx::sb_dfpwm / sb_dfpwm2)
__global__ __launch_bounds__(64) void k_mdfpwm_decode(const unsigned char *src, const unsigned long long *payload_off, const unsigned long long *payload_len,
                                                     unsigned n, signed char *out, const unsigned long long *row_off, const DfInit2 init) {
    const unsigned r = blockIdx.x * 64 + threadIdx.x;  // row = stream * 2 + channel: decoderL / decoderR are independent
    if (r >= 2 * n) return;
    const unsigned s = r >> 1, c = r & 1;
    const unsigned char *p = src + payload_off[s];
    const unsigned long long nb = payload_len[s];
    signed char *o = out + row_off[r];
    DfDec d{};
    if (init.on) { d.p.n = init.st[c][0]; d.p.strength = init.st[c][1]; d.p.pb = init.st[c][2]; d.lpf = init.st[c][3]; d.pn = init.st[c][4]; }
    unsigned long long w = 0;
    for (unsigned long long pos = 6000ull * c; pos < nb; pos += 12000) {
        const unsigned long long cnt = nb - pos < 6000 ? nb - pos : 6000;
        for (unsigned long long b = 0; b < cnt; b++) {
            unsigned byte = p[pos + b];
#pragma unroll
            for (int k = 0; k < 8; k++) { o[w++] = (signed char)df_decode_bit(d, byte & 1); byte >>= 1; }
        }
    }
}

bool dfpwm_decode_parallel_feed(aukit_ctx *ctx, const unsigned char *src, const std::vector<uint64_t> &h_off, const std::vector<uint64_t> &h_fed, uint64_t run,
                                uint64_t stride, int mode, int C, signed char *out, const unsigned long long *d_out_off, const unsigned long long *d_out_stride,
                                uint64_t lead, int *rc, const DfSliceHook *hook = nullptr);
struct MdHeader { uint64_t payload, length; };
static int mdfpwm_header(const uint8_t *h, uint64_t nb, MdHeader *out, const char *badmsg) {
    if (nb < 7 || memcmp(h, "MDFPWM\3", 7) != 0) return fail(AUKIT_E_ARG, "%s", badmsg);
    uint64_t pos = 7;
    if (pos + 4 > nb) return fail(AUKIT_E_LUA, "data string too short");
    out->length = (uint64_t)h[pos] | (uint64_t)h[pos + 1] << 8 | (uint64_t)h[pos + 2] << 16 | (uint64_t)h[pos + 3] << 24;
    pos += 4;
    for (int k = 0; k < 3; k++) {
        if (pos + 1 > nb) return fail(AUKIT_E_LUA, "data string too short");
        pos += 1 + (uint64_t)h[pos];
        if (pos > nb) return fail(AUKIT_E_LUA, "data string too short");
    }
    out->payload = pos;
    return AUKIT_OK;
}

// the first `take` bytes of every stream, side by side (one launch and one copy to the host instead of a copy per stream: 1024 streams
// spent 30 of their 33 ms in 1024 synchronous 300-byte copies)
__global__ __launch_bounds__(64) void k_gather_heads(const unsigned char *src, const unsigned long long *off, unsigned n, unsigned take, unsigned char *dst) {
    const unsigned s = blockIdx.x;
    if (s >= n) return;
    const unsigned long long nb = off[s + 1] - off[s];
    for (unsigned i = threadIdx.x; i < take; i += 64) dst[(size_t)s * take + i] = i < nb ? src[off[s] + i] : 0;
}

// decodes both channels of every stream into int8 rows in ctx->tmp_buf; returns per-row offsets / decoded lengths
static int mdfpwm_rows(aukit_ctx *ctx, const aukit_batch *in, const char *badmsg, std::vector<MdHeader> &hdrs, std::vector<uint64_t> &row_off,
                       std::vector<uint64_t> &row_len) {
    constexpr unsigned HEAD = 300;
    static const bool TT = getenv("AUKIT_HOST_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) { if (TT) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[mdfpwm_rows host] %-14s %8.1f us\n", w, std::chrono::duration<double, std::micro>(t - T0).count()); T0 = t; } };
    std::vector<uint8_t> heads_pageable;
    uint8_t *heads = nullptr;
    if (in->n) {
        int grc = ctx->tmp_buf3.ensure((size_t)in->n * HEAD + 64);
        if (grc) return grc;
        hipLaunchKernelGGL(k_gather_heads, dim3(in->n), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off), in->n, HEAD,
                           reinterpret_cast<unsigned char *>(ctx->tmp_buf3.p));
        AUKIT_HIP_CHECK(hipGetLastError());
        heads = static_cast<uint8_t *>(ctx_host_stage(ctx, (size_t)in->n * HEAD));
        if (!heads) { heads_pageable.resize((size_t)in->n * HEAD); heads = heads_pageable.data(); }
        lap("gather launch");
        AUKIT_HIP_CHECK(hipMemcpyAsync(heads, ctx->tmp_buf3.p, (size_t)in->n * HEAD, hipMemcpyDeviceToHost, ctx->stream));
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        lap("heads d2h+sync");
    }
    hdrs.resize(in->n);
    row_off.assign((size_t)in->n * 2, 0);
    row_len.assign((size_t)in->n * 2, 0);
    std::vector<uint64_t> tab((size_t)in->n * 4, 0);
    uint64_t tot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s], take = std::min<uint64_t>(nb, HEAD);
        const uint8_t *head = heads + (size_t)s * HEAD;
        int rc = mdfpwm_header(head, take == nb ? nb : std::max<uint64_t>(take, 300), &hdrs[s], badmsg);
        if (rc) return rc;
        if (hdrs[s].payload > nb) return fail(AUKIT_E_LUA, "data string too short");
        const uint64_t pl = nb - hdrs[s].payload;
        // bytes seen by decoderL / decoderR
        uint64_t bl = 0, br = 0;
        for (uint64_t pos = 0; pos < pl; pos += 12000) {
            bl += std::min<uint64_t>(6000, pl - pos);
            if (pos + 6000 < pl) br += std::min<uint64_t>(6000, pl - pos - 6000);
        }
        if (bl != br) return fail(AUKIT_E_UNSUPPORTED, "MDFPWM payload is not a whole number of L/R block pairs (the reference builds a table with holes)");
        const uint64_t stride = round_up(std::max<uint64_t>(bl * 8, 1), 16);
        row_off[(size_t)s * 2] = tot; row_off[(size_t)s * 2 + 1] = tot + stride;
        row_len[(size_t)s * 2] = row_len[(size_t)s * 2 + 1] = bl * 8;
        tab[s] = in->off[s] + hdrs[s].payload;
        tab[in->n + s] = pl;
        tab[2 * (size_t)in->n + 2 * s] = tot; tab[2 * (size_t)in->n + 2 * s + 1] = tot + stride;
        tot += 2 * stride;
    }
    lap("host headers");
    int rc = ctx->tmp_buf.ensure((size_t)tot + 64);
    if (rc) return rc;
    if ((rc = upload_table(ctx, ctx->misc_buf, tab.data(), tab.size() * 8))) return rc;  // tmp_buf2 is the parallel decoder's scratch
    lap("ensure+upload");
    if (in->n) {
        const unsigned long long *t = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        // decoderL / decoderR of a file are two independent decoders fed alternate 6000-byte blocks: 2n pseudo-streams for the
        // chunk-parallel exact decoder (dfpwm_par.hip)
        std::vector<uint64_t> p_off((size_t)in->n * 2), p_fed((size_t)in->n * 2);
        for (uint32_t s = 0; s < in->n; s++)
            for (int c = 0; c < 2; c++) { p_off[(size_t)s * 2 + c] = tab[s] + 6000ull * c; p_fed[(size_t)s * 2 + c] = row_len[(size_t)s * 2 + c] / 8; }
        int prc = AUKIT_OK;
        if (dfpwm_decode_parallel_feed(ctx, in->data(), p_off, p_fed, 6000, 12000, 0, 1, reinterpret_cast<signed char *>(ctx->tmp_buf.p), t + 2 * (size_t)in->n, nullptr, 0,
                                       &prc)) {
            if (prc) return prc;
            lap("par decode");
            return ctx_end_kernel(ctx, "k_df_chunks", in->total() + tot);
        }
        DfInit2 I2{};
        I2.on = ctx->sb_dfpwm_on ? 1 : 0;
        for (int i = 0; i < 6; i++) { I2.st[0][i] = ctx->sb_dfpwm[i]; I2.st[1][i] = ctx->sb_dfpwm2[i]; }
        hipLaunchKernelGGL(k_mdfpwm_decode, dim3((2 * in->n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), t, t + in->n, in->n,
                           reinterpret_cast<signed char *>(ctx->tmp_buf.p), t + 2 * (size_t)in->n, I2);
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_mdfpwm_decode", in->total() + tot))) return rc;
    }
    return AUKIT_OK;
}

int decode_mdfpwm_audio(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, double new_rate, int interp, bool do_resample, int dtype,
                        aukit_audio **out) {
    std::vector<MdHeader> hdrs;
    std::vector<uint64_t> row_off, row_len;
    int rc = mdfpwm_rows(ctx, in, "bad argument #1 (not a MDFPWM file)", hdrs, row_off, row_len);
    if (rc) return rc;
    for (uint32_t s = 0; s < in->n; s++) {  // for i = length * 8 + 1, #audio do audio[i] = nil  :1444 (interleaved table)
        const uint64_t inter = row_len[(size_t)s * 2] * 2, keep = std::min<uint64_t>(inter, hdrs[s].length * 8);
        row_len[(size_t)s * 2] = row_len[(size_t)s * 2 + 1] = keep / 2;
        if (keep % 2) return fail(AUKIT_E_ARG, "bad argument #1 (uneven amount of data per channel)");
    }
    return audio_from_int_rows(ctx, SRC_I8, ctx->tmp_buf.p, row_off, row_len, in->n, 2, 48000, new_rate, interp, do_resample, dtype, 127, 128, out);
}

// ================================================================= stream.dfpwm  aukit.lua:2439-2496
struct DfInit { int on; int st[6]; };   // the state a decoder starts from (a bounded reader-function handle's rest of a stream; aukit_ctx::sb_dfpwm)
__global__ __launch_bounds__(64) void k_dfpwm_stream_rows(const unsigned char *src, const unsigned long long *off, unsigned n, unsigned long long adv,
                                                         signed char *out, const unsigned long long *row_off, const DfInit init) {
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    const unsigned char *p = src + off[s];
    const unsigned long long nb = off[s + 1] - off[s];
    signed char *o = out + row_off[s];
    DfDec d{};
    if (init.on) { d.p.n = init.st[0]; d.p.strength = init.st[1]; d.p.pb = init.st[2]; d.lpf = init.st[3]; d.pn = init.st[4]; }
    unsigned long long w = 0;
    o[w++] = init.on ? (signed char)init.st[3] : 0;  // audio[0] of the first chunk = `last` = 0 — or the last sample before the rest of a stream (the decoder's low-pass state IS its last output)
    for (unsigned long long pos = 0; pos < nb; pos += adv) {  // str_sub(data, pos, pos + 6000 * channels): one byte of overlap
        const unsigned long long cnt = nb - pos < adv + 1 ? nb - pos : adv + 1;
        for (unsigned long long b = 0; b < cnt; b++) {
            unsigned byte = p[pos + b];
#pragma unroll
            for (int k = 0; k < 8; k++) { o[w++] = (signed char)df_decode_bit(d, byte & 1); byte >>= 1; }
        }
    }
}

bool dfpwm_decode_parallel(aukit_ctx *ctx, const aukit_batch *in, int mode, int C, signed char *out, const unsigned long long *d_out_off,
                           const unsigned long long *d_out_stride, int *rc, uint64_t adv, uint64_t lead, const DfSliceHook *hook = nullptr);
__global__ __launch_bounds__(256) void k_row_heads_zero(signed char *out, const unsigned long long *row_off, unsigned n, int head = 0) {
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s < n) out[row_off[s]] = (signed char)head;
}
// the decoder's state after `calls` iterator calls of stream.dfpwm (each feeds adv + 1 bytes and moves on by adv, :2456-2467) from `init`: what a
// bounded reader-function handle carries over the bytes it drops (stream_handle.hip).  One lane: 27 ns per sample, a few milliseconds per drop
__global__ void k_dfpwm_state_at(const unsigned char *p, unsigned long long calls, unsigned long long adv, const DfInit init, int *out6) {
    if (threadIdx.x || blockIdx.x) return;
    DfDec d{};
    if (init.on) { d.p.n = init.st[0]; d.p.strength = init.st[1]; d.p.pb = init.st[2]; d.lpf = init.st[3]; d.pn = init.st[4]; }
    for (unsigned long long k = 0; k < calls; k++)
        for (unsigned long long b = 0; b <= adv; b++) {
            unsigned byte = p[k * adv + b];
            for (int i = 0; i < 8; i++) { (void)df_decode_bit(d, byte & 1); byte >>= 1; }
        }
    out6[0] = d.p.n; out6[1] = d.p.strength; out6[2] = d.p.pb; out6[3] = d.lpf; out6[4] = d.pn; out6[5] = 0;
}
// stream.mdfpwm: decoderL / decoderR after `calls` block pairs (6000 bytes each, alternating; no overlap byte), lanes 0 and 1
__global__ void k_mdfpwm_state_at(const unsigned char *p, unsigned long long calls, const DfInit2 init, int *out12) {
    const int c = threadIdx.x;
    if (c > 1 || blockIdx.x) return;
    DfDec d{};
    if (init.on) { d.p.n = init.st[c][0]; d.p.strength = init.st[c][1]; d.p.pb = init.st[c][2]; d.lpf = init.st[c][3]; d.pn = init.st[c][4]; }
    for (unsigned long long k = 0; k < calls; k++)
        for (unsigned long long b = 0; b < 6000; b++) {
            unsigned byte = p[k * 12000 + 6000ull * c + b];
            for (int i = 0; i < 8; i++) { (void)df_decode_bit(d, byte & 1); byte >>= 1; }
        }
    int *o = out12 + 6 * c;
    o[0] = d.p.n; o[1] = d.p.strength; o[2] = d.p.pb; o[3] = d.lpf; o[4] = d.pn; o[5] = 0;
}
int mdfpwm_state_after(aukit_ctx *ctx, const unsigned char *dev_payload, uint64_t calls, const int *in12, bool in_on, int *out12) {
    DfInit2 I{};
    I.on = in_on ? 1 : 0;
    if (in_on) for (int i = 0; i < 12; i++) I.st[i / 6][i % 6] = in12[i];
    int rc = ctx->fmt_flag.ensure(256);
    if (rc) return rc;
    int *d12 = reinterpret_cast<int *>(ctx->fmt_flag.p) + 32;
    hipLaunchKernelGGL(k_mdfpwm_state_at, dim3(1), dim3(64), 0, ctx->stream, dev_payload, (unsigned long long)calls, I, d12);
    if (hipGetLastError() != hipSuccess) return fail(AUKIT_E_HIP, "k_mdfpwm_state_at launch failed");
    if (hipMemcpyAsync(out12, d12, 48, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(AUKIT_E_HIP, "reading the DFPWM states back failed");
    return AUKIT_OK;
}
int dfpwm_state_after(aukit_ctx *ctx, const unsigned char *dev_bytes, uint64_t calls, uint64_t adv, const int *in6, bool in_on, int *out6) {
    DfInit I{};
    I.on = in_on ? 1 : 0;
    if (in_on) for (int i = 0; i < 6; i++) I.st[i] = in6[i];
    int rc = ctx->fmt_flag.ensure(128);
    if (rc) return rc;
    int *d6 = reinterpret_cast<int *>(ctx->fmt_flag.p) + 16;
    hipLaunchKernelGGL(k_dfpwm_state_at, dim3(1), dim3(64), 0, ctx->stream, dev_bytes, (unsigned long long)calls, (unsigned long long)adv, I, d6);
    if (hipGetLastError() != hipSuccess) return fail(AUKIT_E_HIP, "k_dfpwm_state_at launch failed");
    if (hipMemcpyAsync(out6, d6, 24, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(AUKIT_E_HIP, "reading the DFPWM state back failed");
    return AUKIT_OK;
}

// stream.dfpwm on a 48 kHz file — what nearly every ComputerCraft DFPWM file is: ratio 1, every position x is an integer, `s = audio[x]`
// (:2481), the same sample goes to every output channel (Q11: x does not depend on j) and the mono mean of C equal integers is that integer.
// So a chunk's outputs are its decoded samples at stride C: a strided copy with a conversion, 16 bytes per store — instead of the general
// resampler with its stream.dfpwm epilogue (a position, a window in LDS and a clamp per output).
template <typename T>
__global__ __launch_bounds__(256) void k_dfpwm_stream_copy(const signed char *rows, const unsigned long long *row_off, const Seg *segs, T *out, int C, int nd) {
    const Seg g = segs[blockIdx.y];
    const signed char *src = rows + row_off[g.stream] + g.src_base + 1;  // audio[1]
    constexpr int PV = 16 / (int)sizeof(T);
    typedef T ovp __attribute__((ext_vector_type(PV), aligned(sizeof(T))));
    const unsigned groups = g.n_out / PV;
    for (unsigned q = blockIdx.x * 256 + threadIdx.x; q < groups; q += gridDim.x * 256) {
        ovp w;
#pragma unroll
        for (int e = 0; e < PV; e++) w[e] = (T)src[(unsigned long long)(q * PV + e) * (unsigned)C];
        for (int c = 0; c < nd; c++) *reinterpret_cast<ovp *>(out + g.out_off + (unsigned long long)c * g.out_stride + (unsigned long long)q * PV) = w;
    }
    if (blockIdx.x == 0)
        for (unsigned k = groups * PV + threadIdx.x; k < g.n_out; k += 256) {
            const T v = (T)src[(unsigned long long)k * (unsigned)C];
            for (int c = 0; c < nd; c++) out[g.out_off + (unsigned long long)c * g.out_stride + k] = v;
        }
}

bool dfpwm_stream_wave_try(aukit_ctx *ctx, int interp, double sample_rate, int channels, int rows_out, int mono, const std::vector<Seg> &segs, ResampleParams &P,
                           uint64_t algorithmic_bytes, int *rc);
static int stream_dfpwm(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out,
                        aukit_chunks **chunks_out) {
    const int C = d->channels;
    if (d->sample_rate < 1) return fail(AUKIT_E_ARG, "bad argument #2 (number outside of range)");
    if (C < 1 || C > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_ARG, "bad argument #3 (number outside of range)");
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "invalid interpolation");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "stream.dfpwm output must be AUKIT_F64 or AUKIT_F32");
    if (C == 1) mono = 0;
    const uint64_t adv = 6000ull * C, slice = adv + 1;
    const double ratio = 48000 / d->sample_rate;
    const int nd = mono ? 1 : C;
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0); ck->status.assign(in->n, 0); ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0), rowo(in->n, 0);
    std::vector<Seg> segs;
    std::vector<std::vector<uint32_t>> clen(in->n);
    uint64_t tot = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t nb = in->off[s + 1] - in->off[s];
        ck->length_seconds[s] = (double)(nb + ctx->sb_bytes) * 8 / d->sample_rate / C;
        rowo[s] = tot;
        uint64_t fed = 0;
        for (uint64_t pos = 0; pos < nb; pos += adv) {
            const uint64_t cnt = std::min<uint64_t>(slice, nb - pos), na = cnt * 8;
            const double newlen = (double)na * ratio;                         // :2474
            const uint32_t m = newlen >= 1 ? (uint32_t)(std::floor((newlen - 1) / C) + 1) : 0;  // for i = 1, newlen, channels
            Seg g;
            g.src_base = (long long)fed;  // row[fed] = audio[0] (previous chunk's last sample, or the leading 0)
            g.w_lo = 0; g.w_hi = (int)na; g.n_out = m; g.stream = s;
            g.out_off = lens[s]; g.out_stride = 0; g.pad = 0;
            segs.push_back(g);
            clen[s].push_back(m);
            lens[s] += m;
            fed += na;
        }
        ck->nchunks[s] = (uint32_t)clen[s].size();
        ck->max_chunks = std::max(ck->max_chunks, ck->nchunks[s]);
        tot += round_up(fed + 1, 16);
    }
    const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
    ck->lens.assign((size_t)ck->n * mc, 0);
    ck->pos.assign((size_t)ck->n * mc, 0);
    for (uint32_t s = 0; s < in->n; s++)
        for (uint32_t k = 0; k < ck->nchunks[s]; k++) {
            ck->lens[(size_t)s * mc + k] = clen[s][k];
            ck->pos[(size_t)s * mc + k] = (double)(ctx->sb_bytes + k * adv + 1) * 8 / d->sample_rate / C;  // p * 8 / sampleRate / channels :2494
        }
    int rc;
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, nd, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    for (Seg &g : segs) { g.out_off += a->row_off[g.stream]; g.out_stride = (unsigned)a->row_stride[g.stream]; }
    if (in->n && !segs.empty()) {
        if ((rc = ctx->tmp_buf.ensure((size_t)tot + 64))) { delete ck; return rc; }
        if ((rc = upload_table(ctx, ctx->misc_buf, rowo.data(), rowo.size() * 8))) { delete ck; return rc; }
        int prc = AUKIT_OK;
        const unsigned long long *d_rowo = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
        if (dfpwm_decode_parallel(ctx, in, 0, 1, reinterpret_cast<signed char *>(ctx->tmp_buf.p), d_rowo, nullptr, &prc, (uint64_t)adv, 1)) {  // chunk-parallel, exact (dfpwm_par.hip)
            if (prc) { delete ck; return prc; }
            hipLaunchKernelGGL(k_row_heads_zero, dim3((in->n + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<signed char *>(ctx->tmp_buf.p), d_rowo, in->n,
                               ctx->sb_dfpwm_on ? ctx->sb_dfpwm[3] : 0);  // audio[0] of the first chunk = `last` = 0 (or the last sample in front of the rest of a stream)
        } else {
        DfInit I{};
        I.on = ctx->sb_dfpwm_on ? 1 : 0;
        for (int i = 0; i < 6; i++) I.st[i] = ctx->sb_dfpwm[i];
        hipLaunchKernelGGL(k_dfpwm_stream_rows, dim3((in->n + 63) / 64), dim3(64), 0, ctx->stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off),
                           in->n, (unsigned long long)adv, reinterpret_cast<signed char *>(ctx->tmp_buf.p), reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p), I);
        }
        AUKIT_HIP_CHECK(hipGetLastError());
        if (d->sample_rate == 48000 && !getenv("AUKIT_NO_FAST_CONVERT")) {  // ratio 1: a strided copy (k_dfpwm_stream_copy)
            if ((rc = upload_table(ctx, ctx->seg_buf, segs.data(), segs.size() * sizeof(Seg)))) { delete ck; return rc; }
            ctx->plan_key.clear();  // (seg_buf no longer holds a resample plan)
            if ((rc = ctx_begin_kernel(ctx))) { delete ck; return rc; }
            uint32_t mmax = 0;
            for (const Seg &g : segs) mmax = std::max(mmax, g.n_out);
            const unsigned gx = (unsigned)std::max<uint32_t>(1, std::min<uint32_t>((mmax / 4 + 255) / 256, 64));
            const Seg *dsegs = reinterpret_cast<const Seg *>(ctx->seg_buf.p);
            for (size_t first = 0; first < segs.size(); first += 65535) {
                const dim3 grid(gx, (unsigned)std::min<size_t>(65535, segs.size() - first));
                if (dtype == AUKIT_F64) hipLaunchKernelGGL((k_dfpwm_stream_copy<double>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const signed char *>(ctx->tmp_buf.p), d_rowo, dsegs + first, reinterpret_cast<double *>(a->dev), C, nd);
                else hipLaunchKernelGGL((k_dfpwm_stream_copy<float>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const signed char *>(ctx->tmp_buf.p), d_rowo, dsegs + first, reinterpret_cast<float *>(a->dev), C, nd);
            }
            AUKIT_HIP_CHECK(hipGetLastError());
            uint64_t oe1 = 0;
            for (uint64_t l : lens) oe1 += l * nd;
            if ((rc = ctx_end_kernel(ctx, "k_dfpwm_stream_copy", in->total() + oe1 * dtype_size(dtype)))) { delete ck; return rc; }
            if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
            return AUKIT_OK;
        }
        ResampleParams P;
        memset(&P, 0, sizeof P);
        P.src = reinterpret_cast<const unsigned char *>(ctx->tmp_buf.p);
        P.src_off = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
        P.channels = 1;
        P.norm_pos = P.norm_neg = 1;
        P.pos_mul = C;
        P.mix_mono = mono ? 1 : 0;
        P.out_channels = C;
        P.out = a->dev;
        P.safe_lo = P.src;
        P.safe_hi = P.src + (size_t)tot + 64;
        uint64_t oe = 0;
        for (uint64_t l : lens) oe += l * nd;
        bool took = false;
        if (dtype == AUKIT_F32 && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {   // the wave kernel (fast_stream_dfpwm.hip)
            int wrc = AUKIT_OK;
            took = dfpwm_stream_wave_try(ctx, interp, d->sample_rate, C, nd, mono, segs, P, in->total() + oe * dtype_size(dtype), &wrc);
            if (took && wrc) { delete ck; return wrc; }
        }
        if (!took) {
            size_t lds;
            if ((rc = plan_tiles(ctx, segs, ratio, interp, 1, P, &lds))) { delete ck; return rc; }
            rc = launch_resample(ctx, SRC_I8, interp, EPI_STREAM_DFPWM, dtype, P, lds, in->total() + oe * dtype_size(dtype), nullptr);
            if (rc) { delete ck; return rc; }
        }
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

// ================================================================= stream.mdfpwm  aukit.lua:2507-2572
__global__ __launch_bounds__(256) void k_mdfpwm_chunks(const signed char *rows, const unsigned long long *row_off, const unsigned long long *nsamp, unsigned n, int mono,
                                                      signed char *out, const unsigned long long *ooff, const unsigned long long *ostride) {
    const unsigned s = blockIdx.y;
    const unsigned long long L = nsamp[s];
    const signed char *l = rows + row_off[2 * s], *r = rows + row_off[2 * s + 1];
    signed char *o0 = out + ooff[s], *o1 = o0 + ostride[s];
    // sixteen samples per thread and turn where the four rows allow it (decoder rows and audio rows start on 16-byte boundaries): the byte-per-thread
    // loop below alone took 1.1 of the call's 2.7 ms on 1024 files
    const bool vec = ((((uintptr_t)l | (uintptr_t)r | (uintptr_t)o0 | (uintptr_t)o1) & 15) == 0);
    const unsigned long long nv = vec ? L / 16 : 0;
    for (unsigned long long v = (unsigned long long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (unsigned long long)gridDim.x * 256) {
        const uint4 a = reinterpret_cast<const uint4 *>(l)[v], b = reinterpret_cast<const uint4 *>(r)[v];
        if (mono) {
            const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
            unsigned ow[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                unsigned acc = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int lv = (signed char)(aw[w] >> (8 * k)), rv = (signed char)(bw[w] >> (8 * k));
                    int m = lv + (rv >> 1);                                   // floor(L + R / 2) = L + floor(R / 2): L is an integer  (:2563)
                    m = m < -128 ? -128 : (m > 127 ? 127 : m);
                    acc |= (unsigned)(m & 0xFF) << (8 * k);
                }
                ow[w] = acc;
            }
            reinterpret_cast<uint4 *>(o0)[v] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        } else {
            reinterpret_cast<uint4 *>(o0)[v] = a;
            reinterpret_cast<uint4 *>(o1)[v] = b;
        }
    }
    for (unsigned long long i = nv * 16 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < L; i += (unsigned long long)gridDim.x * 256) {
        if (mono) {
            const double v = floor((double)l[i] + (double)r[i] / 2);  // clamp(floor(audioL[i] + audioR[i] / 2))  :2563
            o0[i] = (signed char)(int)(v < -128 ? -128 : (v > 127 ? 127 : v));
        } else {
            o0[i] = l[i];
            o1[i] = r[i];
        }
    }
}

static int stream_mdfpwm(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks_out) {
    if (dtype != AUKIT_I8) return fail(AUKIT_E_ARG, "stream.mdfpwm output must be AUKIT_I8");
    static const bool TT = getenv("AUKIT_HOST_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) { if (TT) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[stream.mdfpwm host] %-12s %8.1f us\n", w, std::chrono::duration<double, std::micro>(t - T0).count()); T0 = t; } };
    std::vector<MdHeader> hdrs;
    std::vector<uint64_t> row_off, row_len;
    int rc = mdfpwm_rows(ctx, in, "bad argument #1 (invalid MDFPWM data)", hdrs, row_off, row_len);
    if (rc) return rc;
    lap("rows");
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0); ck->status.assign(in->n, 0); ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0);
    for (uint32_t s = 0; s < in->n; s++) {
        const uint64_t pl = in->off[s + 1] - in->off[s] - hdrs[s].payload;
        ck->length_seconds[s] = (double)hdrs[s].length / 12000;
        // chunk k (12000-byte pair at payload offset 12000k) is delivered intact unless `pos - headerSize + 12000 > length` (Q12)
        uint32_t good = 0;
        for (uint64_t pos = 0; pos < pl; pos += 12000) {
            const bool full = pos + 12000 <= pl;
            const bool trimmed = (double)(ctx->sb_bytes + pos + 1) + 12000 > (double)hdrs[s].length;   // (pos - headerSize counts from the stream's first payload byte: what a bounded handle dropped included)
            if (!full || trimmed) { ck->status[s] = mono ? AUKIT_E_LUA : AUKIT_E_UNSUPPORTED; break; }
            good++;
        }
        ck->nchunks[s] = good;
        ck->max_chunks = std::max(ck->max_chunks, good);
        lens[s] = (uint64_t)good * 48000;
    }
    const uint32_t mc = std::max<uint32_t>(ck->max_chunks, 1);
    ck->lens.assign((size_t)ck->n * mc, 0);
    ck->pos.assign((size_t)ck->n * mc, 0);
    for (uint32_t s = 0; s < in->n; s++)
        for (uint32_t k = 0; k < ck->nchunks[s]; k++) { ck->lens[(size_t)s * mc + k] = 48000; ck->pos[(size_t)s * mc + k] = (double)(ctx->sb_bytes + 12000ull * k + 1) / 12000; }
    lap("chunk plan");
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, mono ? 1 : 2, 48000, AUKIT_I8, lens.data()))) { delete ck; return rc; }
    *out = a;
    lap("prepare");
    if (in->n) {
        std::vector<uint64_t> tab(row_off);
        tab.insert(tab.end(), lens.begin(), lens.end());
        if ((rc = upload_table(ctx, ctx->misc_buf, tab.data(), tab.size() * 8))) { delete ck; return rc; }
        const unsigned long long *t = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
        const unsigned long long *m = reinterpret_cast<const unsigned long long *>(a->d_meta);
        hipLaunchKernelGGL(k_mdfpwm_chunks, dim3(64, in->n), dim3(256), 0, ctx->stream, reinterpret_cast<const signed char *>(ctx->tmp_buf.p), t, t + 2 * (size_t)in->n,
                           in->n, mono ? 1 : 0, reinterpret_cast<signed char *>(a->dev), m + in->n, m + 2 * (size_t)in->n);
        AUKIT_HIP_CHECK(hipGetLastError());
        ctx->last_kernel = "k_mdfpwm_chunks";
    }
    lap("launch");
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

int stream_qoa(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks);
int stream_flac(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks);

int stream_more_codecs(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out,
                       aukit_chunks **chunks) {
    switch (d->codec) {
    case AUKIT_CODEC_DFPWM: return stream_dfpwm(ctx, in, d, interp, mono, dtype, out, chunks);
    case AUKIT_CODEC_MDFPWM: return stream_mdfpwm(ctx, in, d, mono, dtype, out, chunks);
    case AUKIT_CODEC_MSADPCM: return stream_msadpcm(ctx, in, d, interp, mono, dtype, out, chunks);
    case AUKIT_CODEC_QOA: return stream_qoa(ctx, in, d, interp, mono, dtype, out, chunks);
    case AUKIT_CODEC_FLAC: return stream_flac(ctx, in, d, interp, mono, dtype, out, chunks);
    case AUKIT_CODEC_ADPCM: return fail(AUKIT_E_ARG, "aukit.stream.adpcm takes WAV-style blocks: use AUKIT_CODEC_ADPCM_WAV");
    }
    return fail(AUKIT_E_ARG, "unknown stream codec %d", d->codec);
}

}  // namespace aukit
