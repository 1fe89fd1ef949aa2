// ops.hip — the callers either side of the hot path: structural Audio methods
// (Audio:concat/sub/combine/split/rep/reverse, aukit.lua:690-866), the generators aukit.new / aukit.tone (:1779-1832) and
// aukit.pack / the packing half of Audio:wav (:1861-1878, :947-997).  All of it is data movement: every method becomes a list
// of row segments (copy / zero-fill / reversed copy) executed by one kernel, so a batch of N audios costs one launch.
#include <algorithm>
#include "common.h"

namespace aukit {

typedef unsigned long long u64;

struct RowJob {
    u64 dst;          // element offset in the output buffer
    const void *src;  // first source element (null: zeros)
    u64 len;
    int mode, pad;    // 0 copy, 1 zero, 2 reversed copy (dst[i] = src[len-1-i])
};

template <typename T>
__global__ __launch_bounds__(256) void k_row_jobs(const RowJob *jobs, T *out) {
    const RowJob j = jobs[blockIdx.y];
    const T *src = reinterpret_cast<const T *>(j.src);
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < j.len; i += (u64)gridDim.x * 256)
        out[j.dst + i] = j.mode == 1 ? (T)0 : (j.mode == 2 ? src[j.len - 1 - i] : src[i]);
}

static const void *row_ptr(const aukit_audio *a, uint32_t s, int c, uint64_t at = 0) {
    return reinterpret_cast<const char *>(a->dev) + (a->row_off[s] + (uint64_t)c * a->row_stride[s] + at) * dtype_size(a->dtype);
}

static int run_jobs(aukit_ctx *ctx, aukit_audio *o, const std::vector<RowJob> &jobs, const char *name) {
    if (jobs.empty()) { ctx->last_kernel = name; return AUKIT_OK; }
    int rc = upload_table(ctx, ctx->seg_buf, jobs.data(), jobs.size() * sizeof(RowJob));
    if (rc) return rc;
    ctx->plan_key.clear();
    u64 mx = 1, bytes = 0;
    for (const RowJob &j : jobs) { mx = std::max<u64>(mx, j.len); bytes += j.len * (j.mode == 1 ? 1 : 2) * dtype_size(o->dtype); }
    const unsigned gx = (unsigned)std::min<u64>((mx + 1023) / 1024, 1024);
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    for (size_t first = 0; first < jobs.size(); first += 65535) {  // gridDim.y limit
        const unsigned gy = (unsigned)std::min<size_t>(65535, jobs.size() - first);
        const RowJob *dj = reinterpret_cast<const RowJob *>(ctx->seg_buf.p) + first;
        if (o->dtype == AUKIT_F64) hipLaunchKernelGGL((k_row_jobs<double>), dim3(gx, gy), dim3(256), 0, ctx->stream, dj, reinterpret_cast<double *>(o->dev));
        else hipLaunchKernelGGL((k_row_jobs<float>), dim3(gx, gy), dim3(256), 0, ctx->stream, dj, reinterpret_cast<float *>(o->dev));
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, name, bytes);
}

static int check_group(const aukit_audio *const *in, uint32_t k, const aukit_audio *out) {
    if (!in || k == 0) return fail(AUKIT_E_ARG, "no input audio");
    for (uint32_t a = 0; a < k; a++) {
        if (!in[a]) return fail(AUKIT_E_ARG, "bad argument #%u (expected Audio, got nil)", a);
        if (in[a]->dtype != AUKIT_F64 && in[a]->dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "this operation needs an AUKIT_F64 / AUKIT_F32 audio");
        if (in[a]->dtype != in[0]->dtype || in[a]->n != in[0]->n) return fail(AUKIT_E_ARG, "all audios of one call must share dtype and stream count");
        if (in[a]->rate != in[0]->rate) return fail(AUKIT_E_ARG, "sample rates differ: resample first (the reference does, aukit.lua:702)");
        if (in[a] == out) return fail(AUKIT_E_ARG, "this operation cannot run in place");
    }
    return AUKIT_OK;
}

}  // namespace aukit

using namespace aukit;

extern "C" {

// Audio:concat(...)  aukit.lua:695-718
int aukit_concat(aukit_ctx *ctx, const aukit_audio *const *in, uint32_t count, aukit_audio **out) {
    if (!ctx || !out) return fail(AUKIT_E_ARG, "null argument");
    int rc = check_group(in, count, *out);
    if (rc) return rc;
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    for (uint32_t g = 0; g < count; g++) AUKIT_FLUSH(ctx, in[g]);
    const uint32_t n = in[0]->n;
    int cn = 0;
    for (uint32_t a = 0; a < count; a++) cn = std::max(cn, in[a]->channels);  // :703
    std::vector<uint64_t> lens(n, 0);
    for (uint32_t s = 0; s < n; s++)
        for (uint32_t a = 0; a < count; a++) lens[s] += in[a]->len[s];          // l[i] = #audios[i].data[1]  :697, :702
    aukit_audio *o = *out;
    if ((rc = audio_prepare(ctx, &o, n, cn, in[0]->rate, in[0]->dtype, lens.data()))) return rc;
    *out = o;
    std::vector<RowJob> jobs;
    for (uint32_t s = 0; s < n; s++)
        for (int c = 0; c < cn; c++) {
            uint64_t pos = 0;
            for (uint32_t a = 0; a < count; a++) {
                const uint64_t l = in[a]->len[s];
                if (l) {
                    const bool have = c < in[a]->channels;  // missing channels are silence  :711
                    jobs.push_back(RowJob{o->row_off[s] + (uint64_t)c * o->row_stride[s] + pos, have ? row_ptr(in[a], s, c) : nullptr, l, have ? 0 : 1, 0});
                }
                pos += l;
            }
        }
    return run_jobs(ctx, o, jobs, "k_row_jobs<concat>");
}

// Audio:sub(start, last)  aukit.lua:725-743.  Times in seconds, floored; negative / zero values count from the end.
int aukit_sub(aukit_ctx *ctx, const aukit_audio *in, double start, double last, aukit_audio **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    const aukit_audio *grp[1] = {in};
    int rc = check_group(grp, 1, *out);
    if (rc) return rc;
    AUKIT_FLUSH(ctx, in);
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    const double start_f = std::floor(start), last_f = std::floor(last);
    std::vector<uint64_t> lens(in->n), first(in->n);
    for (uint32_t s = 0; s < in->n; s++) {
        const double len = (double)in->len[s] / in->rate;
        double st = start_f, la = last_f;
        if (st < 0) st = len + st;
        if (la <= 0) la = len + la;
        if (!(st >= 0 && st <= len)) return fail(AUKIT_E_LUA, "number outside of range (expected %.14g to be within 0 and %.14g)", st, len);
        if (!(la >= 0 && la <= len)) return fail(AUKIT_E_LUA, "number outside of range (expected %.14g to be within 0 and %.14g)", la, len);
        const double i0 = st * in->rate + 1, i1 = la * in->rate + 1;  // for i = start, last do ch[i-start+1] = sch[i]  :738
        // sch[i] is nil for a non-integer i, so a fractional start yields an empty channel; otherwise indices i0, i0+1, ... <= i1 that exist
        uint64_t cnt = 0;
        if (i0 == std::floor(i0) && i1 >= i0 && i0 <= (double)in->len[s]) {
            const double top = std::min(std::floor(i1), (double)in->len[s]);
            cnt = (uint64_t)(top - i0 + 1);
        }
        lens[s] = cnt;
        first[s] = cnt ? (uint64_t)i0 - 1 : 0;
    }
    aukit_audio *o = *out;
    if ((rc = audio_prepare(ctx, &o, in->n, in->channels, in->rate, in->dtype, lens.data()))) return rc;
    *out = o;
    std::vector<RowJob> jobs;
    for (uint32_t s = 0; s < in->n; s++)
        for (int c = 0; c < in->channels; c++)
            if (lens[s]) jobs.push_back(RowJob{o->row_off[s] + (uint64_t)c * o->row_stride[s], row_ptr(in, s, c, first[s]), lens[s], 0, 0});
    return run_jobs(ctx, o, jobs, "k_row_jobs<sub>");
}

// Audio:combine(...)  aukit.lua:751-770: channels appended, shorter ones zero-extended
int aukit_combine(aukit_ctx *ctx, const aukit_audio *const *in, uint32_t count, aukit_audio **out) {
    if (!ctx || !out) return fail(AUKIT_E_ARG, "null argument");
    int rc = check_group(in, count, *out);
    if (rc) return rc;
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    for (uint32_t g = 0; g < count; g++) AUKIT_FLUSH(ctx, in[g]);
    const uint32_t n = in[0]->n;
    int cn = 0;
    for (uint32_t a = 0; a < count; a++) cn += in[a]->channels;
    if (cn > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "more than %d channels", AUKIT_MAX_PLANAR_CHANNELS);
    std::vector<uint64_t> lens(n, 0);
    for (uint32_t s = 0; s < n; s++)
        for (uint32_t a = 0; a < count; a++) lens[s] = std::max(lens[s], in[a]->len[s]);  // :757
    aukit_audio *o = *out;
    if ((rc = audio_prepare(ctx, &o, n, cn, in[0]->rate, in[0]->dtype, lens.data()))) return rc;
    *out = o;
    std::vector<RowJob> jobs;
    for (uint32_t s = 0; s < n; s++) {
        int pos = 0;
        for (uint32_t a = 0; a < count; a++) {
            for (int c = 0; c < in[a]->channels; c++) {
                const uint64_t d = o->row_off[s] + (uint64_t)(pos + c) * o->row_stride[s], l = in[a]->len[s];
                if (l) jobs.push_back(RowJob{d, row_ptr(in[a], s, c), l, 0, 0});
                if (lens[s] > l) jobs.push_back(RowJob{d + l, nullptr, lens[s] - l, 1, 0});  // sch[i] or 0  :764
            }
            pos += in[a]->channels;
        }
    }
    return run_jobs(ctx, o, jobs, "k_row_jobs<combine>");
}

// one result of Audio:split(...)  aukit.lua:781-797: `channels` holds 1-based channel numbers
int aukit_split(aukit_ctx *ctx, const aukit_audio *in, const int32_t *channels, uint32_t count, aukit_audio **out) {
    if (!ctx || !in || !out || (!channels && count)) return fail(AUKIT_E_ARG, "null argument");
    const aukit_audio *grp[1] = {in};
    int rc = check_group(grp, 1, *out);
    if (rc) return rc;
    AUKIT_FLUSH(ctx, in);
    if (count == 0) return fail(AUKIT_E_LUA, "bad argument #1 (cannot use empty table)");
    if (count > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "more than %d channels", AUKIT_MAX_PLANAR_CHANNELS);
    for (uint32_t k = 0; k < count; k++)
        if (channels[k] < 1 || channels[k] > in->channels) return fail(AUKIT_E_LUA, "channel %d (in argument 1) out of range", channels[k]);
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    aukit_audio *o = *out;
    if ((rc = audio_prepare(ctx, &o, in->n, (int)count, in->rate, in->dtype, in->len.data()))) return rc;
    *out = o;
    std::vector<RowJob> jobs;
    for (uint32_t s = 0; s < in->n; s++)
        for (uint32_t k = 0; k < count; k++)
            if (in->len[s]) jobs.push_back(RowJob{o->row_off[s] + (uint64_t)k * o->row_stride[s], row_ptr(in, s, channels[k] - 1), in->len[s], 0, 0});
    return run_jobs(ctx, o, jobs, "k_row_jobs<split>");
}

// Audio:rep(count)  aukit.lua:839-852: `for n = 0, count - 1` copies
int aukit_rep(aukit_ctx *ctx, const aukit_audio *in, double count, aukit_audio **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    const aukit_audio *grp[1] = {in};
    int rc = check_group(grp, 1, *out);
    if (rc) return rc;
    AUKIT_FLUSH(ctx, in);
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t reps = count >= 1 ? (uint64_t)std::floor(count) : 0;
    std::vector<uint64_t> lens(in->n);
    for (uint32_t s = 0; s < in->n; s++) lens[s] = in->len[s] * reps;
    aukit_audio *o = *out;
    if ((rc = audio_prepare(ctx, &o, in->n, in->channels, in->rate, in->dtype, lens.data()))) return rc;
    *out = o;
    std::vector<RowJob> jobs;
    for (uint32_t s = 0; s < in->n; s++)
        for (int c = 0; c < in->channels; c++)
            for (uint64_t r = 0; r < reps && in->len[s]; r++)
                jobs.push_back(RowJob{o->row_off[s] + (uint64_t)c * o->row_stride[s] + r * in->len[s], row_ptr(in, s, c), in->len[s], 0, 0});
    return run_jobs(ctx, o, jobs, "k_row_jobs<rep>");
}

// Audio:reverse()  aukit.lua:856-866
int aukit_reverse(aukit_ctx *ctx, const aukit_audio *in, aukit_audio **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    const aukit_audio *grp[1] = {in};
    int rc = check_group(grp, 1, *out);
    if (rc) return rc;
    AUKIT_FLUSH(ctx, in);
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    aukit_audio *o = *out;
    if ((rc = audio_prepare(ctx, &o, in->n, in->channels, in->rate, in->dtype, in->len.data()))) return rc;
    *out = o;
    std::vector<RowJob> jobs;
    for (uint32_t s = 0; s < in->n; s++)
        for (int c = 0; c < in->channels; c++)
            if (in->len[s]) jobs.push_back(RowJob{o->row_off[s] + (uint64_t)c * o->row_stride[s], row_ptr(in, s, c), in->len[s], 2, 0});
    return run_jobs(ctx, o, jobs, "k_row_jobs<reverse>");
}

}  // extern "C"

namespace aukit {

// aukit.pcm on a TABLE of numbers  aukit.lua:1077-1096 (the three read() closures), :1161-1171 (interleaved frames, or the channels one after
// the other): element e of a stream's table goes to (channel, index) and is normalised like an unpacked sample — in doubles, by division
template <typename T>
__global__ __launch_bounds__(256) void k_pcm_table(const double *vals, const u64 *voff, T *out, const u64 *row_off, const u64 *row_stride, int channels, int interleaved,
                                                   int data_type, double max_value) {
    const unsigned s = blockIdx.y;
    const u64 v0 = voff[s], total = voff[s + 1] - v0, len = total / (u64)channels;
    T *row = out + row_off[s];
    const u64 stride = row_stride[s];
    for (u64 e = (u64)blockIdx.x * 256 + threadIdx.x; e < total; e += (u64)gridDim.x * 256) {
        const double v = vals[v0 + e];
        double r;
        if (data_type == AUKIT_SIGNED) r = v / (v < 0 ? max_value : max_value - 1);                 // :1082
        else if (data_type == AUKIT_UNSIGNED) r = (v - 128) / (v < 128 ? max_value : max_value - 1);  // :1088 (Q4)
        else r = v;                                                                                   // :1094
        u64 j, i;
        if (interleaved && channels > 1) { i = e / (u64)channels; j = e - i * (u64)channels; }
        else { j = e / len; i = e - j * len; }
        row[j * stride + i] = (T)r;
    }
}

// wavegen  aukit.lua:286-299, evaluated at x = i / sampleRate, i = 1 .. duration * sampleRate  (:1826)
template <typename T>
__global__ __launch_bounds__(256) void k_tone(T *out, const u64 *row_off, u64 len, double rate, double freq, double amp, int wave, double duty) {
    T *row = out + row_off[blockIdx.y];
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < len; k += (u64)gridDim.x * 256) {
        const double x = (double)(k + 1) / rate;
        double v;
        switch (wave) {
        case AUKIT_WAVE_SINE: v = sin(2 * x * M_PI * freq) * amp; break;
        case AUKIT_WAVE_TRIANGLE: v = 2.0 * fabs(amp * fmod(2.0 * x * freq + 1.5, 2.0) - amp) - amp; break;
        case AUKIT_WAVE_SAWTOOTH: v = amp * fmod(2.0 * x * freq + 1.0, 2.0) - amp; break;
        case AUKIT_WAVE_SQUARE: { const double t = x * freq; v = (t - floor(t)) >= duty ? -amp : amp; break; }  // (x * freq) % 1
        default: v = 0;  // aukit.new
        }
        row[k] = (T)v;
    }
}

// aukit.noise  aukit.lua:1840-1853: l[i] = (random() * 2 - 1) * amplitude.  Philox4x32-10 (Salmon et al., SC'11), one block of four words per
// sample pair index: counter = (i, channel, stream, 0), key = the seed's halves; words 0-1 make the double of sample 2 i, words 2-3 of 2 i + 1
AUKIT_DEV void philox_round(unsigned &c0, unsigned &c1, unsigned &c2, unsigned &c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
template <typename T>
__global__ __launch_bounds__(256) void k_noise(T *out, const u64 *row_off, u64 len, double amp, unsigned k0, unsigned k1, int channels, unsigned stream0) {
    T *row = out + row_off[blockIdx.y];
    const unsigned stream = stream0 + blockIdx.y / (unsigned)channels, ch = blockIdx.y % (unsigned)channels;
    for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; 2 * q < len; q += (u64)gridDim.x * 256) {
        unsigned c0 = (unsigned)q, c1 = (unsigned)(q >> 32) ^ (ch << 8), c2 = stream, c3 = 0x4155u;  // ("AU")
        unsigned a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) { philox_round(c0, c1, c2, c3, a, b); a += 0x9E3779B9u; b += 0xBB67AE85u; }
        const double u0 = (double)(((u64)(c0 >> 5) << 26) | (c1 >> 6)) * 0x1.0p-53, u1 = (double)(((u64)(c2 >> 5) << 26) | (c3 >> 6)) * 0x1.0p-53;
        row[2 * q] = (T)((u0 * 2 - 1) * amp);
        if (2 * q + 1 < len) row[2 * q + 1] = (T)((u1 * 2 - 1) * amp);
    }
}

// string.pack of one sample (aukit.pack :1861-1878; Audio:wav :966-971).  `mode` says what the host VM does with a number
// that has no integer representation — the reference leaves that to string.pack, which is not part of aukit.lua.
// Four output elements per thread, 4 * BYTES contiguous bytes per store (the packed strings start anywhere: unaligned vector stores) —
// one byte per store instruction, the first version, moved 1.5 TB/s.
template <typename T, int BYTES>
__global__ __launch_bounds__(256) void k_pack(const T *in, const u64 *len, const u64 *off, const u64 *stride, int channels, int interleaved, unsigned char *out,
                                              const u64 *out_off, int data_type, int big_endian, int mode, double max_value, double add, int *err) {
    const unsigned s = blockIdx.y;
    const u64 L = len[s], total = L * (u64)channels;
    unsigned char *dst = out + out_off[s];
    const T *src = in + off[s];
    const u64 st = stride[s];
    auto enc = [&](u64 e) -> unsigned {  // the low BYTES bytes of element e in memory order (byte 0 = lowest address)
        u64 i, c;
        if (interleaved) { i = e / (u64)channels; c = e - i * (u64)channels; }  // data[(n-1)*nc+c]  :881
        else { c = e / L; i = e - c * L; }                                        // data[(c-1)*len+n]  :894
        const double d = (double)src[c * st + i];
        unsigned long long bits;
        if (data_type == AUKIT_FLOAT) bits = __float_as_uint((float)d);           // encode = identity, "f"
        else {
            const double v = (mode & AUKIT_PACK_PREENCODED) ? d : d * (d < 0 ? max_value : max_value - 1) + add;  // :875 (or already Audio:pcm's output)
            double r;
            if ((mode & 7) == AUKIT_PACK_FLOOR) r = floor(v);
            else if ((mode & 7) == AUKIT_PACK_TRUNC) r = trunc(v);
            else { r = v; if (v != floor(v)) atomicCAS(err, 0, 1); }            // PUC Lua 5.3: "number has no integer representation"
            if (!(r >= -9.2e18 && r <= 9.2e18)) { atomicCAS(err, 0, 1); r = 0; }
            bits = (unsigned long long)(long long)r;                              // two's complement, low `bytes` bytes kept (no range check in pack)
        }
        unsigned u = (unsigned)bits;
        if (BYTES < 4) u &= (1u << (8 * (BYTES & 3))) - 1u;
        if (big_endian) {
            if (BYTES == 2) u = ((u & 0xFF) << 8) | (u >> 8);
            else if (BYTES == 3) u = ((u & 0xFF) << 16) | (u & 0xFF00) | (u >> 16);
            else if (BYTES == 4) u = __builtin_bswap32(u);
        }
        return u;
    };
    typedef unsigned uvec __attribute__((ext_vector_type(BYTES == 3 ? 3 : (BYTES == 1 ? 1 : BYTES)), aligned(1)));
    const u64 quads = total / 4;
    for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q < quads; q += (u64)gridDim.x * 256) {
        const unsigned a0 = enc(4 * q), a1 = enc(4 * q + 1), a2 = enc(4 * q + 2), a3 = enc(4 * q + 3);
        unsigned char *o = dst + 4 * q * BYTES;
        if constexpr (BYTES == 1) { *reinterpret_cast<uvec *>(o) = uvec(a0 | a1 << 8 | a2 << 16 | a3 << 24); }
        else if constexpr (BYTES == 2) { uvec w; w.x = a0 | a1 << 16; w.y = a2 | a3 << 16; *reinterpret_cast<uvec *>(o) = w; }
        else if constexpr (BYTES == 3) { uvec w; w.x = a0 | a1 << 24; w.y = (a1 >> 8) | a2 << 16; w.z = (a2 >> 16) | a3 << 8; *reinterpret_cast<uvec *>(o) = w; }
        else { uvec w; w.x = a0; w.y = a1; w.z = a2; w.w = a3; *reinterpret_cast<uvec *>(o) = w; }
    }
    if (blockIdx.x == 0)
        for (u64 e = quads * 4 + threadIdx.x; e < total; e += 256) {
            const unsigned u = enc(e);
            for (int b = 0; b < BYTES; b++) dst[e * (u64)BYTES + b] = (unsigned char)(u >> (8 * b));
        }
}

}  // namespace aukit

extern "C" {

// aukit.new (wave = AUKIT_WAVE_NONE) / aukit.tone  aukit.lua:1783-1832: `n` identical audios
int aukit_tone(aukit_ctx *ctx, uint32_t n, double frequency, double duration, double amplitude, int wave, double duty, int channels, double sample_rate,
               int dtype, aukit_audio **out) {
    if (!ctx || !out) return fail(AUKIT_E_ARG, "null argument");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "dtype must be AUKIT_F64 or AUKIT_F32");
    if (wave != AUKIT_WAVE_NONE) {
        if (!(amplitude >= 0 && amplitude <= 1)) return fail(AUKIT_E_LUA, "number outside of range (expected %.14g to be within 0 and 1)", amplitude);
        if (wave < AUKIT_WAVE_SINE || wave > AUKIT_WAVE_SQUARE) return fail(AUKIT_E_LUA, "bad argument #4 (invalid wave type)");
        if (!(duty >= 0 && duty <= 1)) return fail(AUKIT_E_LUA, "number outside of range (expected %.14g to be within 0 and 1)", duty);
    }
    if (!(channels >= 1)) return fail(AUKIT_E_LUA, "number outside of range (expected %d to be within 1 and inf)", channels);
    if (channels > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "more than %d channels", AUKIT_MAX_PLANAR_CHANNELS);
    if (!(sample_rate >= 1)) return fail(AUKIT_E_LUA, "number outside of range (expected %.14g to be within 1 and inf)", sample_rate);
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    const double cnt = duration * sample_rate;  // for i = 1, duration * sampleRate
    const uint64_t len = cnt >= 1 ? (uint64_t)std::floor(cnt) : 0;
    std::vector<uint64_t> lens(n, len);
    aukit_audio *o = *out;
    int rc = audio_prepare(ctx, &o, n, channels, sample_rate, dtype, lens.data());
    if (rc) return rc;
    *out = o;
    if (!n || !len) return AUKIT_OK;
    std::vector<uint64_t> rows((size_t)n * channels);
    for (uint32_t s = 0; s < n; s++)
        for (int c = 0; c < channels; c++) rows[(size_t)s * channels + c] = o->row_off[s] + (uint64_t)c * o->row_stride[s];
    if ((rc = upload_table(ctx, ctx->misc_buf, rows.data(), rows.size() * 8))) return rc;
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    const unsigned gx = (unsigned)std::min<uint64_t>((len + 1023) / 1024, 1024);
    for (size_t first = 0; first < rows.size(); first += 65535) {
        const unsigned gy = (unsigned)std::min<size_t>(65535, rows.size() - first);
        const u64 *ro = reinterpret_cast<const u64 *>(ctx->misc_buf.p) + first;
        if (dtype == AUKIT_F64) hipLaunchKernelGGL((k_tone<double>), dim3(gx, gy), dim3(256), 0, ctx->stream, reinterpret_cast<double *>(o->dev), ro, (u64)len, sample_rate, frequency, amplitude, wave, duty);
        else hipLaunchKernelGGL((k_tone<float>), dim3(gx, gy), dim3(256), 0, ctx->stream, reinterpret_cast<float *>(o->dev), ro, (u64)len, sample_rate, frequency, amplitude, wave, duty);
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, "k_tone", (uint64_t)n * channels * len * dtype_size(dtype));
}

// aukit.noise  aukit.lua:1840-1853
int aukit_noise(aukit_ctx *ctx, uint32_t n, double duration, double amplitude, int channels, double sample_rate, uint64_t seed, int dtype, aukit_audio **out) {
    if (!ctx || !out) return fail(AUKIT_E_ARG, "null argument");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "dtype must be AUKIT_F64 or AUKIT_F32");
    if (!(amplitude >= 0 && amplitude <= 1)) return fail(AUKIT_E_LUA, "number outside of range (expected %.14g to be within 0 and 1)", amplitude);
    if (!(channels >= 1)) return fail(AUKIT_E_LUA, "number outside of range (expected %d to be within 1 and inf)", channels);
    if (channels > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "more than %d channels", AUKIT_MAX_PLANAR_CHANNELS);
    if (!(sample_rate >= 1)) return fail(AUKIT_E_LUA, "number outside of range (expected %.14g to be within 1 and inf)", sample_rate);
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    const double cnt = duration * sample_rate;  // for i = 1, duration * sampleRate
    const uint64_t len = cnt >= 1 ? (uint64_t)std::floor(cnt) : 0;
    std::vector<uint64_t> lens(n, len);
    aukit_audio *o = *out;
    int rc = audio_prepare(ctx, &o, n, channels, sample_rate, dtype, lens.data());
    if (rc) return rc;
    *out = o;
    if (!n || !len) return AUKIT_OK;
    std::vector<uint64_t> rows((size_t)n * channels);
    for (uint32_t s = 0; s < n; s++)
        for (int c = 0; c < channels; c++) rows[(size_t)s * channels + c] = o->row_off[s] + (uint64_t)c * o->row_stride[s];
    if ((rc = upload_table(ctx, ctx->misc_buf, rows.data(), rows.size() * 8))) return rc;
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    const unsigned gx = (unsigned)std::min<uint64_t>((len / 2 + 1024) / 1024, 1024);
    // (rows beyond gridDim.y's limit: the stream / channel of a row come from blockIdx.y, so a launch starts at a multiple of `channels`)
    const size_t per = (size_t)(65535 / channels) * channels;
    for (size_t first = 0; first < rows.size(); first += per) {
        const unsigned gy = (unsigned)std::min<size_t>(per, rows.size() - first);
        const u64 *ro = reinterpret_cast<const u64 *>(ctx->misc_buf.p) + first;
        const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32), stream0 = (unsigned)(first / channels);
        if (dtype == AUKIT_F64) hipLaunchKernelGGL((k_noise<double>), dim3(gx, gy), dim3(256), 0, ctx->stream, reinterpret_cast<double *>(o->dev), ro, (u64)len, amplitude, k0, k1, channels, stream0);
        else hipLaunchKernelGGL((k_noise<float>), dim3(gx, gy), dim3(256), 0, ctx->stream, reinterpret_cast<float *>(o->dev), ro, (u64)len, amplitude, k0, k1, channels, stream0);
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, "k_noise", (uint64_t)n * channels * len * dtype_size(dtype));
}

// aukit.pcm(data, bitDepth, dataType, channels, sampleRate, interleaved) with `data` a TABLE of numbers (aukit.lua:1077-1096): `n` tables
// as one host array of doubles + element offsets.  Values are used as they are (no range check, fractions allowed), as in the reference.
int aukit_decode_table(aukit_ctx *ctx, const double *values, const uint64_t *offsets, uint32_t n, const aukit_codec_desc *d, aukit_audio **out) {
    if (!ctx || !d || !out || (n && (!offsets || (offsets[n] && !values)))) return fail(AUKIT_E_ARG, "null argument");
    if (d->codec != AUKIT_CODEC_PCM) return fail(AUKIT_E_UNSUPPORTED, "table input: aukit.pcm only");
    if (d->bit_depth != 8 && d->bit_depth != 16 && d->bit_depth != 24 && d->bit_depth != 32) return fail(AUKIT_E_ARG, "bad argument #2 (invalid bit depth)");
    if (d->data_type < 0 || d->data_type > 2) return fail(AUKIT_E_ARG, "bad argument #3 (invalid data type)");
    if (d->data_type == AUKIT_FLOAT && d->bit_depth != 32) return fail(AUKIT_E_ARG, "bad argument #2 (float audio must have 32-bit depth)");
    if (d->channels < 1) return fail(AUKIT_E_ARG, "bad argument #4 (number outside of range)");
    if (!(d->sample_rate >= 1)) return fail(AUKIT_E_ARG, "bad argument #5 (number outside of range)");
    if (d->channels > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "more than %d channels", AUKIT_MAX_PLANAR_CHANNELS);
    const int dtype = ctx->dtype == AUKIT_F32 ? AUKIT_F32 : AUKIT_F64;
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<uint64_t> lens(n);
    for (uint32_t s = 0; s < n; s++) {
        if (offsets[s + 1] < offsets[s]) return fail(AUKIT_E_ARG, "offsets must not decrease");
        const uint64_t cnt = offsets[s + 1] - offsets[s];
        if (cnt % (uint64_t)d->channels != 0) return fail(AUKIT_E_ARG, "bad argument #1 (uneven amount of data per channel)");  // :1064
        lens[s] = cnt / (uint64_t)d->channels;
    }
    aukit_audio *o = *out;
    int rc = audio_prepare(ctx, &o, n, d->channels, d->sample_rate, dtype, lens.data());
    if (rc) return rc;
    *out = o;
    const uint64_t total = n ? offsets[n] - offsets[0] : 0;
    if (!total) return AUKIT_OK;
    if ((rc = ctx->tmp_buf.ensure((size_t)total * 8 + 64))) return rc;
    AUKIT_HIP_CHECK(hipMemcpyAsync(ctx->tmp_buf.p, values + offsets[0], (size_t)total * 8, hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint64_t> tab(3 * (size_t)n + 1);
    for (uint32_t s = 0; s <= n; s++) tab[s] = offsets[s] - offsets[0];
    for (uint32_t s = 0; s < n; s++) { tab[n + 1 + s] = o->row_off[s]; tab[2 * (size_t)n + 1 + s] = o->row_stride[s]; }
    if ((rc = upload_table(ctx, ctx->misc_buf, tab.data(), tab.size() * 8))) return rc;
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    uint64_t longest = 0;
    for (uint32_t s = 0; s < n; s++) longest = std::max<uint64_t>(longest, offsets[s + 1] - offsets[s]);
    const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((longest + 1023) / 1024, 1024));
    const u64 *t = reinterpret_cast<const u64 *>(ctx->misc_buf.p);
    const double max_value = std::ldexp(1.0, d->bit_depth - 1);
    for (uint32_t first = 0; first < n; first += 65535) {
        const unsigned gy = std::min<uint32_t>(65535, n - first);
        if (dtype == AUKIT_F64) hipLaunchKernelGGL((k_pcm_table<double>), dim3(gx, gy), dim3(256), 0, ctx->stream, reinterpret_cast<const double *>(ctx->tmp_buf.p), t + first, reinterpret_cast<double *>(o->dev), t + n + 1 + first, t + 2 * (size_t)n + 1 + first, d->channels, d->interleaved, d->data_type, max_value);
        else hipLaunchKernelGGL((k_pcm_table<float>), dim3(gx, gy), dim3(256), 0, ctx->stream, reinterpret_cast<const double *>(ctx->tmp_buf.p), t + first, reinterpret_cast<float *>(o->dev), t + n + 1 + first, t + 2 * (size_t)n + 1 + first, d->channels, d->interleaved, d->data_type, max_value);
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, "k_pcm_table", total * 8 + total * dtype_size(dtype));
}

// aukit.pack(audio:pcm(bitDepth, dataType, interleaved), bitDepth, dataType, bigEndian)  aukit.lua:901-910, :1861-1878 — the
// bytes Audio:wav writes after its header (:966-971: interleaved, unsigned for 8 bits, little-endian)
int aukit_pack_pcm(aukit_ctx *ctx, const aukit_audio *in, int bit_depth, int data_type, int big_endian, int interleaved, int int_mode, aukit_batch **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    if (in->dtype != AUKIT_F64 && in->dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "pack needs a float audio");
    AUKIT_FLUSH(ctx, in);
    if (bit_depth != 8 && bit_depth != 16 && bit_depth != 24 && bit_depth != 32) return fail(AUKIT_E_ARG, "bad argument #2 (invalid bit depth)");
    if (data_type < 0 || data_type > 2) return fail(AUKIT_E_ARG, "bad argument #3 (invalid data type)");
    if (data_type == AUKIT_FLOAT && bit_depth != 32) return fail(AUKIT_E_ARG, "bad argument #2 (float audio must have 32-bit depth)");
    if ((int_mode & 7) > AUKIT_PACK_STRICT || (int_mode & ~(7 | AUKIT_PACK_PREENCODED)) != 0) return fail(AUKIT_E_ARG, "bad integer conversion mode");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    const int bytes = bit_depth / 8;
    std::vector<uint64_t> off(in->n + 1, 0);
    for (uint32_t s = 0; s < in->n; s++) off[s + 1] = off[s] + in->len[s] * (uint64_t)in->channels * bytes;
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
    AUKIT_HIP_CHECK(hipMemcpyAsync(b->d_off, off.data(), ((size_t)in->n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    int *err = reinterpret_cast<int *>(b->d_off + in->n + 1);
    AUKIT_HIP_CHECK(hipMemsetAsync(err, 0, 8, ctx->stream));
    b->version++;
    *out = b;
    if (in->n == 0 || off[in->n] == 0) return AUKIT_OK;
    int rc = ctx_begin_kernel(ctx);
    if (rc) return rc;
    uint64_t mx = 1;
    for (uint32_t s = 0; s < in->n; s++) mx = std::max<uint64_t>(mx, in->len[s] * (uint64_t)in->channels);
    const unsigned gx = (unsigned)std::min<uint64_t>((mx + 1023) / 1024, 1024);
    const u64 *m = reinterpret_cast<const u64 *>(in->d_meta);
    const double maxv = std::ldexp(1.0, bit_depth - 1), add = data_type == AUKIT_UNSIGNED ? maxv : 0.0;
#define AUKIT_PACK_LAUNCH(T, BY)                                                                                                                          \
    hipLaunchKernelGGL((k_pack<T, BY>), dim3(gx, in->n), dim3(256), 0, ctx->stream, reinterpret_cast<const T *>(in->dev), m, m + in->n, m + 2 * (size_t)in->n, in->channels, \
                       interleaved, b->data(), reinterpret_cast<const u64 *>(b->d_off), data_type, big_endian, int_mode, maxv, add, err)
    if (in->dtype == AUKIT_F64) { if (bytes == 1) AUKIT_PACK_LAUNCH(double, 1); else if (bytes == 2) AUKIT_PACK_LAUNCH(double, 2); else if (bytes == 3) AUKIT_PACK_LAUNCH(double, 3); else AUKIT_PACK_LAUNCH(double, 4); }
    else { if (bytes == 1) AUKIT_PACK_LAUNCH(float, 1); else if (bytes == 2) AUKIT_PACK_LAUNCH(float, 2); else if (bytes == 3) AUKIT_PACK_LAUNCH(float, 3); else AUKIT_PACK_LAUNCH(float, 4); }
#undef AUKIT_PACK_LAUNCH
    AUKIT_HIP_CHECK(hipGetLastError());
    if ((rc = ctx_end_kernel(ctx, "k_pack", off[in->n] + (off[in->n] / bytes) * dtype_size(in->dtype)))) return rc;
    int herr = 0;
    AUKIT_HIP_CHECK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (herr) return fail(AUKIT_E_LUA, "bad argument #2 to 'pack' (number has no integer representation)");
    return AUKIT_OK;
}

}  // extern "C"
