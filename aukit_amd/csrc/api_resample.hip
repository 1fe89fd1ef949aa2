// api_resample.hip — host planners that expand C-ABI calls into segments for the fused resample engine:
//   aukit_decode / aukit_decode_resample  (PCM, G.711)        aukit.lua:1049-1171, :1361-1384, :653-673
//   aukit_resample                        (Audio:resample)    aukit.lua:653-673
//   aukit_stream_decode                   (stream.pcm, stream.g711)  aukit.lua:2228-2424, :2850-2913
// Block codecs (ADPCM, MS-ADPCM, DFPWM, QOA, FLAC) are routed to codecs.hip.
#include <algorithm>
#include "resample.h"

namespace aukit {

int decode_block_codec(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, double new_rate, int interp, bool do_resample,
                       int dtype, aukit_audio **out);
int stream_block_codec(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype,
                       aukit_audio **out, aukit_chunks **chunks);

static int check_pcm_desc(const aukit_codec_desc *d) {
    if (d->bit_depth != 8 && d->bit_depth != 16 && d->bit_depth != 24 && d->bit_depth != 32) return fail(AUKIT_E_ARG, "bad argument #2 (invalid bit depth)");
    if (d->data_type < 0 || d->data_type > 2) return fail(AUKIT_E_ARG, "bad argument #3 (invalid data type)");
    if (d->data_type == AUKIT_FLOAT && d->bit_depth != 32) return fail(AUKIT_E_ARG, "bad argument #2 (float audio must have 32-bit depth)");
    if (d->channels < 1) return fail(AUKIT_E_ARG, "bad argument #4 (number outside of range)");
    if (d->sample_rate < 1) return fail(AUKIT_E_ARG, "bad argument #5 (number outside of range)");
    if (d->channels > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_UNSUPPORTED, "at most %d channels are supported", AUKIT_MAX_PLANAR_CHANNELS);
    return AUKIT_OK;
}

static void fill_source(ResampleParams &P, const aukit_batch *in, const aukit_codec_desc *d) {
    P.src = in->data();
    P.src_off = reinterpret_cast<const unsigned long long *>(in->d_off);
    P.src_frames = nullptr;
    P.safe_lo = in->base;
    P.safe_hi = in->base + in->cap;  // the true end: a wrapped batch need not end on a 16-byte boundary, and its last samples are read one by one
    P.channels = d->channels;
    P.bit_depth = d->bit_depth;
    P.data_type = d->data_type;
    P.big_endian = d->big_endian;
    P.planar = 0;
    P.ulaw = d->ulaw;
    P.premix_mono = 0;
    P.mix_mono = 0;
    P.lp_alpha = 0;
    P.norm_pos = P.norm_neg = 1;
    P.g711_scale = 1.0 / 8192.0;
}

static bool all_even(const aukit_batch *in) {
    if (((uintptr_t)in->data()) & 1) return false;
    for (uint32_t s = 0; s < in->n; s++)
        if (in->off[s] & 1) return false;
    return true;
}

static int pick_pcm_source(const aukit_batch *in, const aukit_codec_desc *d, bool planar, bool premix) {
    if (d->bit_depth == 16 && d->data_type == AUKIT_SIGNED && !d->big_endian && d->channels == 1 && !planar && !premix && all_even(in)) return SRC_PCM_S16LE_MONO;
    return SRC_PCM_GENERIC;
}

// number of outputs of Audio:resample: `for i = 1, #data * ratio`  aukit.lua:659-664
static inline uint64_t resample_count(uint64_t n_in, double ratio) {
    double newlen = (double)n_in * ratio;
    return newlen >= 1 ? (uint64_t)std::floor(newlen) : 0;
}

// fused  aukit.pcm/g711(data, ...):resample(new_rate, interp)  (ratio == 1 → plain decode)
// ---- aukit.pcm without a resample behind it (aukit.lua:1097-1171: the unpack + normalise loop): every depth / type / byte order / layout as
// one coalesced pass — consecutive lanes write consecutive 16-byte vectors of one channel row and gather their samples (BYTES wide) from
// the interleaved or planar string.  The general kernel (k_resample, interpolation "none": a position per sample, the window staged
// through LDS) moved 1 TB/s on 16-bit stereo.  Same arithmetic: the integer, then s / (s < 0 and 2^(b-1) or 2^(b-1)-1) in doubles (:1133), (s - 128) / ... for
// unsigned (:1152, Q4), floats as they are (:1114).
struct UnpackRow { unsigned long long src, dst, frames, first, step, mix; };  // sample k of the row is `first + k * step` samples into the stream's bytes;
                                                                                 // mix > 1: the mean of `mix` consecutive samples from there (stream.pcm's mono, :2368)
template <int BYTES, typename T>
__global__ __launch_bounds__(256) void k_pcm_unpack(const unsigned char *src, const UnpackRow *rows, T *out, int data_type, int big_endian) {
    const UnpackRow r = rows[blockIdx.y];
    const unsigned char *p = src + r.src;
    T *o = out + r.dst;
    constexpr int PV = 16 / (int)sizeof(T);
    typedef T tvp __attribute__((ext_vector_type(PV), aligned(16)));  // rows of an Audio start on 64-byte boundaries (audio_prepare)
    const double maxv = (double)(1ull << (8 * BYTES - 1));
    auto one = [&](unsigned long long idx) -> double {
        const unsigned char *q = p + idx * BYTES;
        unsigned u = 0;
#pragma unroll
        for (int b = 0; b < BYTES; b++) u |= (unsigned)q[b] << (8 * (big_endian ? BYTES - 1 - b : b));
        if (data_type == AUKIT_FLOAT) return (double)__uint_as_float(u);
        double v;
        if (data_type == AUKIT_SIGNED) {
            const int sv = BYTES == 4 ? (int)u : ((int)(u << (32 - 8 * BYTES)) >> (32 - 8 * BYTES));
            v = (double)sv;
            return v / (v < 0 ? maxv : maxv - 1);
        }
        v = (double)u;
        return (v - 128) / (v < 128 ? maxv : maxv - 1);
    };
    auto conv = [&](unsigned long long k) -> T {
        const unsigned long long idx = r.first + k * r.step;
        if (r.mix <= 1) return (T)one(idx);
        double acc = 0;  // self[i] = ((0 + read()) + read() ...) / channels  :2368
        for (unsigned long long c = 0; c < r.mix; c++) acc = acc + one(idx + c);
        return (T)(acc / (double)r.mix);
    };
    const unsigned long long groups = r.frames / PV;
#pragma unroll 2
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (unsigned long long)gridDim.x * 256) {
        tvp w;
#pragma unroll
        for (int e = 0; e < PV; e++) w[e] = conv(PV * g + e);
        *reinterpret_cast<tvp *>(o + PV * g) = w;
    }
    if (blockIdx.x == 0) for (unsigned long long i = groups * PV + threadIdx.x; i < r.frames; i += 256) o[i] = conv(i);
}

static int decode_resample_flat(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, double new_rate, int interp, bool do_resample,
                                int dtype, aukit_audio **out) {
    const int C = d->channels;
    int rc;
    bool planar = false;
    if (d->codec == AUKIT_CODEC_PCM) {
        if ((rc = check_pcm_desc(d))) return rc;
        planar = !(d->interleaved && C > 1) && C > 1;  // aukit.lua:1156-1169
    } else {
        if (C < 1 || C > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_ARG, "channels out of range");
    }
    const size_t frame_bytes = d->codec == AUKIT_CODEC_PCM ? (size_t)(d->bit_depth / 8) * C : (size_t)C;
    const double ratio = do_resample ? new_rate / d->sample_rate : 1.0;  // :658
    if (!(ratio > 0)) return fail(AUKIT_E_ARG, "bad sample rate");
    std::vector<uint64_t> frames(in->n), lens(in->n);
    uint64_t in_bytes = 0, out_elems = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        uint64_t nb = in->off[s + 1] - in->off[s];
        if (nb % frame_bytes != 0) {
            if (d->codec == AUKIT_CODEC_PCM) return fail(AUKIT_E_ARG, "bad argument #1 (uneven amount of data per channel)");  // :1064
            return fail(AUKIT_E_UNSUPPORTED, "G.711 data length is not a multiple of the channel count (stream %u)", s);
        }
        frames[s] = nb / frame_bytes;
        lens[s] = do_resample ? resample_count(frames[s], ratio) : frames[s];
        if (frames[s] > 0x7FFFFFF0ull || lens[s] > 0xFFFFFFF0ull) return fail(AUKIT_E_UNSUPPORTED, "stream too long");  // Seg carries 32-bit counts: refuse before anything is allocated
        if (lens[s]) {
            double xl = host_pos(lens[s] - 1, ratio);
            if (std::floor(xl) > (double)frames[s]) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        }
        in_bytes += nb;
        out_elems += lens[s] * (uint64_t)C;
    }
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, C, do_resample ? new_rate : d->sample_rate, dtype, lens.data()))) return rc;
    *out = a;
    if (!do_resample && d->codec == AUKIT_CODEC_PCM && !getenv("AUKIT_NO_FAST_CONVERT")) {
        std::vector<UnpackRow> ur((size_t)in->n * C);
        uint64_t longest = 0;
        for (uint32_t s = 0; s < in->n; s++)
            for (int c = 0; c < C; c++) {
                UnpackRow &u = ur[(size_t)s * C + c];
                u.src = in->off[s];
                u.dst = a->row_off[s] + (uint64_t)c * a->row_stride[s];
                u.frames = frames[s];
                u.first = planar ? (uint64_t)c * frames[s] : (uint64_t)c;   // :1161-1169
                u.step = planar ? 1 : (uint64_t)C;
                u.mix = 0;
                longest = std::max(longest, frames[s]);
            }
        if (ur.empty() || longest == 0) return AUKIT_OK;
        if ((rc = upload_table(ctx, ctx->misc_buf, ur.data(), ur.size() * sizeof(UnpackRow)))) return rc;
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        const UnpackRow *d_ur = reinterpret_cast<const UnpackRow *>(ctx->misc_buf.p);
        const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((longest / 4 + 255) / 256 / 4, 64));
        const int bytes = d->bit_depth / 8;
        for (size_t first = 0; first < ur.size(); first += 65535) {
            const dim3 grid(gx, (unsigned)std::min<size_t>(65535, ur.size() - first));
#define AUKIT_UNPACK(BY)                                                                                                                                 \
            do {                                                                                                                                         \
                if (dtype == AUKIT_F64) hipLaunchKernelGGL((k_pcm_unpack<BY, double>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, reinterpret_cast<double *>(a->dev), d->data_type, d->big_endian ? 1 : 0); \
                else hipLaunchKernelGGL((k_pcm_unpack<BY, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, reinterpret_cast<float *>(a->dev), d->data_type, d->big_endian ? 1 : 0);              \
            } while (0)
            if (bytes == 1) AUKIT_UNPACK(1);
            else if (bytes == 2) AUKIT_UNPACK(2);
            else if (bytes == 3) AUKIT_UNPACK(3);
            else AUKIT_UNPACK(4);
#undef AUKIT_UNPACK
        }
        AUKIT_HIP_CHECK(hipGetLastError());
        return ctx_end_kernel(ctx, "k_pcm_unpack", in_bytes + out_elems * dtype_size(dtype));
    }
    std::vector<Seg> segs(in->n);
    for (uint32_t s = 0; s < in->n; s++) {
        Seg &g = segs[s];
        g.src_base = -1;  // table index 1 ↔ frame 0
        g.w_lo = 1;
        g.w_hi = (int)frames[s];
        g.n_out = (unsigned)lens[s];
        g.stream = s;
        g.out_off = a->row_off[s];
        g.out_stride = (unsigned)a->row_stride[s];
        g.pad = 0;
        if (frames[s] > 0x7FFFFFF0ull) return fail(AUKIT_E_UNSUPPORTED, "stream too long");
    }
    ResampleParams P;
    memset(&P, 0, sizeof P);
    fill_source(P, in, d);
    int src;
    if (d->codec == AUKIT_CODEC_PCM) {
        src = pick_pcm_source(in, d, planar, false);
        P.planar = planar ? 1 : 0;
        if (planar) {
            if ((rc = upload_table(ctx, ctx->misc_buf, frames.data(), frames.size() * sizeof(uint64_t)))) return rc;
            P.src_frames = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
        }
    } else {
        src = C == 1 ? SRC_G711_MONO : SRC_G711;
        P.g711_scale = 1.0 / 8192.0;  // m / 0x2000  :1379
    }
    P.out = a->dev;
    if (do_resample && dtype == AUKIT_F32 && C == 1) {  // HBM-bound tolerance path (fast.hip)
        int frc = AUKIT_OK;
        int fsrc = src;
        if (d->codec == AUKIT_CODEC_PCM && d->bit_depth == 8 && d->data_type != AUKIT_FLOAT) fsrc = SRC_PCM8_MONO;  // 8-bit mono: the wave kernel reads the bytes themselves
        if (fast_try(ctx, fsrc, interp, d->sample_rate, new_rate, segs, P, in_bytes + out_elems * 4, &frc)) return frc;
    }
    if (do_resample && dtype == AUKIT_F32 && C == 2 && d->codec == AUKIT_CODEC_PCM && !planar && d->bit_depth == 16 && d->data_type == AUKIT_SIGNED && !d->big_endian) {
        bool aligned4 = (((uintptr_t)in->data()) & 3) == 0;  // frames must not straddle dwords
        for (uint32_t s = 0; s < in->n && aligned4; s++) aligned4 = (in->off[s] & 3) == 0;
        int frc = AUKIT_OK;
        if (aligned4 && fast_try(ctx, SRC_PCM_S16LE_STEREO, interp, d->sample_rate, new_rate, segs, P, in_bytes + out_elems * 4, &frc)) return frc;
    }
    // every other interleaved format of one or two channels: decode + resample in one launch (fast_fmt.hip).  A float string gets the
    // reference-order kernel queued behind it, to run only if a sample beyond ±1 was met (see fast_fmt.hip)
    const int *only_if = nullptr;
    if (do_resample && dtype == AUKIT_F32 && !ctx->exact_math && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {
        int frc = AUKIT_OK;
        if (fast_fmt_try(ctx, d, interp, new_rate, segs, P, in_bytes + out_elems * 4, &frc, &only_if)) {
            if (frc || !only_if) return frc;
        }
    }
    // (only for formats whose samples lie in [-1, 1] — signed of any depth, 8-bit unsigned: where the reference's rounded position x misses an
    // integer it interpolates and CLAMPS (:667-668) what the exact rational position copies unclamped, and that shows on samples beyond ±1:
    // unsigned 16 / 24 / 32-bit (Q4 reaches 2) and float strings keep the reference-order kernel)
    if (do_resample && dtype == AUKIT_F32 && d->codec == AUKIT_CODEC_PCM && !ctx->exact_math && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC) &&
        (d->data_type == AUKIT_SIGNED || (d->data_type == AUKIT_UNSIGNED && d->bit_depth == 8)) && !only_if && !getenv("AUKIT_NO_FAST_CONVERT")) {
        // every other PCM format with a resample behind it, F32 tolerance path: unpacked to one f32 row per channel (k_pcm_unpack), then the f32
        // wave kernel on the rows — what stream.pcm does for these formats.  (k_resample moved 145-210 G samples/s on 24-bit stereo / float.)
        std::vector<UnpackRow> ur((size_t)in->n * C);
        std::vector<uint64_t> roff((size_t)in->n * C);
        uint64_t tot = 0, longest = 0;
        for (uint32_t s = 0; s < in->n; s++) {
            for (int c = 0; c < C; c++) {
                UnpackRow &u = ur[(size_t)s * C + c];
                u.src = in->off[s]; u.dst = tot; u.frames = frames[s];
                u.first = planar ? (uint64_t)c * frames[s] : (uint64_t)c; u.step = planar ? 1 : (uint64_t)C; u.mix = 0;
                roff[(size_t)s * C + c] = tot;
                tot += round_up(std::max<uint64_t>(frames[s], 1), 16) + 16;
            }
            longest = std::max(longest, frames[s]);
        }
        std::vector<Seg> rsegs;
        rsegs.reserve(segs.size() * (size_t)C);
        for (const Seg &g : segs)
            for (int c = 0; c < C; c++) {
                Seg r = g;
                r.stream = g.stream * (unsigned)C + (unsigned)c;
                r.out_off = g.out_off + (uint64_t)c * g.out_stride;
                r.out_stride = 0;
                rsegs.push_back(r);
            }
        if (longest) {
            if ((rc = ctx->tmp_buf.ensure((size_t)tot * 4 + 256))) return rc;
            if ((rc = upload_table(ctx, ctx->tmp_buf2, ur.data(), ur.size() * sizeof(UnpackRow)))) return rc;
            if ((rc = upload_table(ctx, ctx->misc_buf, roff.data(), roff.size() * sizeof(uint64_t)))) return rc;
            const UnpackRow *d_ur = reinterpret_cast<const UnpackRow *>(ctx->tmp_buf2.p);
            const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((longest / 4 + 255) / 256 / 4, 64));
            const int bytes = d->bit_depth / 8;
            if ((rc = ctx_begin_kernel(ctx))) return rc;
            for (size_t first = 0; first < ur.size(); first += 65535) {
                const dim3 grid(gx, (unsigned)std::min<size_t>(65535, ur.size() - first));
                float *rows = reinterpret_cast<float *>(ctx->tmp_buf.p);
                if (bytes == 1) hipLaunchKernelGGL((k_pcm_unpack<1, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, rows, d->data_type, d->big_endian ? 1 : 0);
                else if (bytes == 2) hipLaunchKernelGGL((k_pcm_unpack<2, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, rows, d->data_type, d->big_endian ? 1 : 0);
                else if (bytes == 3) hipLaunchKernelGGL((k_pcm_unpack<3, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, rows, d->data_type, d->big_endian ? 1 : 0);
                else hipLaunchKernelGGL((k_pcm_unpack<4, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, rows, d->data_type, d->big_endian ? 1 : 0);
            }
            AUKIT_HIP_CHECK(hipGetLastError());
            if ((rc = ctx_end_kernel(ctx, "k_pcm_unpack", in_bytes + tot * 4))) return rc;
            ResampleParams R;
            memset(&R, 0, sizeof R);
            R.src = reinterpret_cast<const unsigned char *>(ctx->tmp_buf.p);
            R.src_off = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
            R.channels = 1;
            R.out = a->dev;
            R.safe_lo = R.src;
            R.safe_hi = R.src + ctx->tmp_buf.cap;
            int frc = AUKIT_OK;
            if (fast_try(ctx, SRC_AUDIO_F32, interp, d->sample_rate, new_rate, rsegs, R, in_bytes + out_elems * 4, &frc)) return frc;
            if (planar) {  // not taken: the reference-order kernel below wants its frame table back
                if ((rc = upload_table(ctx, ctx->misc_buf, frames.data(), frames.size() * sizeof(uint64_t)))) return rc;
            }
        }
    }
    if (do_resample && dtype == AUKIT_F32 && C == 1 && ctx->exact_math == 1 && src == SRC_PCM_S16LE_MONO) {  // fp64 arithmetic, f32 store (wave_f64.hip)
        int wrc = AUKIT_OK;
        if (wave_f64_try(ctx, src, interp, d->sample_rate, new_rate, segs, P, in_bytes + out_elems * 4, &wrc)) return wrc;
    }
    if (do_resample && dtype == AUKIT_F32 && C == 1 && ctx->exact_math == 1 && src == SRC_G711_MONO) {  // the same for G.711 up-sampled by > 4.6 (wave_coef_f64.hip)
        int wrc = AUKIT_OK;
        if (wave_coef_f64_try(ctx, src, interp, d->sample_rate, new_rate, segs, P, in_bytes + out_elems * 4, &wrc)) return wrc;
    }
    if (do_resample && dtype == AUKIT_F32 && ctx->exact_math == 1 && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {
        // every other interleaved format in fp64 arithmetic with an f32 store (fast_fmt.hip with doubles in its tables)
        int frc = AUKIT_OK;
        if (fast_fmt_try(ctx, d, interp, new_rate, segs, P, in_bytes + out_elems * 4, &frc, &only_if)) {
            if (frc || !only_if) return frc;
        }
    }
    if (do_resample && C == 1 && src == SRC_PCM_S16LE_MONO && !only_if) {  // reference-order fp64 on wave tiles (exact_wave.hip)
        int erc = AUKIT_OK;
        if (exact_wave_try(ctx, src, interp, d->sample_rate, new_rate, segs, P, dtype, in_bytes + out_elems * dtype_size(dtype), &erc)) return erc;
    }
    size_t lds;
    if ((rc = plan_tiles(ctx, segs, ratio, do_resample ? interp : AUKIT_INTERP_NONE, C, P, &lds))) return rc;
    if (only_if) {  // the conditional redo behind k_fast_wave_fmt: the call is still named after the kernel that (normally) did the work
        const std::string nm = ctx->last_kernel;
        const float ms = ctx->last_ms;
        P.only_if = only_if;
        rc = launch_resample(ctx, src, interp, EPI_AUDIO, dtype, P, lds, 0, nullptr);
        ctx->last_kernel = nm;
        ctx->last_ms += ms;
        ctx->last_bytes = in_bytes + out_elems * dtype_size(dtype);
        return rc;
    }
    return launch_resample(ctx, src, do_resample ? interp : AUKIT_INTERP_NONE, EPI_AUDIO, dtype, P, lds, in_bytes + out_elems * dtype_size(dtype), nullptr);
}

// ---------------------------------------------------------------- stream.pcm  aukit.lua:2228-2424
struct ChunkPlan {
    double ratio;
    int interp;
    long K;                        // table re-base per full chunk (Q1)
    std::vector<int> acc, req;     // per output j (1-based → [j-1]): highest index touched so far / index that must be non-nil
};

static int build_chunk_plan(double sample_rate, int interp, ChunkPlan &cp) {
    static const int iend[4] = {1, 2, 3, 0};
    cp.ratio = 48000 / sample_rate;  // :2364
    cp.interp = interp;
    cp.acc.resize(48000);
    cp.req.resize(48000);
    int mx = iend[interp];
    for (int j = 1; j <= 48000; j++) {
        double x = ((double)(j - 1) / cp.ratio) + 1;
        double ffx = std::floor(x);
        bool isint = x == ffx;
        int top = isint ? (int)x : (int)ffx + (interp == AUKIT_INTERP_CUBIC ? 2 : (interp == AUKIT_INTERP_LINEAR ? 1 : 0));
        if (top > mx) mx = top;
        cp.acc[j - 1] = mx;
        cp.req[j - 1] = (int)ffx;
    }
    cp.K = mx;
    return AUKIT_OK;
}

long stream_pcm_call_frames(double sample_rate, int interp) {
    ChunkPlan cp;
    build_chunk_plan(sample_rate, interp, cp);
    return cp.K;
}

static int stream_pcm(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out,
                      aukit_chunks **chunks_out, bool table = false) {
    int rc;
    if ((rc = check_pcm_desc(d))) return rc;
    if (interp == AUKIT_INTERP_SINC) return fail(AUKIT_E_UNSUPPORTED, "stream.pcm with sinc interpolation reads its lazy table out of order (not reproduced on the GPU)");
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "invalid interpolation");
    if (d->sample_rate > 48000) return fail(AUKIT_E_UNSUPPORTED, "stream.pcm above 48 kHz is ill-defined in the reference (lazy table read out of order, SURVEY Q3)");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "stream.pcm output must be AUKIT_F64 or AUKIT_F32");
    const int C = d->channels;
    if (C == 1) mono = 0;  // :2243
    const int nd = mono ? 1 : C;
    const int bd = table ? 8 : d->bit_depth / 8;   // a table entry is one number (`len = #data / channels`, :2245)
    static const int istart[4] = {1, 1, 0, 0}, iend[4] = {1, 2, 3, 0};  // :283-284
    const bool is_float = d->data_type == AUKIT_FLOAT;
    // the plan depends on the batch's layout and the descriptor alone: the same batch coming back (austream's loop, bench.py's steps) reuses it
    char keyb[256];
    unsigned long long oh = 1469598103934665603ull;   // FNV-1a over the stream offsets (a freed batch's address can come back with another layout)
    for (uint64_t o : in->off) { oh ^= o; oh *= 1099511628211ull; }
    snprintf(keyb, sizeof keyb, "%p/%llu/%u/%llx/%d/%d/%d/%d/%.17g/%d/%d/%d/%llu/%llu", (const void *)in, (unsigned long long)in->version, in->n, oh,
             d->bit_depth + (table ? 1000 : 0), d->data_type, d->big_endian, C, d->sample_rate, interp, mono, nd, (unsigned long long)ctx->sb_bytes, (unsigned long long)ctx->sb_outputs);
    std::vector<Seg> segs;
    std::vector<uint64_t> lens(in->n, 0);
    uint64_t in_bytes = 0, out_elems = 0;
    aukit_chunks *ck = new aukit_chunks();
    const bool plan_hit = ctx->spcm_ck && ctx->spcm_key == keyb && !getenv("AUKIT_NO_PLAN_CACHE");
    double cp_ratio = 48000 / d->sample_rate;
    if (plan_hit) {
        *ck = *ctx->spcm_ck;
        segs.resize(ctx->spcm_segs.size() / sizeof(Seg));
        if (!segs.empty()) memcpy(segs.data(), ctx->spcm_segs.data(), ctx->spcm_segs.size());
        lens = ctx->spcm_lens;
        in_bytes = ctx->spcm_in_bytes; out_elems = ctx->spcm_out_elems;
    } else {
    ChunkPlan cp;
    build_chunk_plan(d->sample_rate, interp, cp);
    cp_ratio = cp.ratio;

    ck->n = in->n;
    ck->nchunks.assign(in->n, 0);
    ck->status.assign(in->n, 0);
    ck->length_seconds.assign(in->n, 0);
    std::vector<std::vector<uint32_t>> clens(in->n);
    for (uint32_t s = 0; s < in->n; s++) {
        uint64_t nb = in->off[s + 1] - in->off[s];
        // Data that ends inside a frame (round 6).  With the mono mix-down every index is read for ALL channels (`self[i] = (rawget(self, i) or 0) + read()`,
        // :2368): the frame a channel is missing from raises where a frame missing altogether would — inside the pcall (the chunk ends there, :2389-2407) or,
        // in the prefill, as `if not c then return nil end` (:2377-2384): the partial frame counts for nothing.  WITHOUT the mix-down the channels in front
        // of the gap get one output more than the others in the last chunk (`for i ... for y`: they are written before the missing one raises) — chunk
        // lengths per channel, which this ABI does not have: refused, as is data that ends inside a SAMPLE (`string.rep` with a fractional count is the VM's business).
        if (nb % ((size_t)bd * C) != 0) {
            if (nb % (size_t)bd != 0) { delete ck; return fail(AUKIT_E_UNSUPPORTED, "stream.pcm: data ends inside a sample (stream %u)", s); }
            if (!mono) { delete ck; return fail(AUKIT_E_UNSUPPORTED, "stream.pcm: data ends inside a frame and the channels are not mixed down: the reference's last chunk is longer in its first channels (stream %u)", s); }
        }
        const long long nframes = (long long)(nb / ((size_t)bd * C));
        ck->length_seconds[s] = ((double)(nb + ctx->sb_bytes) / bd) / C / d->sample_rate;  // :2245, :2423 (sb_bytes: what a stream handle has dropped already)
        in_bytes += nb;
        for (long c = 0;; c++) {
            const long long src_base = (long long)c * cp.K - istart[interp];  // frame of table index 0
            // prefill :2376-2386 reads eagerly up to index iend
            if (src_base + iend[interp] > nframes - 1) { ck->status[s] = is_float ? 0 : AUKIT_E_LUA; break; }
            const long long w_avail = nframes - 1 - src_base;
            // float strings: read() hands out nil past the end (:2291-2311) and interpolate falls back on its neighbours, so only the
            // floor index must exist; with a mono mix-down of several channels the lazy __index adds that nil (:2368) and raises
            // like the integer readers do on their first read past the end
            const std::vector<int> &need = (is_float && !mono) ? cp.req : cp.acc;
            uint32_t n_out = (uint32_t)(std::upper_bound(need.begin(), need.end(), (int)std::min<long long>(w_avail, 0x7FFFFFFF)) - need.begin());
            if (n_out == 0) break;  // :2407
            Seg g;
            g.src_base = src_base;
            g.w_lo = c == 0 ? istart[interp] : -1;
            g.w_hi = (int)std::min<long long>(w_avail, 0x7FFFFFF0);
            g.n_out = n_out;
            g.stream = s;
            g.out_off = lens[s];  // patched with the row offset below
            g.out_stride = 0;
            g.pad = 0;
            segs.push_back(g);
            clens[s].push_back(n_out);
            lens[s] += n_out;
            if (n_out < 48000) break;  // ok = false → next call returns nil
        }
        ck->nchunks[s] = (uint32_t)clens[s].size();
        ck->max_chunks = std::max<uint32_t>(ck->max_chunks, ck->nchunks[s]);
        out_elems += lens[s] * (uint64_t)nd;
    }
    ck->lens.assign((size_t)ck->n * std::max<uint32_t>(ck->max_chunks, 1), 0);
    ck->pos.assign((size_t)ck->n * std::max<uint32_t>(ck->max_chunks, 1), 0);
    for (uint32_t s = 0; s < in->n; s++) {
        double nacc = (double)ctx->sb_outputs;
        for (uint32_t k = 0; k < ck->nchunks[s]; k++) {
            ck->lens[(size_t)s * ck->max_chunks + k] = clens[s][k];
            ck->pos[(size_t)s * ck->max_chunks + k] = nacc / 48000;  // (n - #chunk[1]) / 48000  :2422
            nacc += clens[s][k];
        }
    }
    ctx->spcm_key = keyb;
    ctx->spcm_segs.assign(reinterpret_cast<const unsigned char *>(segs.data()), reinterpret_cast<const unsigned char *>(segs.data()) + segs.size() * sizeof(Seg));
    ctx->spcm_lens = lens;
    ctx->spcm_in_bytes = in_bytes; ctx->spcm_out_elems = out_elems;
    if (!ctx->spcm_ck) ctx->spcm_ck = new aukit_chunks();
    *ctx->spcm_ck = *ck;
    }  // (plan)
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, nd, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    for (Seg &g : segs) {
        g.out_off += a->row_off[g.stream];
        g.out_stride = (unsigned)a->row_stride[g.stream];
    }
    ResampleParams P;
    memset(&P, 0, sizeof P);
    fill_source(P, in, d);
    P.premix_mono = mono ? 1 : 0;
    P.table = table ? 1 : 0;
    P.lp_alpha = 1 - std::exp(-(d->sample_rate / 96000) * 2 * M_PI);  // :2365
    P.out = a->dev;
    int src = table ? SRC_PCM_GENERIC : pick_pcm_source(in, d, false, mono != 0);
    bool done = false;
    if (table) {   // the numbers of a table are not bytes: only the generic staging reads them (reference order, fp64)
        size_t lds;
        if ((rc = plan_tiles(ctx, segs, cp_ratio, interp, nd, P, &lds))) { delete ck; return rc; }
        rc = launch_resample(ctx, src, interp, EPI_STREAM_PCM, dtype, P, lds, in_bytes + out_elems * dtype_size(dtype), nullptr);
        if (rc) { delete ck; return rc; }
        done = true;
    }
    if (!done)
    if (dtype == AUKIT_F32 && src == SRC_PCM_S16LE_MONO && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {  // f32 tolerance path
        int frc = AUKIT_OK;
        done = fast_try(ctx, src, interp, d->sample_rate, 48000, segs, P, in_bytes + out_elems * 4, &frc, 1, P.lp_alpha);
        if (done && frc) { delete ck; return frc; }
    }
    if (!done && dtype == AUKIT_F32 && C == 2 && bd == 2 && d->data_type == AUKIT_SIGNED && !d->big_endian && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {
        // interleaved 16-bit stereo (nearly every WAV file): both channels, or averaged as they are read when `mono`
        bool aligned4 = (((uintptr_t)in->data()) & 3) == 0;  // frames must not straddle dwords
        for (uint32_t s = 0; s < in->n && aligned4; s++) aligned4 = (in->off[s] & 3) == 0;
        int frc = AUKIT_OK;
        if (aligned4) done = fast_try(ctx, SRC_PCM_S16LE_STEREO, interp, d->sample_rate, 48000, segs, P, in_bytes + out_elems * 4, &frc, mono ? 2 : 1, P.lp_alpha);
        if (done && frc) { delete ck; return frc; }
    }
    if (!done && dtype == AUKIT_F32 && C == 1 && bd == 1 && d->data_type != AUKIT_FLOAT && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC) && !getenv("AUKIT_NO_FAST_CONVERT")) {
        int frc = AUKIT_OK;  // 8-bit mono: the wave kernel reads the bytes themselves (fast_stream_u8.hip)
        done = fast_try(ctx, SRC_PCM8_MONO, interp, d->sample_rate, 48000, segs, P, in_bytes + out_elems * 4, &frc, 1, P.lp_alpha);
        if (done && frc) { delete ck; return frc; }
    }
    if (!done && dtype == AUKIT_F32 && !ctx->exact_math && C <= 2 && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {
        // every other interleaved format of one or two channels: unpack + resample + the stream.pcm epilogue in one launch (fast_fmt.hip, round 3)
        int frc = AUKIT_OK;
        const int *unused = nullptr;
        done = fast_fmt_try(ctx, d, interp, 48000, segs, P, in_bytes + out_elems * 4, &frc, &unused, (mono && C > 1) ? 2 : 1, P.lp_alpha);
        if (done && frc) { delete ck; return frc; }
    }
    if (!done && dtype == AUKIT_F32 && !ctx->exact_math && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC) && !getenv("AUKIT_NO_FAST_CONVERT")) {
        // every other PCM format (8-bit unsigned at 48 kHz is what most ComputerCraft audio is kept in), f32 tolerance path: the string is
        // unpacked to one f32 row per output channel (k_pcm_unpack; the `mono` mix is made there, in the reference's order) and the stream.pcm
        // wave kernel runs on the rows (fast_stream_f32.hip).  The reference-order kernel below moved 190 G samples/s on these.
        std::vector<UnpackRow> ur((size_t)in->n * nd);
        std::vector<uint64_t> roff((size_t)in->n * nd);
        uint64_t tot = 0, longest = 0;
        for (uint32_t s = 0; s < in->n; s++) {
            const uint64_t fr = (in->off[s + 1] - in->off[s]) / ((size_t)bd * C);
            for (int c = 0; c < nd; c++) {
                UnpackRow &u = ur[(size_t)s * nd + c];
                u.src = in->off[s]; u.dst = tot; u.frames = fr;
                u.first = mono ? 0 : (uint64_t)c; u.step = (uint64_t)C; u.mix = mono ? (uint64_t)C : 0;
                roff[(size_t)s * nd + c] = tot;
                tot += round_up(std::max<uint64_t>(fr, 1), 16) + 16;
            }
            longest = std::max(longest, fr);
        }
        std::vector<Seg> rsegs;
        rsegs.reserve(segs.size() * (size_t)nd);
        for (const Seg &g : segs)
            for (int c = 0; c < nd; c++) {
                Seg r = g;
                r.stream = g.stream * (unsigned)nd + (unsigned)c;
                r.out_off = g.out_off + (uint64_t)c * g.out_stride;
                r.out_stride = 0;
                rsegs.push_back(r);
            }
        if (longest && (rc = ctx->tmp_buf.ensure((size_t)tot * 4 + 256)) == AUKIT_OK && (rc = upload_table(ctx, ctx->tmp_buf2, ur.data(), ur.size() * sizeof(UnpackRow))) == AUKIT_OK &&
            (rc = upload_table(ctx, ctx->misc_buf, roff.data(), roff.size() * sizeof(uint64_t))) == AUKIT_OK) {
            const UnpackRow *d_ur = reinterpret_cast<const UnpackRow *>(ctx->tmp_buf2.p);
            const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((longest / 4 + 255) / 256 / 4, 64));
            if ((rc = ctx_begin_kernel(ctx))) { delete ck; return rc; }
            for (size_t first = 0; first < ur.size(); first += 65535) {
                const dim3 grid(gx, (unsigned)std::min<size_t>(65535, ur.size() - first));
                float *rows = reinterpret_cast<float *>(ctx->tmp_buf.p);
                if (bd == 1) hipLaunchKernelGGL((k_pcm_unpack<1, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, rows, d->data_type, d->big_endian ? 1 : 0);
                else if (bd == 2) hipLaunchKernelGGL((k_pcm_unpack<2, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, rows, d->data_type, d->big_endian ? 1 : 0);
                else if (bd == 3) hipLaunchKernelGGL((k_pcm_unpack<3, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, rows, d->data_type, d->big_endian ? 1 : 0);
                else hipLaunchKernelGGL((k_pcm_unpack<4, float>), grid, dim3(256), 0, ctx->stream, in->data(), d_ur + first, rows, d->data_type, d->big_endian ? 1 : 0);
            }
            if (hipGetLastError() != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "k_pcm_unpack launch failed"); }
            if ((rc = ctx_end_kernel(ctx, "k_pcm_unpack", in_bytes + tot * 4))) { delete ck; return rc; }
            ResampleParams R;
            memset(&R, 0, sizeof R);
            R.src = reinterpret_cast<const unsigned char *>(ctx->tmp_buf.p);
            R.src_off = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
            R.channels = 1;
            R.out = a->dev;
            R.lp_alpha = P.lp_alpha;
            R.safe_lo = R.src;
            R.safe_hi = R.src + ctx->tmp_buf.cap;
            int frc = AUKIT_OK;
            done = fast_try(ctx, SRC_AUDIO_F32, interp, d->sample_rate, 48000, rsegs, R, in_bytes + out_elems * 4, &frc, 1, P.lp_alpha);
            if (done && frc) { delete ck; return frc; }
        } else if (rc) { delete ck; return rc; }
    }
    if (!done && ctx->exact_math == 1 && dtype == AUKIT_F32 && src == SRC_PCM_S16LE_MONO && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {
        // AUKIT_OPT_EXACT_MATH = 1: fp64 arithmetic behind the f32 store, phase-weight table (wave_f64.hip with the stream.pcm epilogue)
        int wrc = AUKIT_OK;
        done = wave_f64_try(ctx, src, interp, d->sample_rate, 48000, segs, P, in_bytes + out_elems * 4, &wrc, 1, P.lp_alpha);
        if (done && wrc) { delete ck; return wrc; }
    }
    if (!done && src == SRC_PCM_S16LE_MONO && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {  // reference order, wave tiles
        int erc = AUKIT_OK;
        done = exact_wave_try(ctx, src, interp, d->sample_rate, 48000, segs, P, dtype, in_bytes + out_elems * dtype_size(dtype), &erc, 1);
        if (done && erc) { delete ck; return erc; }
    }
    if (!done) {
        size_t lds;
        if ((rc = plan_tiles(ctx, segs, cp_ratio, interp, nd, P, &lds))) { delete ck; return rc; }
        rc = launch_resample(ctx, src, interp, EPI_STREAM_PCM, dtype, P, lds, in_bytes + out_elems * dtype_size(dtype), nullptr);
        if (rc) { delete ck; return rc; }
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

// ---------------------------------------------------------------- stream.g711  aukit.lua:2850-2913
bool floor_wave_g711_try(aukit_ctx *ctx, int interp, double old_rate, const std::vector<Seg> &segs, ResampleParams &P, int dtype, uint64_t algorithmic_bytes, int *rc);

// stream.g711 with several channels (round 4): the channels of `str_byte(data, pos, ...)` (:2878-2893) are interleaved bytes; every
// channel is resampled on its own (:2897-2911), so C planar byte rows + the MONO three-tier floor kernel give the same chunks.  This pass writes the
// planar rows (1 byte per sample: a sixth of what the 8 kHz -> 48 kHz outputs weigh); row r = stream r / C, channel r % C, padded to 16 bytes.
__global__ __launch_bounds__(256) void k_deinterleave_bytes(const unsigned char *src, const unsigned long long *in_off, const unsigned long long *row_off, int C, unsigned char *dst) {
    const unsigned r = blockIdx.y, s = r / (unsigned)C, c = r - s * (unsigned)C;
    const unsigned long long frames = (in_off[s + 1] - in_off[s]) / (unsigned long long)C;
    const unsigned char *p = src + in_off[s] + c;
    unsigned char *o = dst + row_off[r];
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; 4 * g < frames; g += (unsigned long long)gridDim.x * 256) {
        unsigned w = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (4 * g + j < frames) w |= (unsigned)p[(4 * g + j) * (unsigned long long)C] << (8 * j);
        *reinterpret_cast<unsigned *>(o + 4 * g) = w;
    }
}

// two channels: a thread takes eight frames as one 16-byte load and leaves eight bytes in either row (the general kernel above reads every line of
// the input once per channel, a byte per load: 0.19 of stream.g711 stereo's 0.53 ms per 1024 five-second streams went there)
__global__ __launch_bounds__(256) void k_deinterleave_bytes2(const unsigned char *src, const unsigned long long *in_off, const unsigned long long *row_off, unsigned char *dst) {
    const unsigned s = blockIdx.y;
    const unsigned long long frames = (in_off[s + 1] - in_off[s]) / 2ull;
    const unsigned char *p = src + in_off[s];
    unsigned char *o0 = dst + row_off[2 * (size_t)s], *o1 = dst + row_off[2 * (size_t)s + 1];
    typedef unsigned u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; 8 * g < frames; g += (unsigned long long)gridDim.x * 256) {
        if (8 * g + 8 <= frames) {
            const u32x4u w = *reinterpret_cast<const u32x4u *>(p + 16 * g);
            uint2 a, b;
            a.x = __builtin_amdgcn_perm(w.y, w.x, 0x06040200u); a.y = __builtin_amdgcn_perm(w.w, w.z, 0x06040200u);
            b.x = __builtin_amdgcn_perm(w.y, w.x, 0x07050301u); b.y = __builtin_amdgcn_perm(w.w, w.z, 0x07050301u);
            *reinterpret_cast<uint2 *>(o0 + 8 * g) = a;    // (rows start at multiples of 16 bytes)
            *reinterpret_cast<uint2 *>(o1 + 8 * g) = b;
        } else {
            for (unsigned long long f = 8 * g; f < frames; f++) { o0[f] = p[2 * f]; o1[f] = p[2 * f + 1]; }
        }
    }
}

static int stream_g711(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out,
                       aukit_chunks **chunks_out) {
    const int C = d->channels;
    if (C < 1 || C > AUKIT_MAX_PLANAR_CHANNELS) return fail(AUKIT_E_ARG, "channels out of range");
    if (d->sample_rate != std::floor(d->sample_rate) || d->sample_rate < 1) return fail(AUKIT_E_UNSUPPORTED, "stream.g711 needs an integer sample rate");
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "invalid interpolation");
    if (dtype != AUKIT_I8 && dtype != AUKIT_F64) return fail(AUKIT_E_ARG, "stream.g711 output must be AUKIT_I8 or AUKIT_F64");
    const double ratio = 48000 / d->sample_rate;
    const uint64_t per_call = (uint64_t)d->sample_rate * (uint64_t)C;
    const int nd = mono ? 1 : C;
    aukit_chunks *ck = new aukit_chunks();
    ck->n = in->n;
    ck->nchunks.assign(in->n, 0);
    ck->status.assign(in->n, 0);
    ck->length_seconds.assign(in->n, 0);
    std::vector<uint64_t> lens(in->n, 0);
    std::vector<Seg> segs;
    uint64_t in_bytes = 0, out_elems = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        uint64_t nb = in->off[s + 1] - in->off[s];
        // A byte count that is not a multiple of the channel count (round 6: refused until now): the LAST call's table has one sample more in its
        // first channels than in the others, `newlen` follows the first (:2897), and the first output whose position reaches that sample reads
        // nil in the shorter channels — "attempt to perform arithmetic on a nil value" out of the iterator itself (nothing catches it, :2898-2909):
        // the calls before it deliver their chunks, that call raises.  (Above 48 kHz the last outputs stop short of the last sample: not modelled.)
        const bool ragged = nb % (uint64_t)C != 0;
        if (ragged && ratio < 1) { delete ck; return fail(AUKIT_E_UNSUPPORTED, "G.711 data length is not a multiple of the channel count at a rate above 48 kHz (stream %u)", s); }
        ck->length_seconds[s] = (double)(nb + ctx->sb_bytes) / d->sample_rate / C;
        in_bytes += nb;
        uint32_t calls = (uint32_t)((nb + per_call - 1) / per_call);  // calls that see data; the reference then returns {{}} forever (Q13)
        if (ragged) { calls--; ck->status[s] = AUKIT_E_LUA; }
        ck->nchunks[s] = calls;
        ck->max_chunks = std::max(ck->max_chunks, calls);
    }
    ck->lens.assign((size_t)ck->n * std::max<uint32_t>(ck->max_chunks, 1), 0);
    ck->pos.assign((size_t)ck->n * std::max<uint32_t>(ck->max_chunks, 1), 0);
    for (uint32_t s = 0; s < in->n; s++) {
        uint64_t nb = in->off[s + 1] - in->off[s];
        for (uint32_t k = 0; k < ck->nchunks[s]; k++) {
            uint64_t pos = (uint64_t)k * per_call;
            uint64_t cnt = std::min<uint64_t>(per_call, nb - pos);
            uint64_t m = cnt / C;                                       // #retval[1]
            uint32_t n_out = (uint32_t)std::floor((double)m * ratio);   // :2897
            Seg g;
            g.src_base = (long long)(pos / C) - 1;
            g.w_lo = 1;
            g.w_hi = (int)m;
            g.n_out = n_out;
            g.stream = s;
            g.out_off = lens[s];
            g.out_stride = 0;
            g.pad = 0;
            segs.push_back(g);
            ck->lens[(size_t)s * ck->max_chunks + k] = n_out;
            ck->pos[(size_t)s * ck->max_chunks + k] = ((double)(pos + ctx->sb_bytes + 1) - 1) / d->sample_rate / C;  // (lp - 1) / sampleRate / channels
            lens[s] += n_out;
        }
        out_elems += lens[s] * (uint64_t)nd;
    }
    int rc;
    aukit_audio *a = *out;
    if ((rc = audio_prepare(ctx, &a, in->n, nd, 48000, dtype, lens.data()))) { delete ck; return rc; }
    *out = a;
    for (Seg &g : segs) {
        g.out_off += a->row_off[g.stream];
        g.out_stride = (unsigned)a->row_stride[g.stream];
    }
    ResampleParams P;
    memset(&P, 0, sizeof P);
    fill_source(P, in, d);
    P.g711_scale = 1.0 / 64.0;  // m / 0x40  :2891
    P.mix_mono = mono ? 1 : 0;
    P.out = a->dev;
    bool done = false;
    if (C == 1 && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC)) {  // guarded short-cut under the floor(), bit-exact (floor_wave.hip)
        int frc = AUKIT_OK;
        done = floor_wave_g711_try(ctx, interp, d->sample_rate, segs, P, dtype, in_bytes + out_elems * dtype_size(dtype), &frc);
        if (done && frc) { delete ck; return frc; }
    }
    if (!done && C >= 2 && !mono && (interp == AUKIT_INTERP_LINEAR || interp == AUKIT_INTERP_CUBIC) && !ctx->exact_math && !getenv("AUKIT_G711_NO_PLANAR") && in->n) {
        // planar byte rows, then the mono floor kernel on n * C rows (k_deinterleave_bytes above)
        std::vector<uint64_t> roff((size_t)in->n * C);
        uint64_t tot = 0;
        for (uint32_t s = 0; s < in->n; s++) {
            const uint64_t fr = (in->off[s + 1] - in->off[s]) / (uint64_t)C;
            for (int c = 0; c < C; c++) { roff[(size_t)s * C + c] = tot; tot += (fr + 15) & ~15ull; }
        }
        const size_t tab_at = (size_t)((tot + 16 + 255) & ~255ull);
        if ((rc = ctx->tmp_buf2.ensure(tab_at + roff.size() * 8))) { delete ck; return rc; }
        unsigned char *pl = reinterpret_cast<unsigned char *>(ctx->tmp_buf2.p);
        if ((rc = h2d_table(ctx, pl + tab_at, roff.data(), roff.size() * 8))) { delete ck; return rc; }
        uint64_t maxfr = 0;
        for (uint32_t s = 0; s < in->n; s++) maxfr = std::max<uint64_t>(maxfr, (in->off[s + 1] - in->off[s]) / (uint64_t)C);
        const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>((maxfr / 4 + 255) / 256, 1), 64);
        if (C == 2)
            hipLaunchKernelGGL(k_deinterleave_bytes2, dim3((unsigned)std::min<uint64_t>(std::max<uint64_t>((maxfr / 8 + 255) / 256, 1), 64), in->n), dim3(256), 0, ctx->stream, in->data(),
                               reinterpret_cast<const unsigned long long *>(in->d_off), reinterpret_cast<const unsigned long long *>(pl + tab_at), pl);
        else
        hipLaunchKernelGGL(k_deinterleave_bytes, dim3(gx, in->n * (unsigned)C), dim3(256), 0, ctx->stream, in->data(), reinterpret_cast<const unsigned long long *>(in->d_off),
                           reinterpret_cast<const unsigned long long *>(pl + tab_at), C, pl);
        if (hipGetLastError() != hipSuccess) { delete ck; return fail(AUKIT_E_HIP, "k_deinterleave_bytes launch failed"); }
        std::vector<Seg> ps;
        ps.reserve(segs.size() * C);
        for (const Seg &g : segs)
            for (int c = 0; c < C; c++) {
                Seg q = g;
                q.stream = g.stream * (unsigned)C + (unsigned)c;
                q.out_off = g.out_off + (unsigned long long)c * g.out_stride;
                q.out_stride = 0;
                ps.push_back(q);
            }
        ResampleParams Q = P;
        Q.src = pl;
        Q.src_off = reinterpret_cast<const unsigned long long *>(pl + tab_at);
        Q.safe_lo = pl;
        Q.safe_hi = pl + tot + 16;
        Q.channels = 1;
        int frc = AUKIT_OK;
        done = floor_wave_g711_try(ctx, interp, d->sample_rate, ps, Q, dtype, in_bytes + out_elems * dtype_size(dtype), &frc);
        if (done && frc) { delete ck; return frc; }
    }
    if (!done) {
        size_t lds;
        if ((rc = plan_tiles(ctx, segs, ratio, interp, C, P, &lds))) { delete ck; return rc; }
        rc = launch_resample(ctx, C == 1 ? SRC_G711_MONO : SRC_G711, interp, EPI_STREAM_FLOOR, dtype, P, lds, in_bytes + out_elems * dtype_size(dtype), nullptr);
        if (rc) { delete ck; return rc; }
    }
    if (chunks_out) { if (*chunks_out) aukit_chunks_free(*chunks_out); *chunks_out = ck; } else delete ck;
    return AUKIT_OK;
}

// Block codecs decode to compact integer rows (int16 predictors / int8 DFPWM samples) in a scratch buffer; this turns
// those rows into an Audio: v / (v < 0 and norm_neg or norm_pos), optionally resampled in the same pass
// (`loader(...):resample(new_rate, interp)`).  Row r = stream r / channels, channel r % channels.
// ---- integer rows → Audio without resampling (the loaders aukit.dfpwm / adpcm / qoa / flac: `s / (s < 0 and 2^(b-1) or 2^(b-1)-1)`, :1082 and
// siblings).  The general kernel (k_resample, interpolation "none") evaluates a position per sample and moved 2 TB/s; this is the same division
// with 16-byte stores, consecutive lanes on consecutive vectors.  int8 samples go through a 256-entry table of their (true) quotients.
struct ConvRow { unsigned long long src, dst, len; };
template <typename S, typename T>
__global__ __launch_bounds__(256) void k_convert_rows(const S *src, const ConvRow *rows, T *out, double norm_pos, double norm_neg) {
    __shared__ double lut[256];
    if (sizeof(S) == 1) {
        const int v = (int)threadIdx.x - 128;
        lut[threadIdx.x] = (double)v / (v < 0 ? norm_neg : norm_pos);
        __syncthreads();
    }
    const ConvRow r = rows[blockIdx.y];
    const S *p = src + r.src;
    T *o = out + r.dst;
    constexpr int PV = 16 / (int)sizeof(T);  // samples per 16-byte store: consecutive lanes write consecutive vectors (1 KiB per store instruction)
    typedef S svp __attribute__((ext_vector_type(PV), aligned(1)));
    typedef T tvp __attribute__((ext_vector_type(PV), aligned(16)));  // rows of an Audio start on 64-byte boundaries (audio_prepare)
    auto conv = [&](S v) -> T {
        if (sizeof(S) == 1) return (T)lut[(int)v + 128];
        return (T)((double)v / (v < 0 ? norm_neg : norm_pos));
    };
    const unsigned long long groups = r.len / PV;
#pragma unroll 4
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (unsigned long long)gridDim.x * 256) {
        const svp v = *reinterpret_cast<const svp *>(p + PV * g);
        tvp w;
#pragma unroll
        for (int e = 0; e < PV; e++) w[e] = conv(v[e]);
        *reinterpret_cast<tvp *>(o + PV * g) = w;
    }
    if (blockIdx.x == 0) for (unsigned long long i = groups * PV + threadIdx.x; i < r.len; i += 256) o[i] = conv(p[i]);
}

int audio_from_int_rows(aukit_ctx *ctx, int src_kind, const void *rows_dev, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len,
                        uint32_t n, int channels, double rate, double new_rate, int interp, bool do_resample, int dtype, double norm_pos,
                        double norm_neg, aukit_audio **out) {
    const double ratio = do_resample ? new_rate / rate : 1.0;
    if (!(ratio > 0)) return fail(AUKIT_E_ARG, "bad sample rate");
    // F32 pipelines (round 4): the resample of `loader(...):resample(r)` on the int16 / int8 rows of the IMA / MS-ADPCM / QOA / DFPWM loaders is
    // left OWED exactly as FLAC's is (flac_tail.hip) — a following effects.lowpass / highpass pays it inside its own pass (BASELINE config 3:
    // aukit.wav -> resample -> lowpass as decode + ONE launch), anything else materialises it first (audio_flush)
    if (do_resample && dtype == AUKIT_F32 && rows_dev == ctx->tmp_buf.p && (src_kind == SRC_I16 || src_kind == SRC_I8)) {
        int lrc = AUKIT_OK;
        if (lazy_resample_try(ctx, row_off, row_len, n, channels, rate, new_rate, interp, 1.0, out, &lrc, nullptr, src_kind, norm_pos, norm_neg)) return lrc;
    }
    std::vector<uint64_t> lens(n);
    uint64_t in_elems = 0, out_elems = 0;
    for (uint32_t s = 0; s < n; s++) {
        const uint64_t L = row_len[(size_t)s * channels];
        lens[s] = do_resample ? resample_count(L, ratio) : L;  // newlen uses #data[1]  :659
        if (L > 0x7FFFFFF0ull || lens[s] > 0xFFFFFFF0ull) return fail(AUKIT_E_UNSUPPORTED, "stream too long");
        for (int c = 0; c < channels; c++) {
            if (row_len[(size_t)s * channels + c] < L && lens[s]) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
            in_elems += row_len[(size_t)s * channels + c];
        }
        if (lens[s] && std::floor(host_pos(lens[s] - 1, ratio)) > (double)L) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        out_elems += lens[s] * (uint64_t)channels;
    }
    aukit_audio *a = *out;
    int rc;
    if ((rc = audio_prepare(ctx, &a, n, channels, do_resample ? new_rate : rate, dtype, lens.data()))) return rc;
    *out = a;
    if (!do_resample && (src_kind == SRC_I8 || src_kind == SRC_I16 || src_kind == SRC_I32) && !getenv("AUKIT_NO_FAST_CONVERT")) {
        std::vector<ConvRow> cr((size_t)n * channels);
        uint64_t longest = 0;
        for (uint32_t s = 0; s < n; s++)
            for (int c = 0; c < channels; c++) {
                const size_t r = (size_t)s * channels + c;
                cr[r] = ConvRow{row_off[r], a->row_off[s] + (uint64_t)c * a->row_stride[s], lens[s]};
                longest = std::max(longest, lens[s]);
            }
        if (cr.empty() || longest == 0) return AUKIT_OK;
        if ((rc = upload_table(ctx, ctx->misc_buf, cr.data(), cr.size() * sizeof(ConvRow)))) return rc;
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        const ConvRow *d_cr = reinterpret_cast<const ConvRow *>(ctx->misc_buf.p);
        const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((longest / 4 + 255) / 256 / 8, 64));
        for (size_t first = 0; first < cr.size(); first += 65535) {
            const dim3 grid(gx, (unsigned)std::min<size_t>(65535, cr.size() - first));
#define AUKIT_CONV(S)                                                                                                                                   \
            do {                                                                                                                                        \
                if (dtype == AUKIT_F64) hipLaunchKernelGGL((k_convert_rows<S, double>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const S *>(rows_dev), d_cr + first, reinterpret_cast<double *>(a->dev), norm_pos, norm_neg); \
                else hipLaunchKernelGGL((k_convert_rows<S, float>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const S *>(rows_dev), d_cr + first, reinterpret_cast<float *>(a->dev), norm_pos, norm_neg);              \
            } while (0)
            if (src_kind == SRC_I8) AUKIT_CONV(signed char);
            else if (src_kind == SRC_I16) AUKIT_CONV(short);
            else AUKIT_CONV(int);
#undef AUKIT_CONV
        }
        AUKIT_HIP_CHECK(hipGetLastError());
        return ctx_end_kernel(ctx, "k_convert_rows", in_elems * (src_kind == SRC_I16 ? 2 : src_kind == SRC_I32 ? 4 : 1) + out_elems * dtype_size(dtype));
    }
    std::vector<Seg> segs((size_t)n * channels);
    for (uint32_t s = 0; s < n; s++)
        for (int c = 0; c < channels; c++) {
            const size_t r = (size_t)s * channels + c;
            Seg &g = segs[r];
            g.src_base = -1;
            g.w_lo = 1;
            g.w_hi = (int)row_len[r];
            g.n_out = (unsigned)lens[s];
            g.stream = (unsigned)r;
            g.out_off = a->row_off[s] + (uint64_t)c * a->row_stride[s];
            g.out_stride = 0;
            g.pad = 0;
        }
    if ((rc = upload_table(ctx, ctx->misc_buf, row_off.data(), row_off.size() * sizeof(uint64_t)))) return rc;
    ResampleParams P;
    memset(&P, 0, sizeof P);
    P.src = reinterpret_cast<const unsigned char *>(rows_dev);
    P.src_off = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
    P.channels = 1;
    P.norm_pos = norm_pos;
    P.norm_neg = norm_neg;
    P.out = a->dev;
    const int ip = do_resample ? interp : AUKIT_INTERP_NONE;
    if ((src_kind == SRC_I16 || src_kind == SRC_I8) && dtype == AUKIT_F32 && do_resample && rows_dev == ctx->tmp_buf.p && !ctx->exact_math &&
        ((src_kind == SRC_I16 && norm_pos == 32767 && norm_neg == 32768) || (src_kind == SRC_I8 && norm_pos == 127 && norm_neg == 128))) {
        // int16 / int8 rows (the IMA, MS... , QOA, DFPWM loaders) with a resample behind them, F32 tolerance path: an int16 row IS a 16-bit
        // little-endian mono string with the same normalisation (s / 32768 | 32767), an int8 row an 8-bit signed one — the wave kernels of the
        // PCM path take them as they are (byte offsets instead of element offsets).  k_resample moved 120-200 G samples/s on these.
        const int bytes_per = src_kind == SRC_I16 ? 2 : 1;
        std::vector<uint64_t> boff(row_off.size());
        for (size_t r = 0; r < row_off.size(); r++) boff[r] = row_off[r] * (uint64_t)bytes_per;
        if ((rc = upload_table(ctx, ctx->misc_buf, boff.data(), boff.size() * sizeof(uint64_t)))) return rc;
        P.data_type = AUKIT_SIGNED;
        P.safe_lo = reinterpret_cast<const unsigned char *>(rows_dev);
        P.safe_hi = P.safe_lo + ctx->tmp_buf.cap;
        int frc = AUKIT_OK;
        if (fast_try(ctx, src_kind == SRC_I16 ? SRC_PCM_S16LE_MONO : SRC_PCM8_MONO, ip, rate, new_rate, segs, P, in_elems * bytes_per + out_elems * 4, &frc)) return frc;
        if ((rc = upload_table(ctx, ctx->misc_buf, row_off.data(), row_off.size() * sizeof(uint64_t)))) return rc;  // not taken: element offsets again
    }
    if (src_kind == SRC_I32 && dtype == AUKIT_F32 && do_resample && rows_dev == ctx->tmp_buf.p) {  // F32 pipelines: tolerance path
        P.safe_lo = reinterpret_cast<const unsigned char *>(rows_dev);
        P.safe_hi = P.safe_lo + ctx->tmp_buf.cap;
        int frc = AUKIT_OK;
        if (fast_try(ctx, SRC_I32, ip, rate, new_rate, segs, P, in_elems * 4 + out_elems * 4, &frc)) return frc;
    }
    size_t lds;
    if ((rc = plan_tiles(ctx, segs, ratio, ip, 1, P, &lds))) return rc;
    return launch_resample(ctx, src_kind, ip, EPI_AUDIO, dtype, P, lds, in_elems * (src_kind == SRC_I16 ? 2 : src_kind == SRC_I32 || src_kind == SRC_AUDIO_F32 ? 4 : src_kind == SRC_AUDIO_F64 ? 8 : 1) + out_elems * dtype_size(dtype), nullptr);
}

}  // namespace aukit

using namespace aukit;

extern "C" {

int aukit_decode(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *desc, int dtype, aukit_audio **out) {
    if (!ctx || !in || !desc || !out) return fail(AUKIT_E_ARG, "null argument");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "dtype must be AUKIT_F64 or AUKIT_F32");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    if (desc->codec == AUKIT_CODEC_PCM || desc->codec == AUKIT_CODEC_G711) return decode_resample_flat(ctx, in, desc, desc->sample_rate, AUKIT_INTERP_NONE, false, dtype, out);
    if (*out && ((*out)->lazy_rs || (*out)->lazy_rows.p)) lazy_drop(ctx, *out);   // the output's old rows (a deferred resample nobody paid) return to the scratch the decoder is about to use
    return decode_block_codec(ctx, in, desc, 0, 0, false, dtype, out);
}

int aukit_decode_resample(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *desc, double new_rate, int interp, int dtype,
                          aukit_audio **out) {
    if (!ctx || !in || !desc || !out) return fail(AUKIT_E_ARG, "null argument");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "dtype must be AUKIT_F64 or AUKIT_F32");
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "bad argument #2 (invalid interpolation type)");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    if (desc->codec == AUKIT_CODEC_PCM || desc->codec == AUKIT_CODEC_G711) return decode_resample_flat(ctx, in, desc, new_rate, interp, true, dtype, out);
    if (*out && ((*out)->lazy_rs || (*out)->lazy_rows.p)) lazy_drop(ctx, *out);
    return decode_block_codec(ctx, in, desc, new_rate, interp, true, dtype, out);
}

int aukit_resample(aukit_ctx *ctx, const aukit_audio *in, double new_rate, int interp, aukit_audio **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    if (interp < 0 || interp > 3) return fail(AUKIT_E_ARG, "bad argument #2 (invalid interpolation type)");
    if (in->dtype != AUKIT_F64 && in->dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "resample needs a float audio");
    if (*out == in) return fail(AUKIT_E_ARG, "resample cannot run in place");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    AUKIT_FLUSH(ctx, in);
    const double ratio = new_rate / in->rate;  // :658
    if (!(ratio > 0)) return fail(AUKIT_E_ARG, "bad sample rate");
    const int C = in->channels;
    std::vector<uint64_t> lens(in->n);
    uint64_t in_elems = 0, out_elems = 0;
    for (uint32_t s = 0; s < in->n; s++) {
        lens[s] = resample_count(in->len[s], ratio);
        if (in->len[s] > 0x7FFFFFF0ull || lens[s] > 0xFFFFFFF0ull) return fail(AUKIT_E_UNSUPPORTED, "stream too long");
        if (lens[s] && std::floor(host_pos(lens[s] - 1, ratio)) > (double)in->len[s]) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        in_elems += in->len[s] * (uint64_t)C;
        out_elems += lens[s] * (uint64_t)C;
    }
    aukit_audio *a = *out;
    int rc;
    if ((rc = audio_prepare(ctx, &a, in->n, C, new_rate, in->dtype, lens.data()))) return rc;
    *out = a;
    std::vector<Seg> segs((size_t)in->n * C);
    std::vector<uint64_t> rows((size_t)in->n * C);
    for (uint32_t s = 0; s < in->n; s++)
        for (int c = 0; c < C; c++) {
            size_t r = (size_t)s * C + c;
            rows[r] = in->row_off[s] + (uint64_t)c * in->row_stride[s];
            Seg &g = segs[r];
            g.src_base = -1;
            g.w_lo = 1;
            g.w_hi = (int)in->len[s];
            g.n_out = (unsigned)lens[s];
            g.stream = (unsigned)r;
            g.out_off = a->row_off[s] + (uint64_t)c * a->row_stride[s];
            g.out_stride = 0;
            g.pad = 0;
        }
    if ((rc = upload_table(ctx, ctx->misc_buf, rows.data(), rows.size() * sizeof(uint64_t)))) return rc;
    ResampleParams P;
    memset(&P, 0, sizeof P);
    P.src = reinterpret_cast<const unsigned char *>(in->dev);
    P.src_off = reinterpret_cast<const unsigned long long *>(ctx->misc_buf.p);
    P.channels = 1;
    P.out = a->dev;
    P.safe_lo = reinterpret_cast<const unsigned char *>(in->dev);
    P.safe_hi = P.safe_lo + in->cap_bytes;
    if (in->dtype == AUKIT_F32) {
        int frc = AUKIT_OK;
        if (fast_try(ctx, SRC_AUDIO_F32, interp, in->rate, new_rate, segs, P, (in_elems + out_elems) * 4, &frc)) return frc;
    }
    if (in->dtype == AUKIT_F64) {  // reference-order fp64 on wave tiles (exact_wave.hip)
        int erc = AUKIT_OK;
        if (exact_wave_try(ctx, SRC_AUDIO_F64, interp, in->rate, new_rate, segs, P, AUKIT_F64, (in_elems + out_elems) * 8, &erc)) return erc;
    }
    size_t lds;
    if ((rc = plan_tiles(ctx, segs, ratio, interp, 1, P, &lds))) return rc;
    return launch_resample(ctx, in->dtype == AUKIT_F64 ? SRC_AUDIO_F64 : SRC_AUDIO_F32, interp, EPI_AUDIO, in->dtype, P, lds,
                           (in_elems + out_elems) * dtype_size(in->dtype), nullptr);
}

// aukit.stream.pcm(data, ...) with `data` a TABLE of numbers (aukit.lua:2255-2290): `n` tables as one host array of doubles + element offsets
int aukit_stream_decode_table(aukit_ctx *ctx, const double *values, const uint64_t *offsets, uint32_t n, const aukit_codec_desc *desc, int interp, int mono, int dtype,
                              aukit_audio **out, aukit_chunks **chunks) {
    if (!ctx || !desc || !out || (n && (!offsets || (offsets[n] && !values)))) return fail(AUKIT_E_ARG, "null argument");
    if (desc->codec != AUKIT_CODEC_PCM) return fail(AUKIT_E_UNSUPPORTED, "table input: aukit.stream.pcm only");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<uint64_t> boff((size_t)n + 1);
    for (uint32_t s = 0; s <= n; s++) {
        if (s && offsets[s] < offsets[s - 1]) return fail(AUKIT_E_ARG, "offsets must not decrease");
        boff[s] = (offsets[s] - offsets[0]) * 8;
    }
    aukit_batch *b = nullptr;
    static const double none = 0;
    int rc = aukit_batch_upload(ctx, &b, reinterpret_cast<const uint8_t *>(n && offsets[n] ? values + offsets[0] : &none), boff.data(), n);
    if (rc) return rc;
    rc = stream_pcm(ctx, b, desc, interp, mono, dtype, out, chunks, true);
    if (!rc) rc = aukit_ctx_sync(ctx);   // the uploaded copy goes away with this call
    aukit_batch_free(b);
    return rc;
}

int aukit_stream_decode(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *desc, int interp, int mono, int dtype,
                        aukit_audio **out, aukit_chunks **chunks) {
    if (!ctx || !in || !desc || !out) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    if (desc->codec == AUKIT_CODEC_PCM) return stream_pcm(ctx, in, desc, interp, mono, dtype, out, chunks);
    if (desc->codec == AUKIT_CODEC_G711) return stream_g711(ctx, in, desc, interp, mono, dtype, out, chunks);
    return stream_block_codec(ctx, in, desc, interp, mono, dtype, out, chunks);
}

}  // extern "C"
