// resample.hip — fused stage → interpolate → epilogue kernels (see resample.h) for gfx950.
//
// Work decomposition: 256-thread workgroups walk tiles of `tile_out` consecutive outputs of one segment
// (grid-stride over tiles, ≈8 workgroups per CU).  Per tile: (1) the window of the source table the tile
// needs is decoded into LDS as fp64 with 16-byte coalesced global loads; (2) each wave takes consecutive
// rows of 64 outputs, lane L ↔ output base+L, so the four taps of neighbouring lanes fall in one 256-byte
// LDS bank row (conflict-free ds_read_b64); (3) the epilogue result is stored coalesced.
// The path is HBM-bound by design (no MFMA): algorithmic bytes are input bytes + output bytes.
#include "resample.h"
#include "resample_dev.h"

#define AUKIT_CONST_AS_I __attribute__((address_space(4)))
namespace aukit {

// ------------------------------------------------------------------ sample decoding
AUKIT_DEV double pcm_raw(const unsigned char *p, int bd, int dt, int be) {
    if (dt == AUKIT_FLOAT) {
        unsigned u = be ? ((unsigned)p[0] << 24 | (unsigned)p[1] << 16 | (unsigned)p[2] << 8 | p[3])
                        : ((unsigned)p[3] << 24 | (unsigned)p[2] << 16 | (unsigned)p[1] << 8 | p[0]);
        return (double)__uint_as_float(u);
    }
    unsigned long long u = 0;
    if (be) for (int i = 0; i < bd; i++) u = (u << 8) | p[i];
    else for (int i = bd - 1; i >= 0; i--) u = (u << 8) | p[i];
    if (dt == AUKIT_SIGNED) {
        unsigned long long sign = 1ull << (bd * 8 - 1);
        if (u & sign) return (double)((long long)u - (long long)(1ull << (bd * 8)));
    }
    return (double)u;
}
// aukit.lua:1133 / :1152 (Q4) / :1114
AUKIT_DEV double pcm_norm(double s, int dt, double maxv) {
    if (dt == AUKIT_SIGNED) return s / (s < 0 ? maxv : maxv - 1);
    if (dt == AUKIT_UNSIGNED) return (s - 128) / (s < 128 ? maxv : maxv - 1);
    return s;
}
// aukit.lua:1374-1379: returns ±m as a double (sign folded in), to be scaled by 2^-13 or 2^-6
AUKIT_DEV double g711_value(unsigned byte, int ulaw) {
    unsigned b = byte ^ (ulaw ? 0xFFu : 0x55u);
    int m = b & 15, e = (b >> 4) & 7;
    if (!ulaw && e == 0) m = m * 4 + 2;
    else m = (m * 2 + 33) << e;
    if (ulaw) m -= 33;
    bool neg = ((b & 0x80) != 0) == (ulaw != 0);
    return (double)(neg ? -m : m);  // m / -D == -(m / D) exactly, D a power of two
}

// ------------------------------------------------------------------ staging (returns LDS index of table index k_lo)
template <int SRC>
AUKIT_DEV int stage(const ResampleParams &P, const Seg &sg, int k_lo, int n_stage, double *sm) {
    const int tid = threadIdx.x;
    const long long g0 = sg.src_base + k_lo;  // source frame of table index k_lo
    if constexpr (SRC == SRC_PCM_S16LE_MONO) {
        const unsigned char *a0 = P.src + P.src_off[sg.stream] + 2 * g0;
        const unsigned char *al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
        const int head = (int)(a0 - al) >> 1;
        const int nvec = (head + n_stage + 7) >> 3;
        const double r32767 = 1.0 / 32767.0;
        for (int v = tid; v < nvec; v += 256) {
            const unsigned char *p = al + 16 * (size_t)v;
            short s[8];
            if (p >= P.safe_lo && p + 16 <= P.safe_hi) {
                uint4 u = *reinterpret_cast<const uint4 *>(p);
                s[0] = (short)(u.x & 0xFFFF); s[1] = (short)(u.x >> 16); s[2] = (short)(u.y & 0xFFFF); s[3] = (short)(u.y >> 16);
                s[4] = (short)(u.z & 0xFFFF); s[5] = (short)(u.z >> 16); s[6] = (short)(u.w & 0xFFFF); s[7] = (short)(u.w >> 16);
            } else {
                for (int e = 0; e < 8; e++) {
                    const unsigned char *q = p + 2 * e;
                    s[e] = (q >= P.safe_lo && q + 2 <= P.safe_hi) ? (short)(q[0] | q[1] << 8) : (short)0;
                }
            }
            double d[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                double x = (double)s[e];
                // s / (s < 0 and 32768 or 32767): exact scaling, or a correctly rounded quotient from RN(1/32767)
                d[e] = s[e] < 0 ? x * (1.0 / 32768.0) : div_rcp(x, 32767.0, r32767);
            }
            double2 *o = reinterpret_cast<double2 *>(sm + 8 * v);
            o[0] = make_double2(d[0], d[1]); o[1] = make_double2(d[2], d[3]); o[2] = make_double2(d[4], d[5]); o[3] = make_double2(d[6], d[7]);
        }
        return head;
    } else if constexpr (SRC == SRC_G711_MONO) {
        const unsigned char *a0 = P.src + P.src_off[sg.stream] + g0;
        const unsigned char *al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
        const int head = (int)(a0 - al);
        const int nvec = (head + n_stage + 15) >> 4;
        for (int v = tid; v < nvec; v += 256) {
            const unsigned char *p = al + 16 * (size_t)v;
            unsigned w[4];
            if (p >= P.safe_lo && p + 16 <= P.safe_hi) {
                uint4 u = *reinterpret_cast<const uint4 *>(p);
                w[0] = u.x; w[1] = u.y; w[2] = u.z; w[3] = u.w;
            } else {
                for (int e = 0; e < 4; e++) {
                    unsigned acc = 0;
                    for (int b = 0; b < 4; b++) { const unsigned char *q = p + 4 * e + b; if (q >= P.safe_lo && q < P.safe_hi) acc |= (unsigned)*q << (8 * b); }
                    w[e] = acc;
                }
            }
            double2 *o = reinterpret_cast<double2 *>(sm + 16 * v);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                double d0 = g711_value(w[e] & 0xFF, P.ulaw) * P.g711_scale, d1 = g711_value((w[e] >> 8) & 0xFF, P.ulaw) * P.g711_scale;
                double d2 = g711_value((w[e] >> 16) & 0xFF, P.ulaw) * P.g711_scale, d3 = g711_value(w[e] >> 24, P.ulaw) * P.g711_scale;
                o[2 * e] = make_double2(d0, d1);
                o[2 * e + 1] = make_double2(d2, d3);
            }
        }
        return head;
    } else if constexpr (SRC == SRC_PCM_GENERIC) {
        const int C = P.channels, SC = P.stage_channels, bd = P.table ? 8 : P.bit_depth >> 3;
        const unsigned char *base = P.src + P.src_off[sg.stream];
        // a table entry is the number itself (`local s = data[pos]`, :2263): no byte order, no width — only the normalisation applies
        auto raw_at = [&](size_t e) -> double { return P.table ? reinterpret_cast<const double *>(base)[e] : pcm_raw(base + e * bd, bd, P.data_type, P.big_endian); };
        const double maxv = (double)(1ull << (P.bit_depth - 1));
        const unsigned long long frames = P.planar ? P.src_frames[sg.stream] : 0;
        const int total = n_stage * SC;
        for (int idx = tid; idx < total; idx += 256) {
            int rel = idx / SC, c = idx - rel * SC;
            long long g = g0 + rel;
            double v;
            if (P.premix_mono) {  // self[i] = ((0 + read()) + read() ...) / channels   :2368
                double acc = 0;
                for (int cc = 0; cc < C; cc++) acc = acc + pcm_norm(raw_at((size_t)g * C + cc), P.data_type, maxv);
                v = acc / C;
            } else {
                size_t e = P.planar ? ((size_t)c * frames + (size_t)g) : ((size_t)g * C + c);
                v = pcm_norm(raw_at(e), P.data_type, maxv);
            }
            sm[c * P.cap + rel] = v;
        }
        return 0;
    } else if constexpr (SRC == SRC_G711) {
        const int C = P.channels, SC = P.stage_channels;
        const unsigned char *base = P.src + P.src_off[sg.stream];
        const int total = n_stage * SC;
        for (int idx = tid; idx < total; idx += 256) {
            int rel = idx / SC, c = idx - rel * SC;
            long long g = g0 + rel;
            sm[c * P.cap + rel] = g711_value(base[(size_t)g * C + c], P.ulaw) * P.g711_scale;
        }
        return 0;
    } else {  // planar rows: SRC_AUDIO_F64 / SRC_AUDIO_F32 / SRC_I16 / SRC_I8 / SRC_I32; channel c of a segment is row sg.stream * SC + c
        const int SC = P.stage_channels;
        for (int c = 0; c < SC; c++) {
            const unsigned long long ro = P.src_off[(size_t)sg.stream * SC + c];
            double *dst = sm + c * P.cap;
            if constexpr (SRC == SRC_AUDIO_F64) {
                const double *row = reinterpret_cast<const double *>(P.src) + ro;
                for (int rel = tid; rel < n_stage; rel += 256) dst[rel] = row[g0 + rel];
            } else if constexpr (SRC == SRC_AUDIO_F32) {
                const float *row = reinterpret_cast<const float *>(P.src) + ro;
                for (int rel = tid; rel < n_stage; rel += 256) dst[rel] = (double)row[g0 + rel];
            } else if constexpr (SRC == SRC_I16) {
                const short *row = reinterpret_cast<const short *>(P.src) + ro;
                for (int rel = tid; rel < n_stage; rel += 256) { double v = (double)row[g0 + rel]; dst[rel] = v / (v < 0 ? P.norm_neg : P.norm_pos); }
            } else if constexpr (SRC == SRC_I32) {
                const int *row = reinterpret_cast<const int *>(P.src) + ro;
                for (int rel = tid; rel < n_stage; rel += 256) { double v = (double)row[g0 + rel]; dst[rel] = v / (v < 0 ? P.norm_neg : P.norm_pos); }
            } else {
                const signed char *row = reinterpret_cast<const signed char *>(P.src) + ro;
                for (int rel = tid; rel < n_stage; rel += 256) { double v = (double)row[g0 + rel]; dst[rel] = v / (v < 0 ? P.norm_neg : P.norm_pos); }
            }
        }
        return 0;
    }
}

template <typename T> AUKIT_DEV void store_val(T *p, double v) { *p = (T)v; }
template <> AUKIT_DEV void store_val<signed char>(signed char *p, double v) { *p = (signed char)(int)v; }

// ------------------------------------------------------------------ the kernel
template <int SRC, int INTERP, int EPI, typename OUT_T>
__global__ __launch_bounds__(256) void k_resample(const ResampleParams P) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows_per_wave = P.tile_out >> 8;  // tile_out / 64 rows, 4 waves
    const int halo_l = P.halo_l, halo_r = P.halo_r;
    OUT_T *const out = reinterpret_cast<OUT_T *>(P.out);
    if (P.only_if && *(const AUKIT_CONST_AS_I int *)P.only_if == 0) return;  // a conditional redo (fast_fmt.hip) that is not needed

    for (unsigned t = blockIdx.x; t < P.n_tiles; t += gridDim.x) {
        unsigned sidx, tin;
        if (P.tiles_per_seg) { sidx = t / P.tiles_per_seg; tin = t - sidx * P.tiles_per_seg; }
        else { sidx = P.tile_seg[t]; tin = t - P.seg_tile0[sidx]; }
        const Seg sg = P.segs[sidx];
        const unsigned o0 = tin * (unsigned)P.tile_out;
        if (o0 >= sg.n_out) continue;  // uniform (block-wide) condition
        const unsigned cnt = min((unsigned)P.tile_out, sg.n_out - o0);

        // window of the table this tile touches
        const unsigned o_first = (EPI == EPI_STREAM_PCM && o0 > 0) ? o0 - 1 : o0;  // the FIR needs s(o0-1)
        int k_lo = (int)floor(pos_of(P, o_first)) - halo_l;
        int k_hi = (int)floor(pos_of(P, o0 + cnt - 1)) + halo_r;
        k_lo = max(k_lo, sg.w_lo);
        k_hi = min(k_hi, sg.w_hi);
        const int n_stage = k_hi - k_lo + 1;

        __syncthreads();  // previous tile's LDS reads are done
        const int shift = n_stage > 0 ? stage<SRC>(P, sg, k_lo, n_stage, sm) : 0;
        __syncthreads();

        const unsigned wbase = (unsigned)(wave * rows_per_wave) * 64u;
        if constexpr (EPI == EPI_STREAM_FLOOR) {
            const int SC = P.stage_channels;
            for (int r = 0; r < rows_per_wave; r++) {
                const unsigned j = wbase + (unsigned)r * 64u + lane;
                if (j >= cnt) break;
                const unsigned o = o0 + j;
                bool isint;
                if (P.mix_mono == 2) {  // stream.msadpcm stereo→mono: floor(l + r / 2)  :2672 (Q9)
                    const double l = eval_at<INTERP>(P, sg, sm + shift, k_lo, o, &isint), r2 = eval_at<INTERP>(P, sg, sm + P.cap + shift, k_lo, o, &isint);
                    store_val<OUT_T>(out + sg.out_off + o, lua_clamp(floor(l + r2 / 2), -128, 127));
                } else if (P.mix_mono) {
                    double acc = 0;
                    for (int c = 0; c < SC; c++) acc = acc + eval_at<INTERP>(P, sg, sm + c * P.cap + shift, k_lo, o, &isint);
                    store_val<OUT_T>(out + sg.out_off + o, lua_clamp(floor(acc / SC), -128, 127));  // :2908
                } else {
                    for (int c = 0; c < SC; c++) {
                        double s = eval_at<INTERP>(P, sg, sm + c * P.cap + shift, k_lo, o, &isint);
                        store_val<OUT_T>(out + sg.out_off + (size_t)c * sg.out_stride + o, lua_clamp(floor(s), -128, 127));  // :2909
                    }
                }
            }
        } else {
            for (int c = 0; c < P.stage_channels; c++) {
                const double *tab = sm + c * P.cap + shift;
                OUT_T *orow = out + sg.out_off + (size_t)c * sg.out_stride;
                double carry = 0;  // EPI_STREAM_PCM: ls, the RAW previous sample (Q2); 0 at the start of every chunk
                if constexpr (EPI == EPI_STREAM_PCM) {
                    bool ii;
                    if (wbase < cnt && o0 + wbase > 0) carry = eval_at<INTERP>(P, sg, tab, k_lo, o0 + wbase - 1, &ii);
                }
                for (int r = 0; r < rows_per_wave; r++) {
                    const unsigned rb = wbase + (unsigned)r * 64u;
                    if (rb >= cnt) break;  // wave-uniform
                    const unsigned j = rb + lane;
                    const bool active = j < cnt;
                    const unsigned o = o0 + (active ? j : cnt - 1);
                    bool isint;
                    double s = eval_at<INTERP>(P, sg, tab, k_lo, o, &isint);
                    if constexpr (EPI == EPI_AUDIO) {
                        if (active) store_val<OUT_T>(orow + o, isint ? s : lua_clamp(s, -1, 1));  // :667-668
                    } else if constexpr (EPI == EPI_STREAM_DFPWM) {  // stream.dfpwm :2481-2488: every channel gets the same sample (Q11)
                        if (active) {
                            const double v = isint ? s : lua_clamp(s, -128, 127);
                            if (P.mix_mono) {
                                double acc = 0;
                                for (int cc = 0; cc < P.out_channels; cc++) acc = acc + v;
                                store_val<OUT_T>(orow + o, acc / P.out_channels);
                            } else
                                for (int cc = 0; cc < P.out_channels; cc++) store_val<OUT_T>(orow + (size_t)cc * sg.out_stride + o, v);
                        }
                    } else {
                        double prev = __shfl_up(s, 1);
                        if (lane == 0) prev = carry;
                        carry = __shfl(s, 63);
                        double ns = prev + P.lp_alpha * (s - prev);                                      // :2401
                        if (active) store_val<OUT_T>(orow + o, lua_clamp(ns * (ns < 0 ? 128 : 127), -128, 127));  // :2402
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ launch
template <int SRC, int INTERP, int EPI, typename OUT_T>
static int launch_one(aukit_ctx *ctx, const ResampleParams &P, size_t lds, unsigned grid) {
    hipLaunchKernelGGL((k_resample<SRC, INTERP, EPI, OUT_T>), dim3(grid), dim3(256), lds, ctx->stream, P);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

template <int SRC, int EPI, typename OUT_T>
static int launch_interp(aukit_ctx *ctx, int interp, const ResampleParams &P, size_t lds, unsigned grid) {
    switch (interp) {
    case AUKIT_INTERP_NONE: return launch_one<SRC, AUKIT_INTERP_NONE, EPI, OUT_T>(ctx, P, lds, grid);
    case AUKIT_INTERP_LINEAR: return launch_one<SRC, AUKIT_INTERP_LINEAR, EPI, OUT_T>(ctx, P, lds, grid);
    case AUKIT_INTERP_CUBIC: return launch_one<SRC, AUKIT_INTERP_CUBIC, EPI, OUT_T>(ctx, P, lds, grid);
    case AUKIT_INTERP_SINC: return launch_one<SRC, AUKIT_INTERP_SINC, EPI, OUT_T>(ctx, P, lds, grid);
    }
    return fail(AUKIT_E_ARG, "bad argument #2 (invalid interpolation type)");
}

template <int SRC>
static int launch_src(aukit_ctx *ctx, int interp, int epi, int out_dtype, const ResampleParams &P, size_t lds, unsigned grid) {
    if (epi == EPI_AUDIO) {
        if (out_dtype == AUKIT_F64) return launch_interp<SRC, EPI_AUDIO, double>(ctx, interp, P, lds, grid);
        if (out_dtype == AUKIT_F32) return launch_interp<SRC, EPI_AUDIO, float>(ctx, interp, P, lds, grid);
    } else if (epi == EPI_STREAM_PCM) {
        if constexpr (SRC == SRC_PCM_GENERIC || SRC == SRC_PCM_S16LE_MONO) {
            if (out_dtype == AUKIT_F64) return launch_interp<SRC, EPI_STREAM_PCM, double>(ctx, interp, P, lds, grid);
            if (out_dtype == AUKIT_F32) return launch_interp<SRC, EPI_STREAM_PCM, float>(ctx, interp, P, lds, grid);
        }
    } else if (epi == EPI_STREAM_DFPWM) {
        if constexpr (SRC == SRC_I8) {
            if (out_dtype == AUKIT_F64) return launch_interp<SRC, EPI_STREAM_DFPWM, double>(ctx, interp, P, lds, grid);
            if (out_dtype == AUKIT_F32) return launch_interp<SRC, EPI_STREAM_DFPWM, float>(ctx, interp, P, lds, grid);
        }
    } else if (epi == EPI_STREAM_FLOOR) {
        if constexpr (SRC == SRC_G711 || SRC == SRC_G711_MONO || SRC == SRC_AUDIO_F64) {
            if (out_dtype == AUKIT_I8) return launch_interp<SRC, EPI_STREAM_FLOOR, signed char>(ctx, interp, P, lds, grid);
            if (out_dtype == AUKIT_F64) return launch_interp<SRC, EPI_STREAM_FLOOR, double>(ctx, interp, P, lds, grid);
        }
    }
    return fail(AUKIT_E_UNSUPPORTED, "no kernel for source %d / epilogue %d / dtype %d", SRC, epi, out_dtype);
}

static const char *kernel_name(int src, int interp, int epi) {
    static const char *srcn[] = {"pcm", "pcm_s16le_mono", "g711", "g711_mono", "audio_f64", "audio_f32", "i16", "i8", "i32"};
    static const char *intn[] = {"none", "linear", "cubic", "sinc"};
    static const char *epin[] = {"audio", "stream_pcm", "stream_floor", "stream_dfpwm"};
    static thread_local char buf[96];
    snprintf(buf, sizeof buf, "k_resample<%s,%s,%s>", srcn[src], intn[interp], epin[epi]);
    return buf;
}

int launch_resample(aukit_ctx *ctx, int src_kind, int interp, int epi, int out_dtype, const ResampleParams &P, size_t lds_bytes,
                    uint64_t algorithmic_bytes, const char **name) {
    if (P.n_tiles == 0) return AUKIT_OK;
    unsigned per_cu = (unsigned)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds_bytes, 1));
    if (per_cu < 1) per_cu = 1;
    per_cu *= 16;   // a finer hand-out than the resident count: 8 / 16 / 32 / 64 / 128 workgroups per CU measured 7.37 / 6.57 / 6.28 / 6.11 / 6.02 ms on stream.g711 through this kernel
    if (const char *e = getenv("AUKIT_RESAMPLE_PER_CU")) { const int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }   // tuning knob
    unsigned grid = std::min<unsigned>(P.n_tiles, (unsigned)ctx->num_cus * per_cu);
    int rc = ctx_begin_kernel(ctx);
    if (rc) return rc;
    switch (src_kind) {
    case SRC_PCM_GENERIC: rc = launch_src<SRC_PCM_GENERIC>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    case SRC_PCM_S16LE_MONO: rc = launch_src<SRC_PCM_S16LE_MONO>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    case SRC_G711: rc = launch_src<SRC_G711>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    case SRC_G711_MONO: rc = launch_src<SRC_G711_MONO>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    case SRC_AUDIO_F64: rc = launch_src<SRC_AUDIO_F64>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    case SRC_AUDIO_F32: rc = launch_src<SRC_AUDIO_F32>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    case SRC_I16: rc = launch_src<SRC_I16>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    case SRC_I8: rc = launch_src<SRC_I8>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    case SRC_I32: rc = launch_src<SRC_I32>(ctx, interp, epi, out_dtype, P, lds_bytes, grid); break;
    default: rc = fail(AUKIT_E_ARG, "bad source kind");
    }
    if (rc) return rc;
    const char *nm = kernel_name(src_kind, interp, epi);
    if (name) *name = nm;
    return ctx_end_kernel(ctx, nm, algorithmic_bytes);
}

// ------------------------------------------------------------------ host-side tiling
int plan_tiles(aukit_ctx *ctx, const std::vector<Seg> &segs, double ratio, int interp, int stage_channels, ResampleParams &P, size_t *lds_bytes) {
    int hl = 0, hr = 0;
    if (interp == AUKIT_INTERP_LINEAR) { hl = 0; hr = 1; }
    else if (interp == AUKIT_INTERP_CUBIC) { hl = 1; hr = 2; }
    else if (interp == AUKIT_INTERP_SINC) { hl = ctx->sinc_w; hr = ctx->sinc_w; }
    else if (interp != AUKIT_INTERP_NONE) return fail(AUKIT_E_ARG, "bad argument #2 (invalid interpolation type)");
    P.halo_l = hl; P.halo_r = hr; P.sinc_w = ctx->sinc_w;
    P.ratio = ratio; P.rcp = 1.0 / ratio;
    P.stage_channels = stage_channels;
    // tile size: the staged window (tile_out / ratio + halo) must fit the LDS budget
    const int slack = hl + hr + 2 + 32;  // +32: alignment head of the vector-load paths, FIR look-back
    const double eff_ratio = ratio / (double)(P.pos_mul > 1 ? P.pos_mul : 1);  // outputs advance pos_mul table steps at a time
    auto cap_for = [&](int to) { return (int)std::ceil((double)to / eff_ratio) + slack; };
    int tile_out = 2048;
    const size_t budget = 24 * 1024, hard = 64 * 1024;
    while (tile_out > 256 && (size_t)cap_for(tile_out) * 8 * stage_channels > budget) tile_out -= 256;
    if ((size_t)cap_for(tile_out) * 8 * stage_channels > hard)
        return fail(AUKIT_E_UNSUPPORTED, "resampling ratio %g with %d channels needs more than 64 KiB of LDS per tile", ratio, stage_channels);
    P.tile_out = tile_out;
    P.cap = (cap_for(tile_out) + 1) & ~1;
    *lds_bytes = (size_t)P.cap * 8 * stage_channels;
    uint64_t max_out = 0;
    for (const Seg &g : segs) max_out = std::max<uint64_t>(max_out, g.n_out);
    P.exact_rcp = exact_div_verified(ctx, ratio, (max_out + 1) * (uint64_t)(P.pos_mul > 0 ? P.pos_mul : 1)) ? 1 : 0;
    return plan_tiles_sized(ctx, segs, tile_out, P);
}

// tile tables for a given tile size: uploads segs (+ tile → segment tables for ragged batches)
int plan_tiles_sized(aukit_ctx *ctx, const std::vector<Seg> &segs, int tile_out, ResampleParams &P) {
    P.tile_out = tile_out;
    // The same call on the same batch again (a streaming server's steady state, the bench): the tables are still on the device — for
    // stream.pcm on 4096 ten-second streams they are 1.6 MB of segments and 7.7 MB of tile -> segment entries per call.
    const size_t seg_bytes = segs.size() * sizeof(Seg);
    if (!ctx->plan_key.empty() && ctx->plan_tile_out == tile_out && ctx->plan_segs.size() == seg_bytes && seg_bytes &&
        memcmp(ctx->plan_segs.data(), segs.data(), seg_bytes) == 0 && !getenv("AUKIT_NO_PLAN_CACHE")) {
        P.n_tiles = ctx->plan_n_tiles;
        P.segs = reinterpret_cast<const Seg *>(ctx->seg_buf.p);
        P.tiles_per_seg = ctx->plan_tiles_per_seg;
        if (P.tiles_per_seg) { P.tile_seg = nullptr; P.seg_tile0 = nullptr; }
        else { P.tile_seg = reinterpret_cast<const unsigned *>(ctx->tile_buf.p); P.seg_tile0 = P.tile_seg + P.n_tiles; }
        return AUKIT_OK;
    }
    uint64_t max_out = 0;
    std::vector<unsigned> tile0(segs.size() + 1, 0);
    bool uniform = true;
    unsigned tps0 = segs.empty() ? 0 : (segs[0].n_out + tile_out - 1) / tile_out;
    uint64_t nt = 0;
    for (size_t i = 0; i < segs.size(); i++) {
        unsigned tps = (segs[i].n_out + tile_out - 1) / tile_out;
        if (tps != tps0) uniform = false;
        tile0[i] = (unsigned)nt;
        nt += tps;
        if (segs[i].n_out > max_out) max_out = segs[i].n_out;
    }
    if (nt > 0xFFFFFFF0ull) return fail(AUKIT_E_UNSUPPORTED, "too many tiles");
    P.n_tiles = (unsigned)nt;
    int rc = upload_table(ctx, ctx->seg_buf, segs.data(), seg_bytes);
    if (rc) return rc;
    P.segs = reinterpret_cast<const Seg *>(ctx->seg_buf.p);
    if (uniform && tps0 > 0) {
        P.tiles_per_seg = tps0;
        P.tile_seg = nullptr;
        P.seg_tile0 = nullptr;
    } else {
        P.tiles_per_seg = 0;
        std::vector<unsigned> tab(nt + segs.size() + 1);
        for (size_t i = 0; i < segs.size(); i++) {
            unsigned tps = (segs[i].n_out + tile_out - 1) / tile_out;
            for (unsigned k = 0; k < tps; k++) tab[tile0[i] + k] = (unsigned)i;
        }
        for (size_t i = 0; i < segs.size(); i++) tab[nt + i] = tile0[i];
        rc = upload_table(ctx, ctx->tile_buf, tab.data(), tab.size() * sizeof(unsigned));
        if (rc) return rc;
        P.tile_seg = reinterpret_cast<const unsigned *>(ctx->tile_buf.p);
        P.seg_tile0 = P.tile_seg + nt;
    }
    ctx->plan_segs.assign(reinterpret_cast<const unsigned char *>(segs.data()), reinterpret_cast<const unsigned char *>(segs.data()) + seg_bytes);
    ctx->plan_tile_out = tile_out;
    ctx->plan_n_tiles = P.n_tiles;
    ctx->plan_tiles_per_seg = P.tiles_per_seg;
    ctx->plan_key = "tiles";
    return AUKIT_OK;
}

}  // namespace aukit
