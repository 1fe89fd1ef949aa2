// fast.hip — HBM-bound resampler for AUKIT_F32 storage (tolerance path, SURVEY.md §8d: ≤ 1e-6 RMS).
//
// Same segment/tile machinery as resample.hip, different arithmetic:
//   * position: x - 1 = o·a/b exactly (a/b = old_rate/new_rate reduced), so floor and fraction come from
//     32-bit integer arithmetic (magic-number division by b) instead of an fp64 divide + floor;
//     the reference's double x differs from the rational value by a few ulps at most (≤ 1e-10 in the output);
//   * taps are staged in LDS as f32 (half the LDS bytes of the exact path) with the reference's nil
//     fall-backs (`p0 or p1`, `p2 or p1`, `p3 or p2 or p1`, aukit.lua:264) materialised as replicated
//     edge samples, so the inner loop has no selects;
//   * the cubic (aukit.lua:265) is evaluated in f32 with FMAs in Horner form;
//   * stores: each wave transposes 4 rows of 64 results through LDS and issues one 16-byte store per lane.
// Per output: ≈ 36 full-rate VALU instructions vs ≈ 41 half-rate fp64 ones in the exact kernel, which
// moves the kernel from fp64-issue-bound to HBM-bound (algorithmic bytes = input bytes + 4 B per output).
#include <algorithm>
#include "resample.h"

namespace aukit {

AUKIT_DEV float g711_f32(unsigned byte, int ulaw, float scale) {
    unsigned b = byte ^ (ulaw ? 0xFFu : 0x55u);
    int m = b & 15, e = (b >> 4) & 7;
    if (!ulaw && e == 0) m = m * 4 + 2;
    else m = (m * 2 + 33) << e;
    if (ulaw) m -= 33;
    bool neg = ((b & 0x80) != 0) == (ulaw != 0);
    return (float)(neg ? -m : m) * scale;  // exact: |m| < 2^13, scale a power of two
}

template <int SRC>
AUKIT_DEV float fast_sample(const ResampleParams &P, const FastParams &F, const Seg &sg, long long g) {
    if constexpr (SRC == SRC_PCM_S16LE_MONO) {
        const unsigned char *q = P.src + P.src_off[sg.stream] + 2 * g;
        short s = (short)(q[0] | q[1] << 8);
        return (float)s * (s < 0 ? F.scale_neg : F.scale_pos);
    } else if constexpr (SRC == SRC_G711_MONO) {
        return g711_f32(P.src[P.src_off[sg.stream] + g], P.ulaw, (float)P.g711_scale);
    } else {
        return reinterpret_cast<const float *>(P.src)[P.src_off[sg.stream] + g];
    }
}

// stage table indices k_lo .. k_lo+n_stage-1 (clamped to [w_lo, w_hi]: edge replication) as f32; returns the
// LDS index of table index k_lo
template <int SRC>
AUKIT_DEV int fast_stage(const ResampleParams &P, const FastParams &F, const Seg &sg, int k_lo, int n_stage, float *sm) {
    const int tid = threadIdx.x;
    const bool edges = k_lo < sg.w_lo || k_lo + n_stage - 1 > sg.w_hi;
    float e_lo = 0.f, e_hi = 0.f;
    if (edges) {  // block-uniform
        e_lo = fast_sample<SRC>(P, F, sg, sg.src_base + sg.w_lo);
        e_hi = fast_sample<SRC>(P, F, sg, sg.src_base + sg.w_hi);
    }
    const long long g0 = sg.src_base + k_lo;
    if constexpr (SRC == SRC_PCM_S16LE_MONO) {
        const unsigned char *a0 = P.src + P.src_off[sg.stream] + 2 * g0;
        const unsigned char *al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
        const int head = (int)(a0 - al) >> 1;
        const int nvec = (head + n_stage + 7) >> 3;
        for (int v = tid; v < nvec; v += 256) {
            const unsigned char *p = al + 16 * (size_t)v;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (p >= P.safe_lo && p + 16 <= P.safe_hi) u = *reinterpret_cast<const uint4 *>(p);
            else {
                unsigned w[4] = {0, 0, 0, 0};
                for (int e = 0; e < 16; e++) { const unsigned char *q = p + e; if (q >= P.safe_lo && q < P.safe_hi) w[e >> 2] |= (unsigned)*q << (8 * (e & 3)); }
                u = make_uint4(w[0], w[1], w[2], w[3]);
            }
            short s[8] = {(short)(u.x & 0xFFFF), (short)(u.x >> 16), (short)(u.y & 0xFFFF), (short)(u.y >> 16),
                          (short)(u.z & 0xFFFF), (short)(u.z >> 16), (short)(u.w & 0xFFFF), (short)(u.w >> 16)};
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; e++) d[e] = (float)s[e] * (s[e] < 0 ? F.scale_neg : F.scale_pos);
            if (edges) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    int w = k_lo + 8 * v + e - head;
                    d[e] = w < sg.w_lo ? e_lo : (w > sg.w_hi ? e_hi : d[e]);
                }
            }
            float4 *o = reinterpret_cast<float4 *>(sm + 8 * v);
            o[0] = make_float4(d[0], d[1], d[2], d[3]);
            o[1] = make_float4(d[4], d[5], d[6], d[7]);
        }
        return head;
    } else if constexpr (SRC == SRC_G711_MONO) {
        const unsigned char *a0 = P.src + P.src_off[sg.stream] + g0;
        const unsigned char *al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
        const int head = (int)(a0 - al);
        const int nvec = (head + n_stage + 15) >> 4;
        const float sc = (float)P.g711_scale;
        for (int v = tid; v < nvec; v += 256) {
            const unsigned char *p = al + 16 * (size_t)v;
            unsigned w[4] = {0, 0, 0, 0};
            if (p >= P.safe_lo && p + 16 <= P.safe_hi) { uint4 u = *reinterpret_cast<const uint4 *>(p); w[0] = u.x; w[1] = u.y; w[2] = u.z; w[3] = u.w; }
            else for (int e = 0; e < 16; e++) { const unsigned char *q = p + e; if (q >= P.safe_lo && q < P.safe_hi) w[e >> 2] |= (unsigned)*q << (8 * (e & 3)); }
            float4 *o = reinterpret_cast<float4 *>(sm + 16 * v);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float d[4] = {g711_f32(w[e] & 0xFF, P.ulaw, sc), g711_f32((w[e] >> 8) & 0xFF, P.ulaw, sc), g711_f32((w[e] >> 16) & 0xFF, P.ulaw, sc), g711_f32(w[e] >> 24, P.ulaw, sc)};
                if (edges) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int wi = k_lo + 16 * v + 4 * e + k - head;
                        d[k] = wi < sg.w_lo ? e_lo : (wi > sg.w_hi ? e_hi : d[k]);
                    }
                }
                o[e] = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
        return head;
    } else {  // SRC_AUDIO_F32
        const float *a0 = reinterpret_cast<const float *>(P.src) + P.src_off[sg.stream] + g0;
        const float *al = (const float *)((uintptr_t)a0 & ~(uintptr_t)15);
        const int head = (int)(a0 - al);
        const int nvec = (head + n_stage + 3) >> 2;
        const float *row_lo = reinterpret_cast<const float *>(P.src) + P.src_off[sg.stream] + sg.src_base + sg.w_lo;
        const float *row_hi = reinterpret_cast<const float *>(P.src) + P.src_off[sg.stream] + sg.src_base + sg.w_hi;
        for (int v = tid; v < nvec; v += 256) {
            const float *p = al + 4 * (size_t)v;
            float d[4];
            if (p >= row_lo && p + 3 <= row_hi) { float4 u = *reinterpret_cast<const float4 *>(p); d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = u.w; }
            else for (int e = 0; e < 4; e++) { const float *q = p + e; d[e] = q < row_lo ? *row_lo : (q > row_hi ? *row_hi : *q); }
            *reinterpret_cast<float4 *>(sm + 4 * v) = make_float4(d[0], d[1], d[2], d[3]);
        }
        return head;
    }
}

template <int SRC, int INTERP, bool X4>
__global__ __launch_bounds__(256) void k_fast_resample(const ResampleParams P, const FastParams F) {
    extern __shared__ float smf[];
    float *const so = smf + F.cap;  // X4: 4 waves x 256 floats of output staging
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int HL = INTERP == AUKIT_INTERP_CUBIC ? 1 : 0, HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    float *const out = reinterpret_cast<float *>(P.out);
    const int rows_per_wave = F.tile_out >> 8;

    for (unsigned t = blockIdx.x; t < P.n_tiles; t += gridDim.x) {
        unsigned sidx, tin;
        if (P.tiles_per_seg) { sidx = t / P.tiles_per_seg; tin = t - sidx * P.tiles_per_seg; }
        else { sidx = P.tile_seg[t]; tin = t - P.seg_tile0[sidx]; }
        const Seg sg = P.segs[sidx];
        const unsigned o0 = tin * (unsigned)F.tile_out;
        if (o0 >= sg.n_out) continue;
        const unsigned cnt = min((unsigned)F.tile_out, sg.n_out - o0);
        // exact rational position of the tile's first output: (o0 * a) = kb * b + r0
        const unsigned long long prod = (unsigned long long)o0 * F.a;
        const unsigned kb = (unsigned)(prod / F.b);
        const unsigned r0 = (unsigned)(prod - (unsigned long long)kb * F.b);
        const unsigned klast = (r0 + (cnt - 1) * F.a) / F.b;
        const int k_lo = 1 + (int)kb - HL;  // table index of the first staged element
        const int n_stage = (int)klast + 1 + HL + HR;

        __syncthreads();
        const int shift = fast_stage<SRC>(P, F, sg, k_lo, n_stage, smf);
        __syncthreads();
        const float *tab = smf + shift + HL;  // tab[q] = d[1 + kb + q]

        float *orow = out + sg.out_off + o0;
        const unsigned wbase = (unsigned)(wave * rows_per_wave) * 64u;
        for (int r = 0; r < rows_per_wave; r++) {
            const unsigned rb = wbase + (unsigned)r * 64u;
            if (rb >= cnt) break;
            const unsigned j = rb + lane;
            const unsigned n = r0 + (j < cnt ? j : cnt - 1) * F.a;
            const unsigned q = __umulhi(n, F.magic);
            const unsigned rem = n - q * F.b;
            float fx = (float)rem * F.inv_b;
            fx = fmaf(fmaf(-fx, (float)F.b, (float)rem), F.inv_b, fx);  // one Newton step: fx = RN(rem / b) to ~1 ulp
            float p1 = tab[q], v;
            if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
                float p2 = tab[q + 1];
                v = fmaf(p2 - p1, fx, p1);
            } else {
                float p0 = tab[(int)q - 1], p2 = tab[q + 1], p3 = tab[q + 2];
                float c3 = fmaf(1.5f, p1 - p2, 0.5f * (p3 - p0));
                float c2 = fmaf(-0.5f, p3, fmaf(2.0f, p2, fmaf(-2.5f, p1, p0)));
                float c1 = 0.5f * (p2 - p0);
                v = fmaf(fmaf(fmaf(c3, fx, c2), fx, c1), fx, p1);
            }
            v = rem == 0 ? p1 : fminf(fmaxf(v, -1.0f), 1.0f);  // :667-668
            if constexpr (X4) {
                float *sw = so + wave * 256;
                sw[(r & 3) * 64 + lane] = v;
                if ((r & 3) == 3 || rb + 64 >= cnt) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const unsigned gb = rb - (unsigned)(r & 3) * 64u;  // first output of this group of ≤ 4 rows
                    const unsigned e0 = gb + 4u * lane;
                    float4 val = *reinterpret_cast<const float4 *>(sw + 4 * lane);
                    if (e0 + 3 < cnt) *reinterpret_cast<float4 *>(orow + e0) = val;
                    else {
                        if (e0 < cnt) orow[e0] = val.x;
                        if (e0 + 1 < cnt) orow[e0 + 1] = val.y;
                        if (e0 + 2 < cnt) orow[e0 + 2] = val.z;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                if (j < cnt) orow[j] = v;
            }
        }
    }
}

bool fast_eligible(int src_kind, int interp, double old_rate, double new_rate, FastParams &F) {
    if (src_kind != SRC_PCM_S16LE_MONO && src_kind != SRC_G711_MONO && src_kind != SRC_AUDIO_F32 && src_kind != SRC_I32 && src_kind != SRC_PCM_S16LE_STEREO && src_kind != SRC_PCM8_MONO) return false;
    if (interp != AUKIT_INTERP_LINEAR && interp != AUKIT_INTERP_CUBIC) return false;
    if (old_rate != std::floor(old_rate) || new_rate != std::floor(new_rate) || old_rate < 1 || new_rate < 1 || old_rate > 4e9 || new_rate > 4e9) return false;
    unsigned long long a = (unsigned long long)old_rate, b = (unsigned long long)new_rate, x = a, y = b;
    while (y) { unsigned long long t = x % y; x = y; y = t; }
    a /= x; b /= x;
    // b == 1 (equal rates — a 48 kHz source through stream.pcm, the usual case of austream — or an integer decimation): every position is an
    // integer; written as 2a / 2 the same machinery applies (rem is always 0: the interpolators return the tap itself, :666 / :2395) and the
    // division magic fits 32 bits
    if (b < 2) { a *= 2; b *= 2; }
    F.tile_out = 4096;
    if (const char *e = getenv("AUKIT_FAST_TILE")) { int v = atoi(e); if (v >= 1024 && v % 1024 == 0) F.tile_out = v; }  // tuning knob
    while (F.tile_out > 1024 && ((double)F.tile_out * (double)a / (double)b + 64) * 4 > 40 * 1024) F.tile_out -= 1024;
    if (((double)F.tile_out * (double)a / (double)b + 64) * 4 > 60 * 1024) return false;
    if (((double)b + (double)F.tile_out * (double)a) * (double)b >= 4294967296.0) return false;  // magic division exactness
    F.a = (unsigned)a;
    F.b = (unsigned)b;
    F.magic = (unsigned)((4294967296ull + b - 1) / b);
    F.inv_b = 1.0f / (float)b;
    F.cap = (((int)std::ceil((double)F.tile_out * (double)a / (double)b) + 64) + 3) & ~3;
    F.scale_pos = 1.0f / 32767.0f;
    F.scale_neg = 1.0f / 32768.0f;
    F.epi = 0;
    F.alpha = 0.f;
    F.dq64 = F.dr64 = 0;
    return true;
}

int plan_tiles_sized(aukit_ctx *ctx, const std::vector<Seg> &segs, int tile_out, ResampleParams &P);

template <int SRC, bool X4>
static int launch_fast_interp(aukit_ctx *ctx, int interp, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    if (interp == AUKIT_INTERP_LINEAR) hipLaunchKernelGGL((k_fast_resample<SRC, AUKIT_INTERP_LINEAR, X4>), dim3(grid), dim3(256), lds, ctx->stream, P, F);
    else hipLaunchKernelGGL((k_fast_resample<SRC, AUKIT_INTERP_CUBIC, X4>), dim3(grid), dim3(256), lds, ctx->stream, P, F);
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}

int launch_fast(aukit_ctx *ctx, int src_kind, int interp, const std::vector<Seg> &segs, ResampleParams &P, FastParams &F, uint64_t algorithmic_bytes) {
    int rc = plan_tiles_sized(ctx, segs, F.tile_out, P);
    if (rc) return rc;
    if (P.n_tiles == 0) return AUKIT_OK;
    const bool x4 = ctx->fast_store_x4;
    size_t lds = (size_t)F.cap * 4 + (x4 ? 4 * 256 * 4 : 0);
    unsigned per_cu = 16 * (unsigned)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));   // 16 x the resident count (8 / 32 / 128 per CU measured 1.53 / 1.44 / 1.41 ms on Audio:resample 44.1k linear)
    if (const char *e = getenv("AUKIT_FAST_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }  // tuning knob
    unsigned grid = std::min<unsigned>(P.n_tiles, (unsigned)ctx->num_cus * std::max(per_cu, 1u));
    if ((rc = ctx_begin_kernel(ctx))) return rc;
#define AUKIT_FAST_CASE(S)                                                                                   \
    case S: rc = x4 ? launch_fast_interp<S, true>(ctx, interp, P, F, lds, grid) : launch_fast_interp<S, false>(ctx, interp, P, F, lds, grid); break;
    switch (src_kind) {
        AUKIT_FAST_CASE(SRC_PCM_S16LE_MONO)
        AUKIT_FAST_CASE(SRC_G711_MONO)
        AUKIT_FAST_CASE(SRC_AUDIO_F32)
    default: rc = fail(AUKIT_E_ARG, "bad fast source");
    }
#undef AUKIT_FAST_CASE
    if (rc) return rc;
    static thread_local char nm[96];
    static const char *srcn[] = {"", "pcm_s16le_mono", "", "g711_mono", "", "audio_f32"};
    snprintf(nm, sizeof nm, "k_fast_resample<%s,%s,%s>", srcn[src_kind], interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", x4 ? "x4" : "x1");
    return ctx_end_kernel(ctx, nm, algorithmic_bytes);
}

// returns true when the fast kernel took the launch (*rc holds its status)
bool fast_try(aukit_ctx *ctx, int src_kind, int interp, double old_rate, double new_rate, const std::vector<Seg> &segs, ResampleParams &P,
              uint64_t algorithmic_bytes, int *rc, int epi, double alpha) {
    if (ctx->exact_math) return false;
    FastParams F;
    if (!fast_eligible(src_kind, interp, old_rate, new_rate, F)) return false;
    F.epi = epi;
    F.alpha = (float)alpha;
    for (const Seg &g : segs)
        if (g.w_hi < g.w_lo && g.n_out) return false;
    if (src_kind == SRC_PCM_S16LE_STEREO) {  // wave kernel only (fast_s16x2.hip)
        bool taken = false;
        int r2 = launch_fast_wave(ctx, src_kind, interp, segs, P, F, algorithmic_bytes, &taken);
        if (taken) *rc = r2;
        return taken;
    }
    if (epi) {  // epilogues other than Audio:resample exist as wave kernels only (fast_stream.hip)
        if (src_kind != SRC_PCM_S16LE_MONO && src_kind != SRC_AUDIO_F32 && src_kind != SRC_PCM8_MONO) return false;
        bool taken = false;
        int r2 = launch_fast_wave(ctx, src_kind, interp, segs, P, F, algorithmic_bytes, &taken);
        if (taken) *rc = r2;
        return taken;
    }
    if (src_kind == SRC_PCM8_MONO) {  // 8-bit mono strings: the wave kernel only
        bool taken = false;
        int r2 = launch_fast_wave(ctx, src_kind, interp, segs, P, F, algorithmic_bytes, &taken);
        if (taken) *rc = r2;
        return taken;
    }
    if (src_kind == SRC_I32) {  // integer rows: only the wave kernel, and only when v / norm is an exact f32 operation
        int e = 0;
        if (P.norm_pos != P.norm_neg || std::frexp(P.norm_pos, &e) != 0.5 || P.norm_pos > 16777216.0) return false;
        bool taken = false;
        int r2 = launch_fast_wave(ctx, src_kind, interp, segs, P, F, algorithmic_bytes, &taken);
        if (taken) *rc = r2;
        return taken;
    }
    if (!getenv("AUKIT_FAST_V1")) {  // wave-private pipelined kernel (fast2.hip); v1 kept for A/B and odd ratios
        bool taken = false;
        int r2 = launch_fast_wave(ctx, src_kind, interp, segs, P, F, algorithmic_bytes, &taken);
        if (taken) { *rc = r2; return true; }
    }
    *rc = launch_fast(ctx, src_kind, interp, segs, P, F, algorithmic_bytes);
    return true;
}

}  // namespace aukit
