// fast2.hip — wave-private, software-pipelined variant of the f32 tolerance resampler (see fast.hip for
// the arithmetic).  Round-1 profile of the workgroup-tiled kernel (profiles/r1a_*): HBM traffic equals the
// algorithmic bytes, VALU ≈ 26 % busy, but waves sit 58 % of their cycles in s_waitcnt / s_barrier because
// the load → LDS → compute phases of a workgroup are serialised by two block barriers per tile.  Here:
//   * every wave owns a 1024-output tile and a private LDS window: no __syncthreads at all;
//   * the 16-byte global loads of the NEXT tile are issued into registers before the current tile is
//     interpolated (issue-early / write-late), so HBM latency overlaps the VALU work of the same wave;
//   * per-tile index arithmetic is forced onto the scalar unit (readfirstlane), the 16 rows are unrolled.
#include <algorithm>
#include "resample.h"

namespace aukit {

#ifndef AUKIT_WT
#define AUKIT_WT 1024
#endif
constexpr int WT = AUKIT_WT;  // outputs per wave tile = 16 rows of 64

template <int SRC> struct SrcTraits;
template <> struct SrcTraits<SRC_PCM_S16LE_MONO> { static constexpr int BYTES = 2, SPV = 8; };
template <> struct SrcTraits<SRC_G711_MONO> { static constexpr int BYTES = 1, SPV = 16; };
template <> struct SrcTraits<SRC_AUDIO_F32> { static constexpr int BYTES = 4, SPV = 4; };
template <> struct SrcTraits<SRC_I32> { static constexpr int BYTES = 4, SPV = 4; };  // integer rows (FLAC): v * 2^-depth, exact in f32 for |v| < 2^24

AUKIT_DEV float g711_f32b(unsigned byte, int ulaw, float scale) {
    unsigned b = byte ^ (ulaw ? 0xFFu : 0x55u);
    int m = b & 15, e = (b >> 4) & 7;
    if (!ulaw && e == 0) m = m * 4 + 2;
    else m = (m * 2 + 33) << e;
    if (ulaw) m -= 33;
    bool neg = ((b & 0x80) != 0) == (ulaw != 0);
    return (float)(neg ? -m : m) * scale;
}

struct WaveTile {
    const unsigned char *al;   // 16-byte aligned address of the first vector
    float *orow;               // output of the tile's first sample
    unsigned cnt, r0;
    int k_lo, n_stage, head, nvec;
    int w_lo, w_hi;
    const unsigned char *base; // address of table index 0 (for edge replication)
};

template <int SRC, int HL, int HR>
AUKIT_DEV WaveTile describe(const ResampleParams &P, const FastParams &F, unsigned t) {
    using T = SrcTraits<SRC>;
    unsigned sidx, tin;
    if (P.tiles_per_seg) { sidx = t / P.tiles_per_seg; tin = t - sidx * P.tiles_per_seg; }
    else { sidx = P.tile_seg[t]; tin = t - P.seg_tile0[sidx]; }
    const Seg sg = P.segs[sidx];
    WaveTile w;
    const unsigned o0 = tin * (unsigned)WT;
    w.cnt = o0 < sg.n_out ? min((unsigned)WT, sg.n_out - o0) : 0u;
    const unsigned td = tin * F.wd;                    // (o0 * a) = (tin * wc + td / b) * b + td % b
    const unsigned tq = td / F.b;
    const unsigned kb = tin * F.wc + tq;
    w.r0 = td - tq * F.b;
    const unsigned klast = w.cnt ? (w.r0 + (w.cnt - 1) * F.a) / F.b : 0u;
    w.k_lo = 1 + (int)kb - HL;
    w.n_stage = (int)klast + 1 + HL + HR;
    w.w_lo = sg.w_lo;
    w.w_hi = sg.w_hi;
    if constexpr (SRC == SRC_AUDIO_F32 || SRC == SRC_I32) w.base = P.src + 4 * (size_t)P.src_off[sg.stream] + 4 * sg.src_base;
    else w.base = P.src + (size_t)P.src_off[sg.stream] + (long long)T::BYTES * sg.src_base;
    const unsigned char *a0 = w.base + (long long)T::BYTES * w.k_lo;
    w.al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
    w.head = (int)(a0 - w.al) / T::BYTES;
    w.nvec = (w.head + w.n_stage + T::SPV - 1) / T::SPV;
    w.orow = reinterpret_cast<float *>(P.out) + sg.out_off + o0;
    return w;
}

template <int NV>
AUKIT_DEV void issue_loads(const ResampleParams &P, const WaveTile &w, int lane, uint4 (&pre)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int v = lane + 64 * i;
        const unsigned char *p = w.al + 16 * (size_t)v;
        pre[i] = make_uint4(0, 0, 0, 0);
        if (v < w.nvec && p >= P.safe_lo && p + 16 <= P.safe_hi) pre[i] = *reinterpret_cast<const uint4 *>(p);
    }
}

template <int SRC>
AUKIT_DEV float sample_at(const ResampleParams &P, const FastParams &F, const unsigned char *q) {
    if constexpr (SRC == SRC_PCM_S16LE_MONO) {
        short s = (short)(q[0] | q[1] << 8);
        return (float)s * (s < 0 ? F.scale_neg : F.scale_pos);
    } else if constexpr (SRC == SRC_G711_MONO) {
        return g711_f32b(*q, P.ulaw, (float)P.g711_scale);
    } else if constexpr (SRC == SRC_I32) {
        return (float)*reinterpret_cast<const int *>(q) * F.scale_pos;
    } else {
        return *reinterpret_cast<const float *>(q);
    }
}

template <int SRC, int NV>
AUKIT_DEV void write_lds(const ResampleParams &P, const FastParams &F, const WaveTile &w, int lane, const uint4 (&pre)[NV], float *sm) {
    using T = SrcTraits<SRC>;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int v = lane + 64 * i;
        if (v >= w.nvec) continue;
        const uint4 u = pre[i];
        if constexpr (SRC == SRC_PCM_S16LE_MONO) {
            const unsigned ww[4] = {u.x, u.y, u.z, u.w};
            float d[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                short lo = (short)(ww[e] & 0xFFFF), hi = (short)(ww[e] >> 16);
                d[2 * e] = (float)lo * (lo < 0 ? F.scale_neg : F.scale_pos);
                d[2 * e + 1] = (float)hi * (hi < 0 ? F.scale_neg : F.scale_pos);
            }
            float4 *o = reinterpret_cast<float4 *>(sm + 8 * v);
            o[0] = make_float4(d[0], d[1], d[2], d[3]);
            o[1] = make_float4(d[4], d[5], d[6], d[7]);
        } else if constexpr (SRC == SRC_G711_MONO) {
            const unsigned ww[4] = {u.x, u.y, u.z, u.w};
            const float sc = (float)P.g711_scale;
            float4 *o = reinterpret_cast<float4 *>(sm + 16 * v);
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = make_float4(g711_f32b(ww[e] & 0xFF, P.ulaw, sc), g711_f32b((ww[e] >> 8) & 0xFF, P.ulaw, sc),
                                   g711_f32b((ww[e] >> 16) & 0xFF, P.ulaw, sc), g711_f32b(ww[e] >> 24, P.ulaw, sc));
        } else if constexpr (SRC == SRC_I32) {
            *reinterpret_cast<float4 *>(sm + 4 * v) = make_float4((float)(int)u.x * F.scale_pos, (float)(int)u.y * F.scale_pos, (float)(int)u.z * F.scale_pos, (float)(int)u.w * F.scale_pos);
        } else {
            *reinterpret_cast<float4 *>(sm + 4 * v) = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        }
    }
    // vectors that straddle the safe range were zero-filled: patch them sample by sample (first/last stream of a wrapped buffer)
    {
        const unsigned char *lo = w.al, *hi = w.al + 16 * (size_t)w.nvec;
        if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare
            for (int idx = lane; idx < w.nvec * T::SPV; idx += 64) {
                const unsigned char *q = w.al + (size_t)idx * T::BYTES;
                const unsigned char *vb = w.al + 16 * (size_t)(idx / T::SPV);
                if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) sm[idx] = (q >= P.safe_lo && q + T::BYTES <= P.safe_hi) ? sample_at<SRC>(P, F, q) : 0.f;
            }
        }
    }
    // nil fall-backs of interpolate.{linear,cubic} (aukit.lua:259, :264) = replicated edge samples
    const int k_hi = w.k_lo + w.n_stage - 1;
    if (w.k_lo < w.w_lo) {
        const float e_lo = sample_at<SRC>(P, F, w.base + (long long)T::BYTES * w.w_lo);
        for (int idx = lane; idx < w.w_lo - w.k_lo; idx += 64) sm[w.head + idx] = e_lo;
    }
    if (k_hi > w.w_hi) {
        const float e_hi = sample_at<SRC>(P, F, w.base + (long long)T::BYTES * w.w_hi);
        const int first = w.w_hi + 1 - w.k_lo;
        for (int idx = lane; idx < k_hi - w.w_hi; idx += 64) sm[w.head + first + idx] = e_hi;
    }
}

template <int INTERP>
AUKIT_DEV float interp_row(const FastParams &F, const float *tab, unsigned n) {
    const unsigned q = __umulhi(n, F.magic);
    const unsigned rem = n - q * F.b;
    float fx = (float)rem * F.inv_b;
    fx = fmaf(fmaf(-fx, (float)F.b, (float)rem), F.inv_b, fx);
    const float p1 = tab[q];
    float v;
    if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
        const float p2 = tab[q + 1];
        v = fmaf(p2 - p1, fx, p1);
    } else {
        const float p0 = tab[(int)q - 1], p2 = tab[q + 1], p3 = tab[q + 2];
        const float c3 = fmaf(1.5f, p1 - p2, 0.5f * (p3 - p0));
        const float c2 = fmaf(-0.5f, p3, fmaf(2.0f, p2, fmaf(-2.5f, p1, p0)));
        const float c1 = 0.5f * (p2 - p0);
        v = fmaf(fmaf(fmaf(c3, fx, c2), fx, c1), fx, p1);
    }
    return rem == 0 ? p1 : fminf(fmaxf(v, -1.0f), 1.0f);  // aukit.lua:667-668
}

template <int SRC, int INTERP, int NV>
__global__ __launch_bounds__(256) void k_fast_wave(const ResampleParams P, const FastParams F) {
    extern __shared__ float smf[];
    constexpr int HL = INTERP == AUKIT_INTERP_CUBIC ? 1 : 0, HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *const sm = smf + wave * (unsigned)F.cap;
    const unsigned nwaves = gridDim.x * 4u;
    const unsigned lane_a = (unsigned)lane * F.a;
    const unsigned row_a = 64u * F.a;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    uint4 pre[NV];
    WaveTile cur = describe<SRC, HL, HR>(P, F, t);
    issue_loads<NV>(P, cur, lane, pre);
    for (;;) {
        write_lds<SRC, NV>(P, F, cur, lane, pre, sm);
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe<SRC, HL, HR>(P, F, tn);
            issue_loads<NV>(P, nxt, lane, pre);  // in flight while this tile is interpolated
        }
        const float *tab = sm + cur.head + HL;  // tab[q] = d[1 + kb + q]
        float *orow = cur.orow;
        if (cur.cnt == (unsigned)WT && P.nt_store == 2) {  // experiment: 4 rows transposed through LDS → one 16-byte store per lane
            float *so = smf + 4u * (unsigned)F.cap + wave * 256u;
            unsigned n = cur.r0 + lane_a;
#pragma unroll
            for (int g = 0; g < WT / 256; g++) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) { so[rr * 64 + lane] = interp_row<INTERP>(F, tab, n); n += row_a; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float4 val = *reinterpret_cast<const float4 *>(so + 4 * lane);
                *reinterpret_cast<float4 *>(orow + g * 256 + 4 * lane) = val;
                __builtin_amdgcn_wave_barrier();
            }
        } else if (cur.cnt == (unsigned)WT) {
            unsigned n = cur.r0 + lane_a;
#pragma unroll
            for (int r = 0; r < WT / 64; r++) {
                const float v = interp_row<INTERP>(F, tab, n);
                if (P.nt_store) __builtin_nontemporal_store(v, &orow[r * 64 + lane]);  // outputs are never re-read by this kernel
                else orow[r * 64 + lane] = v;
                n += row_a;
            }
        } else {
            for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
                const unsigned j = rb + lane;
                const float v = interp_row<INTERP>(F, tab, cur.r0 + (j < cur.cnt ? j : cur.cnt - 1) * F.a);
                if (j < cur.cnt) orow[j] = v;
            }
        }
        if (!more) break;
        cur = nxt;
        t = tn;
    }
}

template <int SRC, int INTERP>
static int launch_nv(aukit_ctx *ctx, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    switch (nv) {
    case 1: hipLaunchKernelGGL((k_fast_wave<SRC, INTERP, 1>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 2: hipLaunchKernelGGL((k_fast_wave<SRC, INTERP, 2>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    case 4: hipLaunchKernelGGL((k_fast_wave<SRC, INTERP, 4>), dim3(grid), dim3(256), lds, ctx->stream, P, F); break;
    default: return fail(AUKIT_E_ARG, "bad NV");
    }
    AUKIT_HIP_CHECK(hipGetLastError());
    return AUKIT_OK;
}
template <int SRC>
static int launch_src2(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid) {
    if (interp == AUKIT_INTERP_LINEAR) return launch_nv<SRC, AUKIT_INTERP_LINEAR>(ctx, nv, P, F, lds, grid);
    return launch_nv<SRC, AUKIT_INTERP_CUBIC>(ctx, nv, P, F, lds, grid);
}

int launch_fast_wave(aukit_ctx *ctx, int src_kind, int interp, const std::vector<Seg> &segs, ResampleParams &P, FastParams &F,
                     uint64_t algorithmic_bytes, bool *taken) {
    *taken = false;
    const int spv = src_kind == SRC_PCM_S16LE_MONO ? 8 : (src_kind == SRC_G711_MONO ? 16 : 4);
    const int hl = interp == AUKIT_INTERP_CUBIC ? 1 : 0, hr = interp == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int win = (int)(((unsigned long long)(WT - 1) * F.a) / F.b) + 2 + hl + hr;  // staged samples per wave tile (upper bound)
    int nv = (win + 2 * spv + 64 * spv - 1) / (64 * spv);
    nv = nv <= 1 ? 1 : (nv <= 2 ? 2 : (nv <= 4 ? 4 : 0));
    if (!nv) return AUKIT_OK;  // strong down-sampling: v1 handles it
    uint64_t max_tiles = 0;
    for (const Seg &g : segs) max_tiles = std::max<uint64_t>(max_tiles, (g.n_out + WT - 1) / WT);
    F.wc = (unsigned)(((unsigned long long)WT * F.a) / F.b);
    F.wd = (unsigned)(((unsigned long long)WT * F.a) % F.b);
    if ((double)max_tiles * (double)F.wd >= 4294967296.0 || ((double)max_tiles + 1) * (double)F.wc >= 2147483648.0) return AUKIT_OK;
    if (((double)F.b + (double)WT * (double)F.a) * (double)F.b >= 4294967296.0) return AUKIT_OK;
    F.cap = ((nv * 64 * spv) + 15) & ~15;  // every lane may write a full vector
    int rc = plan_tiles_sized(ctx, segs, WT, P);
    if (rc) return rc;
    *taken = true;
    if (P.n_tiles == 0) return AUKIT_OK;
    if (const char *e = getenv("AUKIT_NT_STORE")) P.nt_store = atoi(e);
    if (src_kind == SRC_I32) F.scale_pos = F.scale_neg = (float)(1.0 / P.norm_pos);  // a power of two (checked by fast_try)
    size_t lds = (size_t)F.cap * 4 * 4 + (P.nt_store == 2 ? 4 * 256 * 4 : 0);  // 4 wave windows (+ 4 × 1 KiB of store-transpose staging in the x4 experiment)
    unsigned per_cu = 16;  // 2x the resident workgroups: measured +3.5 % over 8 (better tail balance across XCDs)
    if (const char *e = getenv("AUKIT_FAST_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }
    unsigned nblk_needed = (P.n_tiles + 3) / 4;
    unsigned grid = std::min<unsigned>(nblk_needed, (unsigned)ctx->num_cus * std::max(per_cu, 1u));
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    switch (src_kind) {
    case SRC_PCM_S16LE_MONO: rc = launch_src2<SRC_PCM_S16LE_MONO>(ctx, interp, nv, P, F, lds, grid); break;
    case SRC_G711_MONO: rc = launch_src2<SRC_G711_MONO>(ctx, interp, nv, P, F, lds, grid); break;
    case SRC_AUDIO_F32: rc = launch_src2<SRC_AUDIO_F32>(ctx, interp, nv, P, F, lds, grid); break;
    case SRC_I32: rc = launch_src2<SRC_I32>(ctx, interp, nv, P, F, lds, grid); break;
    default: rc = fail(AUKIT_E_ARG, "bad fast source");
    }
    if (rc) return rc;
    static thread_local char nm[96];
    static const char *srcn[] = {"", "pcm_s16le_mono", "", "g711_mono", "", "audio_f32", "", "", "i32"};
    snprintf(nm, sizeof nm, "k_fast_wave<%s,%s,nv%d>", srcn[src_kind], interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv);
    return ctx_end_kernel(ctx, nm, algorithmic_bytes);
}

}  // namespace aukit
