// fast2.hip — wave-private, software-pipelined variant of the f32 tolerance resampler (see fast.hip for
// the arithmetic).  Round-1 profile of the workgroup-tiled kernel (profiles/r1a_*): HBM traffic equals the
// algorithmic bytes, VALU ≈ 26 % busy, but waves sit 58 % of their cycles in s_waitcnt / s_barrier because
// the load → LDS → compute phases of a workgroup are serialised by two block barriers per tile.  Here:
//   * every wave owns a 1024-output tile and a private LDS window: no __syncthreads at all;
//   * the 16-byte global loads of the NEXT tile are issued into registers before the current tile is
//     interpolated (issue-early / write-late), so HBM latency overlaps the VALU work of the same wave;
//   * per-tile index arithmetic is forced onto the scalar unit (readfirstlane), the 16 rows are unrolled.
#include <algorithm>
#include "fast_wave_dev.h"

namespace aukit {

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
            const unsigned n0 = cur.r0 + lane_a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
#pragma unroll
            for (int r = 0; r < WT / 64; r++) {
                const float v = interp_qr<SRC, INTERP>(F, tab, q, rem);
                if (P.nt_store) __builtin_nontemporal_store(v, &orow[r * 64 + lane]);  // outputs are never re-read by this kernel
                else orow[r * 64 + lane] = v;
                rem += F.dr64;  // the same lane, one row (64 outputs) further
                q += F.dq64;
                const bool wrap = rem >= F.b;
                rem -= wrap ? F.b : 0u;
                q += wrap ? 1u : 0u;
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
    const int spv = src_kind == SRC_PCM_S16LE_MONO ? 8 : (src_kind == SRC_G711_MONO || src_kind == SRC_PCM8_MONO ? 16 : 4);  // source elements (frames for stereo) per 16-byte vector
    const int hl = (interp == AUKIT_INTERP_CUBIC ? 1 : 0) + (F.epi ? 1 : 0), hr = interp == AUKIT_INTERP_CUBIC ? 2 : 1;  // stream.pcm: one more tap to the left
    const int win = (int)(((unsigned long long)(WT - 1) * F.a) / F.b) + 2 + hl + hr;  // staged samples per wave tile (upper bound)
    int nv = (win + 2 * spv + 64 * spv - 1) / (64 * spv);
    nv = nv <= 1 ? 1 : (nv <= 2 ? 2 : (nv <= 4 ? 4 : (nv <= 8 && (src_kind == SRC_PCM_S16LE_STEREO || (src_kind == SRC_AUDIO_F32 && F.epi == 1)) ? 8 : 0)));  // (8: interleaved stereo at equal rates)
    if (!nv) return AUKIT_OK;  // strong down-sampling: v1 handles it
    uint64_t max_tiles = 0;
    for (const Seg &g : segs) max_tiles = std::max<uint64_t>(max_tiles, (g.n_out + WT - 1) / WT);
    F.wc = (unsigned)(((unsigned long long)WT * F.a) / F.b);
    F.wd = (unsigned)(((unsigned long long)WT * F.a) % F.b);
    if ((double)max_tiles * (double)F.wd >= 4294967296.0 || ((double)max_tiles + 1) * (double)F.wc >= 2147483648.0) return AUKIT_OK;
    if (((double)F.b + (double)WT * (double)F.a) * (double)F.b >= 4294967296.0) return AUKIT_OK;
    F.cap = ((nv * 64 * spv) + 15) & ~15;  // every lane may write a full vector
    F.dq64 = (unsigned)((64ull * F.a) / F.b);
    F.dr64 = (unsigned)((64ull * F.a) % F.b);
    int rc = plan_tiles_sized(ctx, segs, WT, P);
    if (rc) return rc;
    *taken = true;
    if (P.n_tiles == 0) return AUKIT_OK;
    if (const char *e = getenv("AUKIT_NT_STORE")) P.nt_store = atoi(e);
    if (src_kind == SRC_I32) F.scale_pos = F.scale_neg = (float)(1.0 / P.norm_pos);  // a power of two (checked by fast_try)
    size_t lds = (size_t)F.cap * 4 * 4 + (P.nt_store == 2 ? 4 * 256 * 4 : 0);  // 4 wave windows (+ 4 × 1 KiB of store-transpose staging in the x4 experiment)
    // workgroups per CU in the grid (a finer hand-out of the tiles than the resident count): 16 / 32 / 64 / 128 measured 2.010 / 1.958 / 1.891 / 1.868 ms on
    // config T in f32 arithmetic, 1.605 / 1.601 / 1.579 / 1.508 on config 2a in f32, 2.195 / 2.154 / 2.125 / 2.112 on stream.pcm (f32); the interleaved-stereo
    // kernels are best at 16 (2.03 against 2.10-2.19)
    unsigned per_cu = src_kind == SRC_PCM_S16LE_STEREO ? 16 : 128;
    if (const char *e = getenv("AUKIT_FAST_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }
    unsigned nblk_needed = (P.n_tiles + 3) / 4;
    unsigned grid = std::min<unsigned>(nblk_needed, (unsigned)ctx->num_cus * std::max(per_cu, 1u));
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    if (F.epi && src_kind == SRC_PCM_S16LE_STEREO) {  // stream.pcm on interleaved stereo (fast_stream_s16x2.hip)
        if ((rc = launch_fast_wave_stream_s16x2(ctx, interp, nv, P, F, grid))) return rc;
        static thread_local char nmx[96];
        snprintf(nmx, sizeof nmx, "k_fast_wave_stream_s16x2<%s,nv%d,%s>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv, F.epi == 2 ? "mono" : "stereo");
        return ctx_end_kernel(ctx, nmx, algorithmic_bytes);
    }
    if (F.epi == 1 && src_kind == SRC_PCM8_MONO) {  // stream.pcm on 8-bit mono strings (fast_stream_u8.hip)
        if ((rc = launch_fast_wave_stream_u8(ctx, interp, nv, P, F, lds, grid))) return rc;
        static thread_local char nmu[96];
        snprintf(nmu, sizeof nmu, "k_fast_wave_stream<pcm8_mono,%s,nv%d,stream_pcm>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv);
        return ctx_end_kernel(ctx, nmu, algorithmic_bytes);
    }
    if (F.epi == 1 && src_kind == SRC_AUDIO_F32) {  // stream.pcm on unpacked f32 rows (fast_stream_f32.hip)
        if ((rc = launch_fast_wave_stream_f32(ctx, interp, nv, P, F, lds, grid))) return rc;
        static thread_local char nmf[96];
        snprintf(nmf, sizeof nmf, "k_fast_wave_stream<audio_f32,%s,nv%d,stream_pcm>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv);
        return ctx_end_kernel(ctx, nmf, algorithmic_bytes);
    }
    if (F.epi == 1) {
        if ((rc = launch_fast_wave_stream(ctx, interp, nv, P, F, lds, grid))) return rc;
        static thread_local char nms[96];
        snprintf(nms, sizeof nms, "k_fast_wave_stream<pcm_s16le_mono,%s,nv%d,stream_pcm>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv);
        return ctx_end_kernel(ctx, nms, algorithmic_bytes);
    }
    if (src_kind == SRC_PCM_S16LE_STEREO) {
        if ((rc = launch_fast_wave_s16x2(ctx, interp, nv, P, F, grid))) return rc;
        static thread_local char nm2[96];
        snprintf(nm2, sizeof nm2, "k_fast_wave_s16x2<%s,nv%d>", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv);
        return ctx_end_kernel(ctx, nm2, algorithmic_bytes);
    }
    if (F.b >= 2 * F.a && nv <= 2 && (src_kind == SRC_PCM_S16LE_MONO || src_kind == SRC_G711_MONO) && !getenv("AUKIT_FAST_NOCOEF")) {
        // up-sampling by 2x and more: per-source-sample coefficient table (fast_coef.hip)
        // (the write-dominated coefficient-table kernel wants a finer hand-out still: 128 / 256 / 512 / 1024 workgroups per CU measured 1.52 / 1.46 / 1.36-1.38 / 1.42 ms on config 2a in f32)
        const unsigned gridc = getenv("AUKIT_FAST_BLOCKS_PER_CU") ? grid : std::min<unsigned>(nblk_needed, (unsigned)ctx->num_cus * 512u);
        if ((rc = launch_fast_wave_coef(ctx, src_kind, interp, nv, win, P, F, gridc))) return rc;
        static thread_local char nmc[96];
        snprintf(nmc, sizeof nmc, "k_fast_wave_coef<%s,%s,nv%d>", src_kind == SRC_PCM_S16LE_MONO ? "pcm_s16le_mono" : "g711_mono", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv);
        return ctx_end_kernel(ctx, nmc, algorithmic_bytes);
    }
    switch (src_kind) {
    case SRC_PCM_S16LE_MONO: rc = launch_src2<SRC_PCM_S16LE_MONO>(ctx, interp, nv, P, F, lds, grid); break;
    case SRC_G711_MONO: rc = launch_src2<SRC_G711_MONO>(ctx, interp, nv, P, F, lds, grid); break;
    case SRC_AUDIO_F32: rc = launch_src2<SRC_AUDIO_F32>(ctx, interp, nv, P, F, lds, grid); break;
    case SRC_I32: rc = launch_src2<SRC_I32>(ctx, interp, nv, P, F, lds, grid); break;
    case SRC_PCM8_MONO: rc = launch_src2<SRC_PCM8_MONO>(ctx, interp, nv, P, F, lds, grid); break;
    default: rc = fail(AUKIT_E_ARG, "bad fast source");
    }
    if (rc) return rc;
    static thread_local char nm[96];
    static const char *srcn[] = {"", "pcm_s16le_mono", "", "g711_mono", "", "audio_f32", "", "", "i32", "", "pcm8_mono"};
    snprintf(nm, sizeof nm, "k_fast_wave<%s,%s,nv%d>", srcn[src_kind], interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv);
    return ctx_end_kernel(ctx, nm, algorithmic_bytes);
}

}  // namespace aukit
