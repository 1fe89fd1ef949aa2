// exact_wave.hip — Audio:resample in the reference's fp64 operation order on the wave-private tile engine of fast2.hip.
//
// AUKIT_F64 storage (the default of the Lua / Python mirrors) means bit-exact results: x = (i-1)/ratio + 1 as a rounded double,
// `x % 1 == 0` deciding between copy and interpolation, interpolate.linear / .cubic term by term with pow(), clamp (aukit.lua:
// 653-673, :257-266).  k_resample does that with workgroup tiles, two block barriers per tile and loads that are consumed as they
// arrive; this kernel runs the same per-output code (eval_at, resample_dev.h) on wave-private 1024-output tiles with the next
// tile's window in flight in registers while the current one is evaluated — no barrier, eight workgroups' worth of independent
// waves per CU.  Sources: 16-bit signed little-endian mono PCM (the fused decode + resample of aukit_decode_resample) and f64 audio
// rows (aukit_resample, what `audio:resample(48000)` is in the mirrors).  Bit-identical to k_resample (tests compare both with
// the oracle and with each other); anything it does not take (other formats, sinc, non-integer rates) stays with k_resample.
// With EPI = 1 the same tiles carry the stream.pcm epilogue (aukit.stream.pcm on s16 mono, F64 storage).
#include <algorithm>
#include "fast_wave_dev.h"
#include "resample_dev.h"

namespace aukit {

template <> struct SrcTraits<SRC_AUDIO_F64> { static constexpr int BYTES = 8, SPV = 2; };

// describe() of fast_wave_dev.h with the row addressing of f64 audio (element offsets) — kept local: that header is shared with the
// headline kernel's translation unit
template <int SRC, int HL, int HR>
AUKIT_DEV WaveTile describe64(const ResampleParams &P, const FastParams &F, unsigned t, Seg &sg_out, unsigned &tin_out) {
    using T = SrcTraits<SRC>;
    unsigned sidx, tin;
    if (P.tiles_per_seg) { sidx = t / P.tiles_per_seg; tin = t - sidx * P.tiles_per_seg; }
    else { sidx = P.tile_seg[t]; tin = t - P.seg_tile0[sidx]; }
    const Seg sg = P.segs[sidx];
    sg_out = sg; tin_out = tin;
    WaveTile w;
    const unsigned o0 = tin * (unsigned)WT;
    w.cnt = o0 < sg.n_out ? min((unsigned)WT, sg.n_out - o0) : 0u;
    const unsigned td = tin * F.wd;
    const unsigned tq = td / F.b;
    const unsigned kb = tin * F.wc + tq;
    w.r0 = td - tq * F.b;
    const unsigned klast = w.cnt ? (w.r0 + (w.cnt - 1) * F.a) / F.b : 0u;
    w.k_lo = 1 + (int)kb - HL;
    w.n_stage = (int)klast + 1 + HL + HR;
    w.w_lo = sg.w_lo;
    w.w_hi = sg.w_hi;
    if constexpr (SRC == SRC_AUDIO_F64) w.base = P.src + 8 * (size_t)P.src_off[sg.stream] + 8 * sg.src_base;
    else w.base = P.src + (size_t)P.src_off[sg.stream] + (long long)T::BYTES * sg.src_base;
    const unsigned char *a0 = w.base + (long long)T::BYTES * w.k_lo;
    w.al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
    w.head = (int)(a0 - w.al) / T::BYTES;
    w.nvec = (w.head + w.n_stage + T::SPV - 1) / T::SPV;
    w.orow = nullptr;
    return w;
}

template <int SRC>
AUKIT_DEV double sample64(const unsigned char *q) {
    if constexpr (SRC == SRC_PCM_S16LE_MONO) {
        const short s = (short)(q[0] | q[1] << 8);
        const double x = (double)s;
        return s < 0 ? x * (1.0 / 32768.0) : div_rcp(x, 32767.0, 1.0 / 32767.0);  // s / (s < 0 and 32768 or 32767)  :1081
    } else {
        return *reinterpret_cast<const double *>(q);
    }
}

template <typename T> AUKIT_DEV void store_exact(T *p, double v) { *p = (T)v; }

// the previous lane's double (lane 0: `carry`) and lane 63's, in the VALU (wave-wide DPP shift: an invalid source lane keeps `old`)
AUKIT_DEV double prev_lane64(double s, double carry) {
    const long long sb = __double_as_longlong(s), cb = __double_as_longlong(carry);
    const int lo = __builtin_amdgcn_update_dpp((int)cb, (int)sb, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(cb >> 32), (int)(sb >> 32), 0x138, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
AUKIT_DEV double last_lane64(double s) {
    const long long sb = __double_as_longlong(s);
    const int lo = __builtin_amdgcn_readlane((int)sb, 63), hi = __builtin_amdgcn_readlane((int)(sb >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// EPI 0: Audio:resample (copy unclamped at integer positions, clamp to [-1, 1] otherwise, :666-668)
// EPI 1: aukit.stream.pcm (:2397-2403): 2-tap low-pass on the RAW previous sample of the same iterator call (0 for its first output),
//        x 127 / 128, clamp to [-128, 127]; a segment is one iterator call, so a tile with tin == 0 starts with ls = 0
template <int SRC, int INTERP, int NV, typename OUT_T, int EPI = 0>
__global__ __launch_bounds__(256) void k_exact_wave(const ResampleParams P, const FastParams F) {
    extern __shared__ double smd[];
    using T = SrcTraits<SRC>;
    // one more tap to the left than the polynomial needs: at a mathematically integer position the reference's x may round to just
    // below the integer, and its floor(x) is then one table index lower; the stream epilogue needs the sample before the tile too
    constexpr int HL = (INTERP == AUKIT_INTERP_CUBIC ? 1 : 0) + 1 + (EPI == 1 ? 1 : 0), HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *const sm = smd + wave * (unsigned)F.cap;
    const unsigned nwaves = gridDim.x * 4u;
    OUT_T *const out = reinterpret_cast<OUT_T *>(P.out);

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    uint4 pre[NV];
    Seg sg, sg_n;
    unsigned tin, tin_n;
    WaveTile cur = describe64<SRC, HL, HR>(P, F, t, sg, tin);
    issue_loads<NV>(P, cur, lane, pre);
    for (;;) {
        // ---- window → LDS as the reference's doubles
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = lane + 64 * i;
            if (v >= cur.nvec) continue;
            if constexpr (SRC == SRC_PCM_S16LE_MONO) {
                const unsigned ww[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
                double d[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const short s = (short)(ww[e >> 1] >> (16 * (e & 1)));
                    const double x = (double)s;
                    d[e] = s < 0 ? x * (1.0 / 32768.0) : div_rcp(x, 32767.0, 1.0 / 32767.0);
                }
                double2 *o = reinterpret_cast<double2 *>(sm + 8 * v);
                o[0] = make_double2(d[0], d[1]); o[1] = make_double2(d[2], d[3]); o[2] = make_double2(d[4], d[5]); o[3] = make_double2(d[6], d[7]);
            } else {
                *reinterpret_cast<uint4 *>(sm + 2 * v) = pre[i];
            }
        }
        {
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare: vectors that straddle the allocation were zero-filled
                for (int idx = lane; idx < cur.nvec * T::SPV; idx += 64) {
                    const unsigned char *q = cur.al + (size_t)idx * T::BYTES;
                    const unsigned char *vb = cur.al + 16 * (size_t)(idx / T::SPV);
                    if (!(vb >= P.safe_lo && vb + 16 <= P.safe_hi)) sm[idx] = (q >= P.safe_lo && q + T::BYTES <= P.safe_hi) ? sample64<SRC>(q) : 0.0;
                }
            }
            // slots below / above the table replicate its ends (never read: eval_at clamps its indices to [w_lo, w_hi]; kept defined)
            const int k_hi = cur.k_lo + cur.n_stage - 1;
            if (cur.k_lo < cur.w_lo) {
                const double e_lo = sample64<SRC>(cur.base + (long long)T::BYTES * cur.w_lo);
                for (int idx = lane; idx < cur.w_lo - cur.k_lo; idx += 64) sm[cur.head + idx] = e_lo;
            }
            if (k_hi > cur.w_hi) {
                const double e_hi = sample64<SRC>(cur.base + (long long)T::BYTES * cur.w_hi);
                const int first = cur.w_hi + 1 - cur.k_lo;
                for (int idx = lane; idx < k_hi - cur.w_hi; idx += 64) sm[cur.head + first + idx] = e_hi;
            }
        }
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe64<SRC, HL, HR>(P, F, tn, sg_n, tin_n);
            issue_loads<NV>(P, nxt, lane, pre);  // in flight while this tile is evaluated
        }
        const double *tab_klo = sm + cur.head;  // slot of table index cur.k_lo
        OUT_T *orow = out + sg.out_off + (size_t)tin * WT;
        double carry = 0;  // EPI 1: ls, the RAW sample before the tile's first output (Q2)
        if constexpr (EPI == 1) {
            bool ii;
            if (tin > 0) carry = eval_at<INTERP>(P, sg, tab_klo, cur.k_lo, tin * (unsigned)WT - 1, &ii);
        }
        for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
            const unsigned j = rb + lane;
            const bool active = j < cur.cnt;
            const unsigned o = tin * (unsigned)WT + (active ? j : cur.cnt - 1);
            bool isint;
            const double s = eval_at<INTERP>(P, sg, tab_klo, cur.k_lo, o, &isint);
            if constexpr (EPI == 0) {
                if (active) store_exact<OUT_T>(orow + j, isint ? s : lua_clamp(s, -1, 1));  // :667-668
            } else {
                const double prev = prev_lane64(s, carry);
                carry = last_lane64(s);
                const double ns = prev + P.lp_alpha * (s - prev);                                          // :2401
                if (active) store_exact<OUT_T>(orow + j, lua_clamp(ns * (ns < 0 ? 128 : 127), -128, 127));  // :2402
            }
        }
        if (!more) break;
        cur = nxt; sg = sg_n; tin = tin_n;
        t = tn;
    }
}

bool fast_eligible(int src_kind, int interp, double old_rate, double new_rate, FastParams &F);

template <int SRC, int INTERP, typename OUT_T, int EPI = 0>
static void launch_exact_nv(int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid, hipStream_t st) {
    switch (nv) {
    case 1: hipLaunchKernelGGL((k_exact_wave<SRC, INTERP, 1, OUT_T, EPI>), dim3(grid), dim3(256), lds, st, P, F); break;
    case 2: hipLaunchKernelGGL((k_exact_wave<SRC, INTERP, 2, OUT_T, EPI>), dim3(grid), dim3(256), lds, st, P, F); break;
    case 4: hipLaunchKernelGGL((k_exact_wave<SRC, INTERP, 4, OUT_T, EPI>), dim3(grid), dim3(256), lds, st, P, F); break;
    default: hipLaunchKernelGGL((k_exact_wave<SRC, INTERP, 8, OUT_T, EPI>), dim3(grid), dim3(256), lds, st, P, F); break;
    }
}

// returns true when this kernel took the launch (*rc = its status).  src_kind: SRC_PCM_S16LE_MONO or SRC_AUDIO_F64; dtype of the
// output: AUKIT_F64, or AUKIT_F32 (exact-math option: the exact value rounded once).  Needs linear / cubic and integer rates.
bool exact_wave_try(aukit_ctx *ctx, int src_kind, int interp, double old_rate, double new_rate, const std::vector<Seg> &segs, ResampleParams &P, int dtype,
                    uint64_t algorithmic_bytes, int *rc, int epi) {
    if (epi && src_kind != SRC_PCM_S16LE_MONO) return false;
    if (getenv("AUKIT_EXACT_TILED")) return false;  // A/B: the workgroup-tiled kernel
    if (src_kind != SRC_PCM_S16LE_MONO && src_kind != SRC_AUDIO_F64) return false;
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32) return false;
    FastParams F;
    if (!fast_eligible(SRC_PCM_S16LE_MONO, interp, old_rate, new_rate, F)) return false;  // the rational a / b of the ratio, the division magic
    for (const Seg &g : segs)
        if (g.w_hi < g.w_lo && g.n_out) return false;
    const int spv = src_kind == SRC_PCM_S16LE_MONO ? 8 : 2;
    const int hl = (interp == AUKIT_INTERP_CUBIC ? 1 : 0) + 1 + (epi ? 1 : 0), hr = interp == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int win = (int)(((unsigned long long)(WT - 1) * F.a) / F.b) + 2 + hl + hr;
    int nv = (win + 2 * spv + 64 * spv - 1) / (64 * spv);
    nv = nv <= 1 ? 1 : (nv <= 2 ? 2 : (nv <= 4 ? 4 : (nv <= 8 ? 8 : 0)));
    if (!nv) return false;
    uint64_t max_tiles = 0, max_out = 0;
    for (const Seg &g : segs) { max_tiles = std::max<uint64_t>(max_tiles, (g.n_out + WT - 1) / WT); max_out = std::max<uint64_t>(max_out, g.n_out); }
    F.wc = (unsigned)(((unsigned long long)WT * F.a) / F.b);
    F.wd = (unsigned)(((unsigned long long)WT * F.a) % F.b);
    if ((double)max_tiles * (double)F.wd >= 4294967296.0 || ((double)max_tiles + 1) * (double)F.wc >= 2147483648.0) return false;
    if (((double)F.b + (double)WT * (double)F.a) * (double)F.b >= 4294967296.0) return false;
    F.cap = nv * 64 * spv;  // doubles per wave window
    const size_t lds = (size_t)F.cap * 8 * 4;
    if (lds > 64 * 1024) return false;
    P.ratio = new_rate / old_rate;
    P.rcp = 1.0 / P.ratio;
    P.exact_rcp = exact_div_verified(ctx, P.ratio, max_out + 2) ? 1 : 0;
    P.halo_l = hl; P.halo_r = hr; P.sinc_w = ctx->sinc_w;
    if ((*rc = plan_tiles_sized(ctx, segs, WT, P))) return true;
    if (P.n_tiles == 0) { *rc = AUKIT_OK; return true; }
    unsigned per_cu = 128;   // workgroups per CU in the grid: 16 / 32 / 64 / 128 / 256 measured 5.24 / 5.03 / 4.99 / 4.96 / 4.94 ms on config T with F64 storage
    if (const char *e = getenv("AUKIT_EXACT_PER_CU")) { const int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }   // tuning knob
    const unsigned grid = std::min<unsigned>((P.n_tiles + 3) / 4, (unsigned)ctx->num_cus * per_cu);
    if ((*rc = ctx_begin_kernel(ctx))) return true;
#define AUKIT_EW(S, I) do { if (dtype == AUKIT_F64) launch_exact_nv<S, I, double>(nv, P, F, lds, grid, ctx->stream); else launch_exact_nv<S, I, float>(nv, P, F, lds, grid, ctx->stream); } while (0)
    if (epi) {
        if (dtype == AUKIT_F64) { if (interp == AUKIT_INTERP_LINEAR) launch_exact_nv<SRC_PCM_S16LE_MONO, AUKIT_INTERP_LINEAR, double, 1>(nv, P, F, lds, grid, ctx->stream); else launch_exact_nv<SRC_PCM_S16LE_MONO, AUKIT_INTERP_CUBIC, double, 1>(nv, P, F, lds, grid, ctx->stream); }
        else { if (interp == AUKIT_INTERP_LINEAR) launch_exact_nv<SRC_PCM_S16LE_MONO, AUKIT_INTERP_LINEAR, float, 1>(nv, P, F, lds, grid, ctx->stream); else launch_exact_nv<SRC_PCM_S16LE_MONO, AUKIT_INTERP_CUBIC, float, 1>(nv, P, F, lds, grid, ctx->stream); }
    } else if (src_kind == SRC_PCM_S16LE_MONO) { if (interp == AUKIT_INTERP_LINEAR) AUKIT_EW(SRC_PCM_S16LE_MONO, AUKIT_INTERP_LINEAR); else AUKIT_EW(SRC_PCM_S16LE_MONO, AUKIT_INTERP_CUBIC); }
    else { if (interp == AUKIT_INTERP_LINEAR) AUKIT_EW(SRC_AUDIO_F64, AUKIT_INTERP_LINEAR); else AUKIT_EW(SRC_AUDIO_F64, AUKIT_INTERP_CUBIC); }
#undef AUKIT_EW
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_exact_wave launch failed"); return true; }
    static thread_local char nm[96];
    snprintf(nm, sizeof nm, "k_exact_wave<%s,%s,nv%d%s>", src_kind == SRC_PCM_S16LE_MONO ? "pcm_s16le_mono" : "audio_f64", interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv,
             epi ? ",stream_pcm" : "");
    *rc = ctx_end_kernel(ctx, nm, algorithmic_bytes);
    return true;
}

}  // namespace aukit
