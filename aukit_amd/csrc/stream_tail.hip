// stream_tail.hip — k_iir_tail: resample + recursive one-pole low-pass of aukit.stream.qoa / aukit.stream.flac in one launch (stream_tail.h).
//
// The recurrence `s = ls + lp_alpha * (s - ls); ls = s` (aukit.lua:3324-3325, :3179-3180) runs over a whole iterator call (QOA: ≈ 48 000
// outputs) or a whole FLAC block.  It is a contraction: y_i = a y_(i-1) + lp_alpha x_i with a = 1 - lp_alpha = exp(-2 pi rate / 96000), so
// what a state W outputs back contributes is a^W of it — below 1e-16 after W = ceil(37 / (2 pi rate / 96000)) outputs (13 at 44.1 kHz, 71 at
// 8 kHz; more by log M for samples that can reach a magnitude M).  A tile of T outputs therefore needs nothing from the tile before it: it interpolates its own outputs and the W before them,
// and every thread runs the recurrence over its E consecutive outputs behind a private warm-up of W (from the exact seed where the
// warm-up reaches the start of the job).  No scan, no carry chain between tiles, the interpolated samples never leave LDS:
//   1. the window of the table the tile touches → LDS as doubles (coalesced row reads; indices 0 / -1 are the history samples);
//   2. lane ↔ consecutive output: position and interpolation in the reference's fp64 operation order (pos_of / cubic_exact), → LDS;
//   3. thread ↔ E consecutive outputs: warm-up + recurrence, results in registers, then back to LDS;
//   4. lane ↔ consecutive output: epilogue (QOA: as is / channel mean; FLAC: scale and clamp) and coalesced stores.
// HBM traffic = the rows once (+ W / T of overlap) + the outputs once.
#include <algorithm>
#include "stream_tail.h"
#include "resample.h"
#include "resample_dev.h"

namespace aukit {

struct TailParams {
    const TailJob *jobs;
    unsigned tiles_per_job;      // tile t = (job t / tiles_per_job, tile t % tiles_per_job of it); tiles beyond a job's outputs are skipped
    unsigned long long n_tiles;
    int W, cap, xbn, C;          // warm-up outputs; LDS doubles per channel for the table window / for the tile's samples; channels per job
    double ratio, rcp;
    int exact;
    double lp_alpha, full;
    const void *rows;
    void *out;
    // k_iir_tail_fast: x - 1 = o fa / fb exactly; fmagic = ceil(2^32 / fb); 256 fa = dq256 fb + dr256; wg: fb phases (cubic: w0..w3, linear: fx)
    unsigned fa, fb, fmagic, dq256, dr256;
    const float *wg;
};

template <typename R> static AUKIT_DEV double tail_row(const R *rows, unsigned long long at, double full) {
    if constexpr (sizeof(R) == 4) return (double)rows[at] * (1.0 / full);   // :505: full is a power of two, so the reciprocal multiply IS the quotient
    else return (double)rows[at];
}

template <int KIND, int INTERP, int E, typename R, typename OUT_T>
__global__ __launch_bounds__(256) void k_iir_tail(const TailParams P) {
    extern __shared__ double tsm[];
    constexpr int T = 256 * E;
    const int tid = threadIdx.x;
    double *const win = tsm;
    double *const xb = tsm + (size_t)P.C * P.cap;
    const R *const rows = reinterpret_cast<const R *>(P.rows);
    OUT_T *const out = reinterpret_cast<OUT_T *>(P.out);
    auto skew = [](int i) { return i + i / E; };
    auto pos = [&](unsigned o) { const double nn = (double)o; return (P.exact ? div_rcp(nn, P.ratio, P.rcp) : nn / P.ratio) + 1.0; };  // x = (i - 1) / ratio + 1
    for (unsigned long long t = blockIdx.x; t < P.n_tiles; t += gridDim.x) {
        const unsigned long long ji = t / P.tiles_per_job;
        const TailJob job = P.jobs[ji];
        const unsigned o0 = (unsigned)(t - ji * P.tiles_per_job) * (unsigned)T;
        if (o0 >= (unsigned)job.nout) continue;   // (uniform: a short job's spare tiles)
        const int cnt = (int)min((unsigned)T, (unsigned)job.nout - o0);
        const int wl = (int)min((unsigned)P.W, o0);
        const unsigned of = o0 - (unsigned)wl;
        const int total = wl + cnt;
        const int n = job.n;
        int k_lo = (int)floor(pos(of)) - 1, k_hi = (int)floor(pos(o0 + (unsigned)cnt - 1)) + 2;
        k_lo = max(k_lo, -1);
        k_hi = min(k_hi, n);
        const int nst = k_hi - k_lo + 1;
        __syncthreads();   // the previous tile's reads are done
        for (int c = 0; c < P.C; c++) {
            double z0 = 0, m1 = 0;   // table indices 0, -1: the history
            if (job.last_off != ~0ull) z0 = tail_row(rows, job.last_off + (unsigned long long)c * job.last_cstride, P.full);
            if (job.m1_off != ~0ull) m1 = tail_row(rows, job.m1_off + (unsigned long long)c * job.last_cstride, P.full);
            const unsigned long long base = job.src_off + (unsigned long long)c * job.src_cstride;
            for (int rel = tid; rel < nst; rel += 256) {
                const int k = k_lo + rel;
                win[c * P.cap + rel] = k >= 1 ? tail_row(rows, base + (unsigned long long)(k - 1), P.full) : (k == 0 ? z0 : m1);
            }
        }
        __syncthreads();
        for (int c = 0; c < P.C; c++) {
            const double *tab = win + c * P.cap - k_lo;   // tab[k] = table index k
            for (int idx = tid; idx < total; idx += 256) {
                const double x = pos(of + (unsigned)idx);
                const double ffx = floor(x);
                const int k = (int)ffx;
                double s;
                if (x == ffx) s = tab[k];
                else {
                    const double fx = x - ffx;
                    if constexpr (INTERP == AUKIT_INTERP_NONE) s = tab[k];
                    else if constexpr (INTERP == AUKIT_INTERP_LINEAR) { const double a = tab[k], b = (k + 1 <= n) ? tab[k + 1] : a; s = linear_exact(a, b, fx); }
                    else {
                        const double p1 = tab[k], p0 = (k - 1 >= -1) ? tab[k - 1] : p1, p2 = (k + 1 <= n) ? tab[k + 1] : p1, p3 = (k + 2 <= n) ? tab[k + 2] : p2;
                        s = cubic_exact(p0, p1, p2, p3, fx);
                    }
                    if constexpr (KIND == TAIL_QOA) s = lua_clamp(s, -128, 127);   // :3323
                }
                xb[c * P.xbn + skew(idx)] = s;
            }
        }
        __syncthreads();
        const int e0 = tid * E;
        for (int c = 0; c < P.C; c++) {
            double y[E];
            if (e0 < cnt) {
                const double *xs = xb + c * P.xbn;
                const int start = wl + e0;
                const int ws = min(P.W, start), begin = start - ws;
                double ls = 0;
                if (of + (unsigned)begin == 0) {   // the warm-up reaches the job's first output: the exact seed
                    double z0 = 0;
                    if (job.last_off != ~0ull) z0 = tail_row(rows, job.last_off + (unsigned long long)c * job.last_cstride, P.full);
                    ls = KIND == TAIL_FLAC ? z0 / (z0 < 0 ? 128 : 127) : z0;   // :3172 / :3316
                }
                for (int i = begin; i < start; i++) { const double s = ls + P.lp_alpha * (xs[skew(i)] - ls); ls = s; }
#pragma unroll
                for (int i = 0; i < E; i++) {
                    const double xv = e0 + i < cnt ? xs[skew(start + i)] : 0.0;
                    const double s = ls + P.lp_alpha * (xv - ls);   // :3324 / :3179
                    ls = s;
                    y[i] = s;
                }
            }
            __syncthreads();   // every warm-up has read this channel's interpolated samples: the filtered ones take their place
            if (e0 < cnt) {
#pragma unroll
                for (int i = 0; i < E; i++) if (e0 + i < cnt) xb[c * P.xbn + skew(wl + e0 + i)] = y[i];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < cnt; idx += 256) {
            const unsigned long long o = (unsigned long long)o0 + (unsigned)idx;
            if constexpr (KIND == TAIL_QOA) {
                if (P.C == 1) out[job.out_off + o] = (OUT_T)xb[skew(wl + idx)];
                else {
                    double acc = 0;
                    for (int c = 0; c < P.C; c++) acc = acc + xb[c * P.xbn + skew(wl + idx)];
                    out[job.out_off + o] = (OUT_T)(acc / P.C);   // lines[1][i] = n / file_channels  :3329
                }
            } else {
                const double s = xb[skew(wl + idx)];
                out[job.out_off + o] = (OUT_T)lua_clamp(s * (s < 0 ? 128 : 127), -128, 127);   // :3181
            }
        }
    }
}

// The tolerance path (F32 storage): the same four phases with the interpolation in f32 — positions as exact rationals (q, rem advanced by
// additions), the phase's weights from an LDS table, four FMAs per output on an f32 window — and the recurrence still in fp64.  Against the
// reference-order kernel above: ≈ 1e-7 of the [-128, 127] scale (the bar of the un-floored stream outputs is 1e-6 RMS of it);
// 117 → ≈ 35 instructions per output.
// NT = threads per tile: 256 (a workgroup tile, three block barriers per tile: the first cut) or 64 — a wave-private tile, four independent
// tiles per workgroup, wave barriers only (round 3: tiles need nothing from each other, so nothing is gained by making four waves march in step)
template <int KIND, int INTERP, int E, typename R, typename OUT_T, int NT>
__global__ __launch_bounds__(256) void k_iir_tail_fast(const TailParams P) {
    extern __shared__ float fsm[];
    constexpr int T = NT * E;
    constexpr int WF = INTERP == AUKIT_INTERP_CUBIC ? 4 : 1;
    const int tid = NT == 256 ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    const unsigned sub = NT == 256 ? 0u : (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nsub = 256 / NT;
    auto tile_sync = [&]() { if constexpr (NT == 256) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } };
    float *const wt = fsm;
    float *const win = fsm + ((P.fb * WF + 3) & ~3u) + sub * (unsigned)(P.C * (P.cap + P.xbn));
    float *const xb = win + (size_t)P.C * P.cap;
    const R *const rows = reinterpret_cast<const R *>(P.rows);
    OUT_T *const out = reinterpret_cast<OUT_T *>(P.out);
    for (unsigned i = threadIdx.x; i < P.fb * WF; i += 256) wt[i] = P.wg[i];
    __syncthreads();
    const double full_rcp = 1.0 / P.full;   // a power of two
    auto rowf = [&](unsigned long long at) -> float { if constexpr (sizeof(R) == 4) return (float)((double)rows[at] * full_rcp); else return (float)rows[at]; };
    auto skew = [](int i) { return i + i / E; };
    auto qr = [&](unsigned o, unsigned &q, unsigned &rem) { q = __umulhi(o * P.fa, P.fmagic); rem = o * P.fa - q * P.fb; };
    for (unsigned long long t = (unsigned long long)blockIdx.x * nsub + sub; t < P.n_tiles; t += (unsigned long long)gridDim.x * nsub) {
        const unsigned long long ji = t / P.tiles_per_job;
        const TailJob job = P.jobs[ji];
        const unsigned o0 = (unsigned)(t - ji * P.tiles_per_job) * (unsigned)T;
        if (o0 >= (unsigned)job.nout) continue;   // (uniform: a short job's spare tiles)
        const int cnt = (int)min((unsigned)T, (unsigned)job.nout - o0);
        const int wl = (int)min((unsigned)P.W, o0);
        const unsigned of = o0 - (unsigned)wl;
        const int total = wl + cnt;
        const int n = job.n;
        unsigned qf, rf, ql, rl;
        qr(of, qf, rf);
        qr(o0 + (unsigned)cnt - 1, ql, rl);
        const int k_lo = (int)qf;                      // floor(x) - 1 of the tile's first output: the lowest tap
        const int k_hi = min((int)ql + 3, n);
        tile_sync();
        for (int c = 0; c < P.C; c++) {
            float z0 = 0;   // table index 0: the history sample
            if (job.last_off != ~0ull) z0 = rowf(job.last_off + (unsigned long long)c * job.last_cstride);
            const unsigned long long base = job.src_off + (unsigned long long)c * job.src_cstride;
            // The window as 16-byte vectors (round 3).  The first cut walked it one element per thread and turn — for int8 rows eight dependent
            // byte loads per thread and tile, each waited for where it was issued.  Now: the aligned vectors that lie wholly inside the window
            // (all issued before any is used), the few elements in front of and behind them one per thread; nothing outside the window is read.
            constexpr int EPV = 16 / (int)sizeof(R);
            const int kf = k_lo < 1 ? 1 : k_lo, nreal = k_hi - kf + 1;
            float *const wrow = win + c * P.cap + (kf - k_lo);      // slot of table index kf
            if (k_lo < 1 && tid == 0) win[c * P.cap] = z0;
            if (nreal > 0) {
                const R *const p0 = rows + base + (unsigned long long)(kf - 1);
                int head = (int)(((16u - (unsigned)((uintptr_t)p0 & 15u)) & 15u) / (unsigned)sizeof(R));
                head = head < nreal ? head : nreal;
                const int nvec = (nreal - head) / EPV, tail0 = head + nvec * EPV;
                constexpr int VU = NT == 256 ? (sizeof(R) == 1 ? 1 : 2) : (sizeof(R) == 1 ? 1 : 2);          // vectors per thread issued together (a tile's window: ~120 of int8, ~480 of int32)
                for (int v0 = 0; v0 < nvec; v0 += NT * VU) {
                    uint4 vv[VU];
#pragma unroll
                    for (int u = 0; u < VU; u++) { const int v = v0 + tid + NT * u; vv[u] = make_uint4(0, 0, 0, 0); if (v < nvec) vv[u] = *reinterpret_cast<const uint4 *>(p0 + head + (size_t)v * EPV); }
#pragma unroll
                    for (int u = 0; u < VU; u++) {
                        const int v = v0 + tid + NT * u;
                        if (v >= nvec) continue;
                        float *d = wrow + head + v * EPV;
                        const unsigned ww[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
                        if constexpr (sizeof(R) == 4) {
#pragma unroll
                            for (int j = 0; j < 4; j++) d[j] = (float)((double)(int)ww[j] * full_rcp);
                        } else {
#pragma unroll
                            for (int j = 0; j < 16; j++) d[j] = (float)(signed char)(ww[j >> 2] >> (8 * (j & 3)));
                        }
                    }
                }
                if (tid < head) wrow[tid] = rowf(base + (unsigned long long)(kf - 1 + tid));
                if (tid >= 32 && tid - 32 < nreal - tail0) wrow[tail0 + tid - 32] = rowf(base + (unsigned long long)(kf - 1 + tail0 + tid - 32));
            }
        }
        tile_sync();
        for (int c = 0; c < P.C; c++) {
            const float *tab = win + c * P.cap - k_lo;   // tab[k] = table index k
            unsigned q, rem;
            qr(of + (unsigned)tid, q, rem);
            for (int idx = tid; idx < total; idx += NT, q += P.dq256, rem += P.dr256) {   // (dq256 / dr256: the step of NT outputs)
                if (rem >= P.fb) { rem -= P.fb; q++; }
                const int k = (int)q + 1;   // floor(x)
                float s;
                if constexpr (INTERP == AUKIT_INTERP_NONE) {
                    // data[math.floor(x)] is a step function of x: where the exact position is an integer the reference's ROUNDED x may sit just
                    // below it (44.1 -> 48 kHz: 192 of 300 such positions per second) and picks the sample before — only there take its fp64 x
                    int kk = k;
                    if (rem == 0) { const double nn = (double)(of + (unsigned)idx); kk = (int)floor((P.exact ? div_rcp(nn, P.ratio, P.rcp) : nn / P.ratio) + 1.0); }
                    s = tab[kk];
                }
                else if constexpr (INTERP == AUKIT_INTERP_LINEAR) { const float a = tab[k], b = tab[min(k + 1, n)]; s = __builtin_fmaf(b - a, wt[rem], a); }   // data[ffx+1] or data[ffx]
                else {
                    const float4 w = *reinterpret_cast<const float4 *>(wt + 4 * rem);
                    const float p0 = tab[k - 1], p1 = tab[k], p2 = tab[min(k + 1, n)], p3 = tab[min(k + 2, n)];   // p2 or p1, p3 or p2 or p1  :262-264
                    s = __builtin_fmaf(w.w, p3, __builtin_fmaf(w.z, p2, __builtin_fmaf(w.y, p1, w.x * p0)));
                }
                if constexpr (KIND == TAIL_QOA) s = __builtin_amdgcn_fmed3f(s, -128.0f, 127.0f);   // :3323 (a table entry is inside anyway)
                xb[c * P.xbn + skew(idx)] = s;
            }
        }
        tile_sync();
        // thread ↔ E consecutive outputs: warm-up, recurrence, epilogue, and E consecutive elements stored straight from registers — a wave's
        // 64 × E outputs are one contiguous run, so its two 16-byte stores per lane fill whole lines between them (no trip back through LDS)
        const int e0 = tid * E;
        float acc[E];
#pragma unroll
        for (int i = 0; i < E; i++) acc[i] = 0.f;
        if (e0 < cnt) {
            for (int c = 0; c < P.C; c++) {
                const float *xs = xb + c * P.xbn;
                const int start = wl + e0;
                const int ws = min(P.W, start), begin = start - ws;
                double ls = 0;
                if (of + (unsigned)begin == 0) {   // the warm-up reaches the job's first output: the exact seed
                    double z0 = 0;
                    if (job.last_off != ~0ull) z0 = tail_row(rows, job.last_off + (unsigned long long)c * job.last_cstride, P.full);
                    ls = KIND == TAIL_FLAC ? z0 / (z0 < 0 ? 128 : 127) : z0;   // :3172 / :3316
                }
                for (int i = begin; i < start; i++) { ls = __builtin_fma(P.lp_alpha, (double)xs[skew(i)] - ls, ls); }
#pragma unroll
                for (int i = 0; i < E; i++) {
                    const double xv = e0 + i < cnt ? (double)xs[skew(start + i)] : 0.0;
                    const double s = __builtin_fma(P.lp_alpha, xv - ls, ls);   // :3324 / :3179 (fused: the tolerance path)
                    ls = s;
                    if constexpr (KIND == TAIL_QOA) acc[i] = c == 0 ? (float)s : acc[i] + (float)s;   // n = n + s  :3327
                    else { const float f = (float)s; acc[i] = __builtin_amdgcn_fmed3f(f * (f < 0 ? 128.0f : 127.0f), -128.0f, 127.0f); }   // :3181
                }
            }
            if constexpr (KIND == TAIL_QOA) {
                if (P.C > 1) {
#pragma unroll
                    for (int i = 0; i < E; i++) acc[i] = acc[i] / (float)P.C;   // lines[1][i] = n / file_channels  :3329
                }
            }
        }
        if (NT == 64 && cnt == T && true) {
            // a full tile leaves through LDS once more (round 3): a lane's E consecutive results as they stand are E stores of 16 bytes at a stride of
            // 4 E bytes — every store instruction touches all 64 lines of the tile; transposed back (the skewed layout of the way in, read the other
            // way round) a store instruction writes 256 contiguous bytes
            tile_sync();
#pragma unroll
            for (int i = 0; i < E; i++) xb[skew(e0 + i)] = acc[i];
            tile_sync();
            OUT_T *const ot = out + job.out_off + (unsigned long long)o0 + (unsigned)tid;
#pragma unroll
            for (int u = 0; u < E; u++) ot[64 * u] = xb[skew(tid + 64 * u)];
        } else if (e0 < cnt) {
            OUT_T *const op = out + job.out_off + (unsigned long long)o0 + (unsigned)e0;
            if (e0 + E <= cnt) {
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
                for (int v = 0; v < E / 4; v++) { f4u w; w.x = acc[4 * v]; w.y = acc[4 * v + 1]; w.z = acc[4 * v + 2]; w.w = acc[4 * v + 3]; *reinterpret_cast<f4u *>(op + 4 * v) = w; }
            } else {
                for (int i = 0; i < E && e0 + i < cnt; i++) op[i] = acc[i];
            }
        }
    }
}

template <int KIND, int E, typename R>
static void tail_launch_fast(int interp, const TailParams &P, unsigned grid, size_t lds, hipStream_t st) {
    if (interp == AUKIT_INTERP_NONE) hipLaunchKernelGGL((k_iir_tail_fast<KIND, AUKIT_INTERP_NONE, E, R, float, 64>), dim3(grid), dim3(256), lds, st, P);
    else if (interp == AUKIT_INTERP_LINEAR) hipLaunchKernelGGL((k_iir_tail_fast<KIND, AUKIT_INTERP_LINEAR, E, R, float, 64>), dim3(grid), dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_iir_tail_fast<KIND, AUKIT_INTERP_CUBIC, E, R, float, 64>), dim3(grid), dim3(256), lds, st, P);
}

template <int KIND, int E, typename R, typename OUT_T>
static void tail_launch_interp(int interp, const TailParams &P, unsigned grid, size_t lds, hipStream_t st) {
    if (interp == AUKIT_INTERP_NONE) hipLaunchKernelGGL((k_iir_tail<KIND, AUKIT_INTERP_NONE, E, R, OUT_T>), dim3(grid), dim3(256), lds, st, P);
    else if (interp == AUKIT_INTERP_LINEAR) hipLaunchKernelGGL((k_iir_tail<KIND, AUKIT_INTERP_LINEAR, E, R, OUT_T>), dim3(grid), dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_iir_tail<KIND, AUKIT_INTERP_CUBIC, E, R, OUT_T>), dim3(grid), dim3(256), lds, st, P);
}
template <int KIND, int E, typename R>
static void tail_launch_out(int interp, int dtype, const TailParams &P, unsigned grid, size_t lds, hipStream_t st) {
    if (dtype == AUKIT_F64) tail_launch_interp<KIND, E, R, double>(interp, P, grid, lds, st);
    else tail_launch_interp<KIND, E, R, float>(interp, P, grid, lds, st);
}

struct TailShape { bool ok = false, fast = false; int C = 1, E = 4, T = 1024, W = 0, cap = 0, xbn = 0, wf = 1; unsigned long long fa = 0, fb = 0; size_t lds = 0; double ratio = 1, lp_alpha = 0; };
static TailShape tail_shape(aukit_ctx *ctx, int kind, int rows_kind, int mix_channels, double rate, double full, int interp, int dtype, uint64_t max_nout, uint64_t avg_nout) {
    TailShape S;
    if (interp < 0 || interp > 2 || (dtype != AUKIT_F64 && dtype != AUKIT_F32) || getenv("AUKIT_NO_IIR_TAIL")) return S;
    if (ctx->exact_math == 2) return S;   // "always the reference-order kernels": the carried two-pass recurrence, not the warm-up tiles
    if (kind == TAIL_QOA && rows_kind != TAIL_ROWS_I8) return S;
    if (kind == TAIL_FLAC && (rows_kind == TAIL_ROWS_I8 || mix_channels > 1)) return S;
    S.C = std::max(1, mix_channels);
    S.ratio = 48000 / rate;
    S.lp_alpha = 1 - std::exp(-(rate / 96000) * 2 * M_PI);   // :3251 / :3155
    const double decay = (rate / 96000) * 2 * M_PI;          // a = exp(-decay)
    if (!(decay > 0) || !(S.ratio > 0)) return S;
    // what the samples can reach: a state of magnitude M still shows as M a^W after W outputs, so the warm-up grows with log M — int8 rows 128,
    // int32 rows 2^31 / full (garbage frames decode to anything an int32 holds); rows of doubles (values beyond int32) have no bound: not served
    if (rows_kind == TAIL_ROWS_F64) return S;
    const double mag = rows_kind == TAIL_ROWS_I8 ? 128.0 : std::max(1.0, std::ldexp(1.0, 31) / full);
    const double wd = std::ceil((37.0 + std::log(mag)) / decay);
    // F32 storage: the f32 interpolation (k_iir_tail_fast, wave-private tiles) — integer sample rates with at most 512 output phases
    S.fast = dtype == AUKIT_F32 && !ctx->exact_math && !getenv("AUKIT_NO_TAIL_FAST") && rate == std::floor(rate) && rate <= 4e9;
    if (S.fast) {
        unsigned long long x = 48000, y = (unsigned long long)rate;
        while (y) { const unsigned long long tq = x % y; x = y; y = tq; }
        S.fa = (unsigned long long)rate / x; S.fb = 48000 / x;   // x - 1 = (i - 1) / ratio = (i - 1) fa / fb
        if (S.fb == 1) { S.fa *= 2; S.fb = 2; }
        S.fast = S.fb <= 512 && ((double)max_nout * (double)S.fa + (double)S.fb) * (double)S.fb < 4294967296.0 && wd <= 192;
    }
    S.E = ((avg_nout >= 6144 && kind == TAIL_QOA) || (S.fast && avg_nout >= 2048)) ? 8 : 4;
    if (S.fast && S.C == 1 && avg_nout >= 4096 && !getenv("AUKIT_TAIL_E8")) S.E = 16;   // a thread's warm-up (W steps) is paid once per E outputs: stream.flac's tail 4.6 / 3.3 / 3.1 ms for E = 4 / 8 / 16, stream.qoa's 2.8 -> 2.5   // long jobs (QOA calls): 2048-output tiles; FLAC blocks: 1024; the fast kernel's wave tiles: 512 outputs from 2048 per job on
    S.T = (S.fast ? 64 : 256) * S.E;
    if (wd > (double)S.T) return S;
    S.W = (int)wd;
    const double capd = std::ceil((double)(S.T + S.W) / S.ratio) + 8;
    if (capd > 32768) return S;
    S.cap = ((int)capd + 3) & ~3;
    S.xbn = ((S.T + S.W) + (S.T + S.W) / S.E + 2 + 3) & ~3;
    S.wf = interp == AUKIT_INTERP_CUBIC ? 4 : 1;
    S.lds = S.fast ? ((((size_t)S.fb * S.wf + 3) & ~(size_t)3) + 4 * (size_t)S.C * ((size_t)S.cap + (size_t)S.xbn)) * 4 : (size_t)S.C * ((size_t)S.cap + (size_t)S.xbn) * 8;   // (fast: four wave tiles per workgroup)
    if (S.lds > 150 * 1024) return S;
    S.ok = true;
    return S;
}

bool iir_tail_served(aukit_ctx *ctx, int kind, int rows_kind, int mix_channels, double rate, double full, int interp, int dtype, uint64_t max_nout) {
    return tail_shape(ctx, kind, rows_kind, mix_channels, rate, full, interp, dtype, max_nout, max_nout).ok;
}

bool iir_tail_try_dev(aukit_ctx *ctx, int kind, int rows_kind, const void *rows, double full, const TailJob *d_jobs, size_t njobs, uint64_t max_nout, uint64_t sum_nout,
                      int mix_channels, double rate, int interp, int dtype, void *out, uint64_t algorithmic_bytes, const char *name, int *rc) {
    *rc = AUKIT_OK;
    if (!njobs || !max_nout) return true;
    const TailShape S = tail_shape(ctx, kind, rows_kind, mix_channels, rate, full, interp, dtype, max_nout, sum_nout / njobs);
    if (!S.ok) return false;
    TailParams P{};
    P.jobs = d_jobs;
    P.tiles_per_job = (unsigned)((max_nout + S.T - 1) / S.T);
    P.n_tiles = (unsigned long long)P.tiles_per_job * njobs;
    P.W = S.W; P.cap = S.cap; P.xbn = S.xbn; P.C = S.C;
    P.ratio = S.ratio; P.rcp = 1.0 / S.ratio;
    P.exact = exact_div_verified(ctx, S.ratio, std::max<uint64_t>(max_nout + 2, 1ull << 17)) ? 1 : 0;
    P.lp_alpha = S.lp_alpha; P.full = full;
    P.rows = rows; P.out = out;
    if (S.fast) {
        std::vector<float> w((size_t)S.fb * S.wf);
        for (unsigned r = 0; r < S.fb; r++) {
            const long double f = (long double)r / (long double)S.fb, f2 = f * f, f3 = f2 * f;
            if (S.wf == 1) w[r] = (float)f;
            else {
                w[4 * r] = (float)(-0.5L * f3 + f2 - 0.5L * f); w[4 * r + 1] = (float)(1.5L * f3 - 2.5L * f2 + 1.0L);
                w[4 * r + 2] = (float)(-1.5L * f3 + 2.0L * f2 + 0.5L * f); w[4 * r + 3] = (float)(0.5L * f3 - 0.5L * f2);
            }
        }
        if ((*rc = upload_table(ctx, ctx->tile_buf, w.data(), w.size() * 4))) return true;
        P.wg = reinterpret_cast<const float *>(ctx->tile_buf.p);
        P.fa = (unsigned)S.fa; P.fb = (unsigned)S.fb; P.fmagic = (unsigned)((4294967296ull + S.fb - 1) / S.fb);
        P.dq256 = (unsigned)((64ull * S.fa) / S.fb); P.dr256 = (unsigned)((64ull * S.fa) % S.fb);   // the step of one row of a wave tile (64 outputs)
    }
    unsigned per_cu = 64;   // workgroups per CU in the grid (what the LDS lets stay resident is fewer: a finer hand-out of the tiles): 8 / 16 / 32 / 64 / 128 / 256 measured 3.51 / 3.42 / 3.40 / 3.37 / 3.46 / 3.55 ms on stream.qoa
    if (const char *e = getenv("AUKIT_TAIL_PER_CU")) { const int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }   // tuning knob
    const unsigned grid = (unsigned)std::min<uint64_t>(S.fast ? (P.n_tiles + 3) / 4 : P.n_tiles, (uint64_t)ctx->num_cus * per_cu);
    if ((*rc = ctx_begin_kernel(ctx))) return true;
    const size_t lds = S.lds;
    if (S.fast) {
        if (kind == TAIL_QOA) { if (S.E == 16) tail_launch_fast<TAIL_QOA, 16, signed char>(interp, P, grid, lds, ctx->stream); else if (S.E == 8) tail_launch_fast<TAIL_QOA, 8, signed char>(interp, P, grid, lds, ctx->stream); else tail_launch_fast<TAIL_QOA, 4, signed char>(interp, P, grid, lds, ctx->stream); }
        else if (S.E == 16) tail_launch_fast<TAIL_FLAC, 16, int>(interp, P, grid, lds, ctx->stream);
        else if (S.E == 8) tail_launch_fast<TAIL_FLAC, 8, int>(interp, P, grid, lds, ctx->stream);
        else tail_launch_fast<TAIL_FLAC, 4, int>(interp, P, grid, lds, ctx->stream);
    } else if (kind == TAIL_QOA) { if (S.E == 8) tail_launch_out<TAIL_QOA, 8, signed char>(interp, dtype, P, grid, lds, ctx->stream); else tail_launch_out<TAIL_QOA, 4, signed char>(interp, dtype, P, grid, lds, ctx->stream); }
    else tail_launch_out<TAIL_FLAC, 4, int>(interp, dtype, P, grid, lds, ctx->stream);
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_iir_tail launch failed"); return true; }
    *rc = ctx_end_kernel(ctx, name, algorithmic_bytes);
    return true;
}

bool iir_tail_try(aukit_ctx *ctx, int kind, int rows_kind, const void *rows, double full, const std::vector<TailJob> &jobs, int mix_channels, double rate,
                  int interp, int dtype, void *out, uint64_t algorithmic_bytes, const char *name, int *rc) {
    *rc = AUKIT_OK;
    if (jobs.empty()) return true;
    uint64_t max_nout = 0, sum_nout = 0;
    for (const TailJob &j : jobs) { max_nout = std::max<uint64_t>(max_nout, (uint64_t)std::max(j.nout, 0)); sum_nout += (uint64_t)std::max(j.nout, 0); }
    if (max_nout == 0) return true;
    if (!tail_shape(ctx, kind, rows_kind, mix_channels, rate, full, interp, dtype, max_nout, sum_nout / jobs.size()).ok) return false;
    if ((*rc = upload_table(ctx, ctx->seg_buf, jobs.data(), jobs.size() * sizeof(TailJob)))) return true;
    return iir_tail_try_dev(ctx, kind, rows_kind, rows, full, reinterpret_cast<const TailJob *>(ctx->seg_buf.p), jobs.size(), max_nout, sum_nout, mix_channels, rate, interp, dtype, out,
                            algorithmic_bytes, name, rc);
}

}  // namespace aukit
