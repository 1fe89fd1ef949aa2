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
    const unsigned *tile_job;    // tile → job
    const unsigned *job_tile0;   // job → its first tile
    unsigned n_tiles;
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
    for (unsigned t = blockIdx.x; t < P.n_tiles; t += gridDim.x) {
        const unsigned ji = P.tile_job[t];
        const TailJob job = P.jobs[ji];
        const unsigned o0 = (t - P.job_tile0[ji]) * (unsigned)T;
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
template <int KIND, int INTERP, int E, typename R, typename OUT_T>
__global__ __launch_bounds__(256) void k_iir_tail_fast(const TailParams P) {
    extern __shared__ float fsm[];
    constexpr int T = 256 * E;
    constexpr int WF = INTERP == AUKIT_INTERP_CUBIC ? 4 : 1;
    const int tid = threadIdx.x;
    float *const wt = fsm;
    float *const win = fsm + ((P.fb * WF + 3) & ~3u);
    float *const xb = win + (size_t)P.C * P.cap;
    const R *const rows = reinterpret_cast<const R *>(P.rows);
    OUT_T *const out = reinterpret_cast<OUT_T *>(P.out);
    for (unsigned i = tid; i < P.fb * WF; i += 256) wt[i] = P.wg[i];
    const double full_rcp = 1.0 / P.full;   // a power of two
    auto rowf = [&](unsigned long long at) -> float { if constexpr (sizeof(R) == 4) return (float)((double)rows[at] * full_rcp); else return (float)rows[at]; };
    auto skew = [](int i) { return i + i / E; };
    auto qr = [&](unsigned o, unsigned &q, unsigned &rem) { q = __umulhi(o * P.fa, P.fmagic); rem = o * P.fa - q * P.fb; };
    for (unsigned t = blockIdx.x; t < P.n_tiles; t += gridDim.x) {
        const unsigned ji = P.tile_job[t];
        const TailJob job = P.jobs[ji];
        const unsigned o0 = (t - P.job_tile0[ji]) * (unsigned)T;
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
        const int nst = k_hi - k_lo + 1;
        __syncthreads();
        for (int c = 0; c < P.C; c++) {
            float z0 = 0;   // table index 0: the history sample
            if (job.last_off != ~0ull) z0 = rowf(job.last_off + (unsigned long long)c * job.last_cstride);
            const unsigned long long base = job.src_off + (unsigned long long)c * job.src_cstride;
            for (int rel = tid; rel < nst; rel += 256) {
                const int k = k_lo + rel;
                win[c * P.cap + rel] = k >= 1 ? rowf(base + (unsigned long long)(k - 1)) : z0;
            }
        }
        __syncthreads();
        for (int c = 0; c < P.C; c++) {
            const float *tab = win + c * P.cap - k_lo;   // tab[k] = table index k
            unsigned q, rem;
            qr(of + (unsigned)tid, q, rem);
            for (int idx = tid; idx < total; idx += 256, q += P.dq256, rem += P.dr256) {
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
        __syncthreads();
        // thread ↔ E consecutive outputs: warm-up, recurrence, epilogue, and E consecutive elements stored straight from registers — a wave's
        // 64 × E outputs are one contiguous run, so its two 16-byte stores per lane fill whole lines between them (no trip back through LDS)
        const int e0 = tid * E;
        if (e0 < cnt) {
            float acc[E];
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
                for (int i = begin; i < start; i++) { const double s = ls + P.lp_alpha * ((double)xs[skew(i)] - ls); ls = s; }
#pragma unroll
                for (int i = 0; i < E; i++) {
                    const double xv = e0 + i < cnt ? (double)xs[skew(start + i)] : 0.0;
                    const double s = ls + P.lp_alpha * (xv - ls);   // :3324 / :3179
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
    if (interp == AUKIT_INTERP_NONE) hipLaunchKernelGGL((k_iir_tail_fast<KIND, AUKIT_INTERP_NONE, E, R, float>), dim3(grid), dim3(256), lds, st, P);
    else if (interp == AUKIT_INTERP_LINEAR) hipLaunchKernelGGL((k_iir_tail_fast<KIND, AUKIT_INTERP_LINEAR, E, R, float>), dim3(grid), dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_iir_tail_fast<KIND, AUKIT_INTERP_CUBIC, E, R, float>), dim3(grid), dim3(256), lds, st, P);
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

bool iir_tail_try(aukit_ctx *ctx, int kind, int rows_kind, const void *rows, double full, const std::vector<TailJob> &jobs, int mix_channels, double rate,
                  int interp, int dtype, void *out, uint64_t algorithmic_bytes, const char *name, int *rc) {
    *rc = AUKIT_OK;
    if (jobs.empty()) return true;
    if (interp < 0 || interp > 2 || (dtype != AUKIT_F64 && dtype != AUKIT_F32) || getenv("AUKIT_NO_IIR_TAIL")) return false;
    if (kind == TAIL_QOA && rows_kind != TAIL_ROWS_I8) return false;
    if (kind == TAIL_FLAC && (rows_kind == TAIL_ROWS_I8 || mix_channels > 1)) return false;
    const int C = std::max(1, mix_channels);
    const double ratio = 48000 / rate;
    const double lp_alpha = 1 - std::exp(-(rate / 96000) * 2 * M_PI);   // :3251 / :3155
    const double decay = (rate / 96000) * 2 * M_PI;                     // a = exp(-decay)
    if (!(decay > 0) || !(ratio > 0)) return false;
    // what the samples can reach: a state of magnitude M still shows as M a^W after W outputs, so the warm-up grows with log M — int8 rows 128,
    // int32 rows 2^31 / full (garbage frames decode to anything an int32 holds); rows of doubles (values beyond int32) have no bound: not served
    if (rows_kind == TAIL_ROWS_F64) return false;
    const double mag = rows_kind == TAIL_ROWS_I8 ? 128.0 : std::max(1.0, std::ldexp(1.0, 31) / full);
    const double wd = std::ceil((37.0 + std::log(mag)) / decay);
    uint64_t max_nout = 0, sum_nout = 0;
    for (const TailJob &j : jobs) { max_nout = std::max<uint64_t>(max_nout, (uint64_t)std::max(j.nout, 0)); sum_nout += (uint64_t)std::max(j.nout, 0); }
    if (max_nout == 0) return true;
    const int E = (sum_nout / jobs.size() >= 6144 && kind == TAIL_QOA) ? 8 : 4;   // long jobs (QOA calls): 2048-output tiles; FLAC blocks: 1024
    const int T = 256 * E;
    if (wd > (double)T) return false;
    const int W = (int)wd;
    const double capd = std::ceil((double)(T + W) / ratio) + 8;
    if (capd > 32768) return false;
    const int cap = ((int)capd + 3) & ~3;
    const int xbn = ((T + W) + (T + W) / E + 2 + 3) & ~3;
    // F32 storage: the f32 interpolation (k_iir_tail_fast) — integer sample rates with at most 512 output phases
    bool fast = dtype == AUKIT_F32 && !ctx->exact_math && !getenv("AUKIT_NO_TAIL_FAST") && rate == std::floor(rate) && rate <= 4e9 && rows_kind != TAIL_ROWS_F64;
    unsigned long long fa = 0, fb = 0;
    if (fast) {
        unsigned long long x = 48000, y = (unsigned long long)rate;
        while (y) { const unsigned long long tq = x % y; x = y; y = tq; }
        fa = (unsigned long long)rate / x; fb = 48000 / x;   // x - 1 = (i - 1) / ratio = (i - 1) fa / fb
        if (fb == 1) { fa *= 2; fb = 2; }
        fast = fb <= 512 && ((double)max_nout * (double)fa + (double)fb) * (double)fb < 4294967296.0;
    }
    const int wf = interp == AUKIT_INTERP_CUBIC ? 4 : 1;
    const size_t lds = fast ? ((((size_t)fb * wf + 3) & ~(size_t)3) + (size_t)C * ((size_t)cap + (size_t)xbn)) * 4 : (size_t)C * ((size_t)cap + (size_t)xbn) * 8;
    if (lds > 150 * 1024) return false;
    std::vector<unsigned> tile_job, job_tile0(jobs.size());
    for (size_t k = 0; k < jobs.size(); k++) {
        job_tile0[k] = (unsigned)tile_job.size();
        const unsigned nt = (unsigned)(((uint64_t)std::max(jobs[k].nout, 0) + T - 1) / T);
        for (unsigned q = 0; q < nt; q++) tile_job.push_back((unsigned)k);
    }
    if (tile_job.empty()) return true;
    if (tile_job.size() > 0x7FFFFFFFull) return false;
    // one table: jobs | tile_job | job_tile0
    std::vector<float> w;
    if (fast) {
        w.resize((size_t)fb * wf);
        for (unsigned r = 0; r < fb; r++) {
            const long double f = (long double)r / (long double)fb, f2 = f * f, f3 = f2 * f;
            if (wf == 1) w[r] = (float)f;
            else {
                w[4 * r] = (float)(-0.5L * f3 + f2 - 0.5L * f); w[4 * r + 1] = (float)(1.5L * f3 - 2.5L * f2 + 1.0L);
                w[4 * r + 2] = (float)(-1.5L * f3 + 2.0L * f2 + 0.5L * f); w[4 * r + 3] = (float)(0.5L * f3 - 0.5L * f2);
            }
        }
    }
    const size_t jb = jobs.size() * sizeof(TailJob), tb = (tile_job.size() * 4 + 7) & ~(size_t)7, j0b = (job_tile0.size() * 4 + 15) & ~(size_t)15, wb = w.size() * 4;
    std::vector<unsigned char> tab(jb + tb + j0b + wb);
    memcpy(tab.data(), jobs.data(), jb);
    memcpy(tab.data() + jb, tile_job.data(), tile_job.size() * 4);
    memcpy(tab.data() + jb + tb, job_tile0.data(), job_tile0.size() * 4);
    if (wb) memcpy(tab.data() + jb + tb + j0b, w.data(), wb);
    if ((*rc = upload_table(ctx, ctx->seg_buf, tab.data(), tab.size()))) return true;
    TailParams P{};
    const unsigned char *d = reinterpret_cast<const unsigned char *>(ctx->seg_buf.p);
    P.jobs = reinterpret_cast<const TailJob *>(d);
    P.tile_job = reinterpret_cast<const unsigned *>(d + jb);
    P.job_tile0 = reinterpret_cast<const unsigned *>(d + jb + tb);
    P.n_tiles = (unsigned)tile_job.size();
    P.W = W; P.cap = cap; P.xbn = xbn; P.C = C;
    P.ratio = ratio; P.rcp = 1.0 / ratio;
    P.exact = exact_div_verified(ctx, ratio, std::max<uint64_t>(max_nout + 2, 1ull << 17)) ? 1 : 0;
    P.lp_alpha = lp_alpha; P.full = full;
    P.rows = rows; P.out = out;
    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
    const unsigned grid = (unsigned)std::min<uint64_t>(P.n_tiles, (uint64_t)ctx->num_cus * per_cu);
    if ((*rc = ctx_begin_kernel(ctx))) return true;
    if (fast) {
        P.fa = (unsigned)fa; P.fb = (unsigned)fb; P.fmagic = (unsigned)((4294967296ull + fb - 1) / fb);
        P.dq256 = (unsigned)((256ull * fa) / fb); P.dr256 = (unsigned)((256ull * fa) % fb);
        P.wg = reinterpret_cast<const float *>(d + jb + tb + j0b);
        if (kind == TAIL_QOA) { if (E == 8) tail_launch_fast<TAIL_QOA, 8, signed char>(interp, P, grid, lds, ctx->stream); else tail_launch_fast<TAIL_QOA, 4, signed char>(interp, P, grid, lds, ctx->stream); }
        else tail_launch_fast<TAIL_FLAC, 4, int>(interp, P, grid, lds, ctx->stream);
    } else
    if (kind == TAIL_QOA) { if (E == 8) tail_launch_out<TAIL_QOA, 8, signed char>(interp, dtype, P, grid, lds, ctx->stream); else tail_launch_out<TAIL_QOA, 4, signed char>(interp, dtype, P, grid, lds, ctx->stream); }
    else if (rows_kind == TAIL_ROWS_I32) tail_launch_out<TAIL_FLAC, 4, int>(interp, dtype, P, grid, lds, ctx->stream);
    else tail_launch_out<TAIL_FLAC, 4, double>(interp, dtype, P, grid, lds, ctx->stream);
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_iir_tail launch failed"); return true; }
    *rc = ctx_end_kernel(ctx, name, algorithmic_bytes);
    return true;
}

}  // namespace aukit
