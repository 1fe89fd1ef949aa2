// effects.hip — Audio:mono / Audio:mix / Audio:pcm and aukit.effects.* on device-resident audio batches.
//
// All of these are HBM-bound fp64 (or f32-stored) element-wise maps, reductions, or linear recurrences:
//   * maps (amplify, invert, fade, normalize-scale, delay, mono, mix, encodePCM): one coalesced pass;
//   * reductions (normalize peak, center mean): wave shuffles + LDS, atomicMax on the bit pattern;
//   * first-order recurrences (lowpass, highpass): affine-map scan — each thread owns 8 consecutive samples,
//     wave64 shuffle scan of the (A, B) composites, cross-wave scan in LDS, carry across tiles;
//   * lag-`samples` recurrences (echo, reverb combs / all-pass): the `samples` residue chains are independent,
//     so chains map to lanes (coalesced) and only the chain direction is sequential.
// Arithmetic is fp64 in the reference's operation order wherever the algorithm is a map (bit-identical to the
// Lua with AUKIT_F64 storage); the scans re-associate products and are tolerance-level (≤ 1e-12 observed).
#include <algorithm>
#include "common.h"

namespace aukit {

struct RowMeta {  // device view of an audio's layout
    const unsigned long long *len, *off, *stride;
    int channels;
    unsigned n;
};
static RowMeta meta_of(const aukit_audio *a) {
    RowMeta m;
    m.len = reinterpret_cast<const unsigned long long *>(a->d_meta);
    m.off = m.len + a->n;
    m.stride = m.len + 2 * (size_t)a->n;
    m.channels = a->channels;
    m.n = a->n;
    return m;
}
AUKIT_DEV void row_of(const RowMeta &m, unsigned r, unsigned long long *base, unsigned long long *len) {
    unsigned s = r / (unsigned)m.channels, c = r - s * (unsigned)m.channels;
    *base = m.off[s] + (unsigned long long)c * m.stride[s];
    *len = m.len[s];
}

enum MapOp { MAP_AMPLIFY, MAP_INVERT, MAP_FADE, MAP_SCALE_ROWMAX, MAP_DELAY };
struct MapArgs {
    double a0, a1, a2, a3;       // op-specific scalars
    const double *rowmax;        // MAP_SCALE_ROWMAX: max |x| per ROW; the stream's maximum (not `independent`) is taken over its rows here
    int independent;
    const void *aux;             // MAP_DELAY: copy of the original samples
    unsigned long long lag;
};

// effects.normalize's multiplier for row r from the per-row maxima: peak / max over the row (independent) or over every row of its stream;
// math.max(max, abs(x)) skips NaNs, so do the comparisons here (:3440-3451)
AUKIT_DEV double norm_mult(const double *rowmax, double peak, int independent, unsigned r, int channels) {
    if (independent) return peak / rowmax[r];
    const unsigned s0 = r / (unsigned)channels * (unsigned)channels;
    double mx = 0;
    for (int c = 0; c < channels; c++) { const double v = rowmax[s0 + c]; mx = mx < v ? v : mx; }
    return peak / mx;
}

template <typename T, int OP>
__global__ __launch_bounds__(256) void k_map(T *data, RowMeta m, MapArgs A) {
    const unsigned r = blockIdx.y;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    T *row = data + base;
    double mult = 0;
    if constexpr (OP == MAP_SCALE_ROWMAX) mult = norm_mult(A.rowmax, A.a0, A.independent, r, m.channels);  // peak / max  :3444
    auto op = [&](unsigned long long i, T cur) -> T {
        const double x = (double)cur;
        if constexpr (OP == MAP_AMPLIFY) return (T)lua_clamp(x * A.a0, -1, 1);                       // :3365
        else if constexpr (OP == MAP_INVERT) return (T)(-x);                                           // :3421
        else if constexpr (OP == MAP_SCALE_ROWMAX) return (T)lua_clamp(x * mult, -1, 1);              // :3455
        else if constexpr (OP == MAP_FADE) {  // Lua index li = i + 1 in [start, limit]  :3406-3408
            const double li = (double)(i + 1);
            return (li >= A.a0 && li <= A.a1) ? (T)lua_clamp(x * (A.a2 * (li - A.a0) + A.a3), -1, 1) : cur;
        } else {                              // MAP_DELAY: o[i] = clamp(o[i] + original[i - samples] * multiplier)  :3514
            return i >= A.lag ? (T)lua_clamp(x + (double)reinterpret_cast<const T *>(A.aux)[base + i - A.lag] * A.a0, -1, 1) : cur;
        }
    };
    // 16 bytes per lane and access (rows start on 64-byte boundaries)
    constexpr int PV = 16 / (int)sizeof(T);
    typedef T tvp __attribute__((ext_vector_type(PV), aligned(16)));
    const unsigned long long groups = len / PV;
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (unsigned long long)gridDim.x * 256) {
        tvp w = *reinterpret_cast<const tvp *>(row + PV * g);
#pragma unroll
        for (int e = 0; e < PV; e++) w[e] = op(PV * g + e, w[e]);
        *reinterpret_cast<tvp *>(row + PV * g) = w;
    }
    if (blockIdx.x == 0)
        for (unsigned long long i = groups * PV + threadIdx.x; i < len; i += 256) row[i] = op(i, row[i]);
}

// max |x| per row → atomicMax on the (non-negative) bit pattern; NaNs are skipped like math.max does
template <typename T>
__global__ __launch_bounds__(256) void k_rowmax(const T *data, RowMeta m, unsigned long long *out) {
    const unsigned r = blockIdx.y;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    const T *row = data + base;
    double mx = 0;
    // 16 bytes per lane and load (rows start on 64-byte boundaries); one element per load moved 1.3 TB/s of this read-only pass
    constexpr int PV = 16 / (int)sizeof(T);
    typedef T tvp __attribute__((ext_vector_type(PV), aligned(16)));
    const unsigned long long groups = len / PV;
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (unsigned long long)gridDim.x * 256) {
        const tvp w = *reinterpret_cast<const tvp *>(row + PV * g);
#pragma unroll
        for (int e = 0; e < PV; e++) { const double v = fabs((double)w[e]); mx = mx < v ? v : mx; }
    }
    if (blockIdx.x == 0)
        for (unsigned long long i = groups * PV + threadIdx.x; i < len; i += 256) { const double v = fabs((double)row[i]); mx = mx < v ? v : mx; }
    __shared__ double red[4];
    for (int o = 32; o; o >>= 1) { double t = __shfl_xor(mx, o); mx = mx < t ? t : mx; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per workgroup (per-wave atomics on ~N addresses serialised: 13.6 ms → see profiles/)
        for (int w = 1; w < 4; w++) mx = mx < red[w] ? red[w] : mx;
        atomicMax(out + r, (unsigned long long)__double_as_longlong(mx));
    }
}

// Audio:mono  :682-687  (s = 0 + c1 + c2 ...; s / cn)
// NORM: the input carries a deferred effects.normalize (aukit_audio::pend_norm): every sample is read as the value the map would have
// stored, (T)clamp(x * mult, -1, 1) — the same bits as k_map<normalize> followed by the plain kernel, one pass over the rows less
template <typename T, bool NORM>
__global__ __launch_bounds__(256) void k_mono(const T *in, RowMeta mi, T *out, RowMeta mo, const double *rowmax, double peak, int independent) {
    const unsigned s = blockIdx.y;
    const unsigned long long len = mi.len[s], ib = mi.off[s], st = mi.stride[s], ob = mo.off[s];
    double mult[AUKIT_MAX_CHANNELS];
    if constexpr (NORM)
        for (int c = 0; c < mi.channels; c++) mult[c] = norm_mult(rowmax, peak, independent, s * (unsigned)mi.channels + c, mi.channels);
    // 16 bytes per thread and channel (rows start on 64-byte boundaries and are padded to 16 elements, audio_prepare: whole vectors are
    // readable and writable); the scalar version of round 1 ran at 4.0 TB/s
    constexpr int V = 16 / (int)sizeof(T);
    const unsigned long long nvec = (len + V - 1) / V;
    for (unsigned long long v = (unsigned long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (unsigned long long)gridDim.x * 256) {
        double acc[V];
#pragma unroll
        for (int k = 0; k < V; k++) acc[k] = 0;
        for (int c = 0; c < mi.channels; c++) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(in + ib + (unsigned long long)c * st + v * V);
            const T *xv = reinterpret_cast<const T *>(&raw);
#pragma unroll
            for (int k = 0; k < V; k++) {
                double x = (double)xv[k];
                if constexpr (NORM) x = (double)(T)lua_clamp(x * mult[c], -1, 1);  // :3455
                acc[k] = acc[k] + x;
            }
        }
        T o[V];
#pragma unroll
        for (int k = 0; k < V; k++) o[k] = (T)(acc[k] / mi.channels);
        *reinterpret_cast<uint4 *>(out + ob + v * V) = *reinterpret_cast<const uint4 *>(o);
    }
}

// Audio:mix  :823-833
struct MixSrc { const void *data[8]; RowMeta m[8]; int count; };
template <typename T>
__global__ __launch_bounds__(256) void k_mix(MixSrc S, T *out, RowMeta mo, double amplifier) {
    const unsigned r = blockIdx.y;
    const unsigned s = r / (unsigned)mo.channels, c = r - s * (unsigned)mo.channels;
    const unsigned long long len = mo.len[s], ob = mo.off[s] + (unsigned long long)c * mo.stride[s];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < len; i += (unsigned long long)gridDim.x * 256) {
        double acc = 0;
        for (int a = 0; a < S.count; a++)
            if ((int)c < S.m[a].channels) {
                const RowMeta &m = S.m[a];
                acc = acc + (i < m.len[s] ? (double)reinterpret_cast<const T *>(S.data[a])[m.off[s] + (unsigned long long)c * m.stride[s] + i] : 0.0);
            }
        out[ob + i] = (T)lua_clamp(acc * amplifier, -1, 1);
    }
}

// the same for any number of audios (the reference sums whatever `...` holds, :804-835): the sources' table lies in device memory
struct MixEnt { const void *data; RowMeta m; };
template <typename T>
__global__ __launch_bounds__(256) void k_mix_many(const MixEnt *ents, int count, T *out, RowMeta mo, double amplifier) {
    const unsigned r = blockIdx.y;
    const unsigned s = r / (unsigned)mo.channels, c = r - s * (unsigned)mo.channels;
    const unsigned long long len = mo.len[s], ob = mo.off[s] + (unsigned long long)c * mo.stride[s];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < len; i += (unsigned long long)gridDim.x * 256) {
        double acc = 0;
        for (int a = 0; a < count; a++) {   // (in the order of the argument list: the sum is the reference's, term by term)
            const MixEnt e = ents[a];
            if ((int)c < e.m.channels)
                acc = acc + (i < e.m.len[s] ? (double)reinterpret_cast<const T *>(e.data)[e.m.off[s] + (unsigned long long)c * e.m.stride[s] + i] : 0.0);
        }
        out[ob + i] = (T)lua_clamp(acc * amplifier, -1, 1);
    }
}

// encodePCM  :874-892 : d * (d < 0 and maxValue or maxValue-1) + add, laid out interleaved or channel-after-channel
template <typename T>
__global__ __launch_bounds__(256) void k_encode_pcm(const T *in, RowMeta mi, double *out, RowMeta mo, double maxValue, double add, int is_float, int interleaved) {
    const unsigned s = blockIdx.y;
    const unsigned long long len = mi.len[s], ib = mi.off[s], st = mi.stride[s], ob = mo.off[s];
    const int nc = mi.channels;
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x; k < len * nc; k += (unsigned long long)gridDim.x * 256) {
        unsigned long long i, c;
        if (interleaved) { i = k / nc; c = k - i * nc; } else { c = k / len; i = k - c * len; }
        double d = (double)in[ib + c * st + i];
        out[ob + k] = is_float ? d : d * (d < 0 ? maxValue : maxValue - 1) + add;
    }
}

// effects.center  :3468-3474 : per 1 s window, subtract the mean
template <typename T>
__global__ __launch_bounds__(256) void k_center(T *data, RowMeta m, unsigned long long win) {
    __shared__ double red[4];
    const unsigned r = blockIdx.y;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    T *row = data + base;
    const unsigned long long nwin = (len + win - 1) / win;
    for (unsigned long long w = blockIdx.x; w < nwin; w += gridDim.x) {
        const unsigned long long i0 = w * win, l = (len - i0) < win ? (len - i0) : win;
        double acc = 0;
        for (unsigned long long j = threadIdx.x; j < l; j += 256) acc += (double)row[i0 + j];
        for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        const double avg = (red[0] + red[1] + red[2] + red[3]) / (double)l;
        for (unsigned long long j = threadIdx.x; j < l; j += 256) row[i0 + j] = (T)lua_clamp((double)row[i0 + j] - avg, -1, 1);
    }
}

// ---------------------------------------------------------------- first-order recurrences (lowpass / highpass)
// y_out = A * y_in + B composites; compose(f then g) = (gA * fA, gA * fB + gB)
struct Aff { double A, B; };
AUKIT_DEV Aff aff_then(const Aff &f, const Aff &g) { return Aff{g.A * f.A, __builtin_fma(g.A, f.B, g.B)}; }
// an affine map moved between lanes by DPP: lanes without a source lane, and rows outside ROW_MASK, receive the identity {1, 0}
// (row_shr:n = 0x110 + n, row_bcast15 = 0x142, row_bcast31 = 0x143, wave_shr:1 = 0x138)
template <int CTRL, int ROW_MASK = 0xF>
AUKIT_DEV Aff aff_dpp(const Aff &v) {
    const long long a = __double_as_longlong(v.A), b = __double_as_longlong(v.B);
    const int alo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)a, CTRL, ROW_MASK, 0xF, false);
    const int ahi = __builtin_amdgcn_update_dpp(0x3FF00000, (int)(a >> 32), CTRL, ROW_MASK, 0xF, false);
    const int blo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, false);
    const int bhi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return Aff{__longlong_as_double(((long long)ahi << 32) | (unsigned)alo), __longlong_as_double(((long long)bhi << 32) | (unsigned)blo)};
}

// One workgroup per row, 64 bytes per thread and tile (rows start on 64-byte boundaries and are padded to 16 elements,
// audio_prepare).  Per tile: every thread composes the affine maps of its own samples, a wave scan + one LDS exchange give the
// value entering each thread, the recurrence is then run for real in the reference's operation order.  The next tile's vector
// loads are issued before the scan, and carries / wave totals are double-buffered so that one barrier per tile is enough
// (round 1: scalar strided accesses, no prefetch, 4 barriers per tile — 1.9 TB/s; see profiles/).
template <typename T, bool HIGHPASS>
__global__ __launch_bounds__(256) void k_onepole(T *data, RowMeta m, double a, unsigned long long *rowmax) {
    constexpr int PER = 64 / (int)sizeof(T), TILE = 256 * PER, NV = PER * (int)sizeof(T) / 16;
    __shared__ Aff wave_tot[2][4];
    __shared__ double carry_y[2], carry_x[2];
    const unsigned r = blockIdx.x;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    T *row = data + base;
    if (len < 2) {  // nothing to filter (:3590, :3608 start at the second sample); the row's maximum is still owed
        if (threadIdx.x == 0) {
            double v = len ? fabs((double)row[0]) : 0.0;
            if (!(v == v)) v = 0;
            rowmax[r] = (unsigned long long)__double_as_longlong(v);
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry_y[0] = 0; carry_x[0] = 0; }
    double mx = 0;  // max |stored value| over this thread's samples: effects.normalize's peak search (:3440-3442), NaNs skipped like math.max
    uint4 raw[NV], nraw[NV];
    T xb = 0, nxb = 0;  // HIGHPASS: the ORIGINAL sample before this thread's first one, for the first lane of waves 1..3
    auto fetch = [&](unsigned long long t0, uint4 (&dst)[NV], T &before) {
        const unsigned long long i0 = t0 + (unsigned long long)threadIdx.x * PER;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            dst[v] = make_uint4(0, 0, 0, 0);
            if (i0 + (unsigned long long)v * (16 / sizeof(T)) < len) dst[v] = *reinterpret_cast<const uint4 *>(row + i0 + v * (16 / sizeof(T)));  // padded rows: whole vectors are readable
        }
        if (HIGHPASS && lane == 0 && wave > 0 && i0 < len) before = row[i0 - 1];
    };
    fetch(0, raw, xb);
    int ph = 0;
    for (unsigned long long t0 = 0; t0 < len; t0 += TILE, ph ^= 1) {
        const unsigned long long i0 = t0 + (unsigned long long)threadIdx.x * PER;
        double x[PER];
        const T *rt = reinterpret_cast<const T *>(raw);
#pragma unroll
        for (int k = 0; k < PER; k++) x[k] = (double)rt[k];
        const int cnt = i0 >= len ? 0 : (int)(len - i0 < (unsigned long long)PER ? len - i0 : PER);
        const double xb_d = (double)xb;
        if (t0 + TILE < len) fetch(t0 + TILE, nraw, nxb);  // in flight during the scan
        const unsigned long long tile_last = (t0 + TILE <= len ? t0 + TILE : len) - 1;
        double xprev = 0;
        if (HIGHPASS) {
            const double up = __shfl_up(x[PER - 1], 1);
            xprev = lane > 0 ? up : (wave > 0 ? xb_d : carry_x[ph]);  // thread 0: the previous tile's last ORIGINAL sample (LDS: its owner has stored over it)
        }
        // local composite of this thread's samples; sample 0 of the row is the constant map y = x  (:3589, :3607)
        Aff f{1.0, 0.0};
        double xp = xprev;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (k < cnt) {
                Aff g = HIGHPASS ? Aff{a, a * (x[k] - xp)} : Aff{1.0 - a, a * x[k]};
                if (i0 + k == 0) g = Aff{0.0, x[k]};
                f = aff_then(f, g);
                xp = x[k];
            }
        }
        if (HIGHPASS && cnt && i0 + cnt - 1 == tile_last) carry_x[ph ^ 1] = xp;  // xp = this thread's last ORIGINAL sample
        // the wave scan on DPP moves (round 3: VALU latency instead of four LDS crossbar trips per step; a lane without a source — and the rows
        // a row_bcast does not write — receive the identity map, so no step needs a select): four steps inside rows of 16, then lane 15 / 31
        Aff inc = f;
        inc = aff_then(aff_dpp<0x111>(inc), inc);
        inc = aff_then(aff_dpp<0x112>(inc), inc);
        inc = aff_then(aff_dpp<0x114>(inc), inc);
        inc = aff_then(aff_dpp<0x118>(inc), inc);
        inc = aff_then(aff_dpp<0x142, 0xA>(inc), inc);
        inc = aff_then(aff_dpp<0x143, 0xC>(inc), inc);
        if (lane == 63) wave_tot[ph][wave] = inc;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only: the prefetched vectors stay in flight across the barrier
        __builtin_amdgcn_s_barrier();        // also publishes the carries written during the previous tile
        Aff pre{1.0, 0.0};
        for (int w = 0; w < wave; w++) pre = aff_then(pre, wave_tot[ph][w]);
        pre = aff_then(pre, aff_dpp<0x138>(inc));   // the lane before (lane 0: the identity)
        double y = __builtin_fma(pre.A, carry_y[ph], pre.B);  // value entering this thread's first sample
        xp = xprev;
        T out[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (i0 + k == 0) y = x[k];
            else if (HIGHPASS) y = a * (y + x[k] - xp);  // d[i] = a * (d[i-1] + llx - lx)  :3613
            else y = y + a * (x[k] - y);                 // d[i] = l + a * (d[i] - l)      :3594
            xp = x[k];
            out[k] = (T)y;
            if (k < cnt) { const double v = fabs((double)out[k]); mx = mx < v ? v : mx; }
            if (k == cnt - 1 && i0 + k == tile_last) carry_y[ph ^ 1] = y;
        }
        const uint4 *ov = reinterpret_cast<const uint4 *>(out);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const unsigned long long e = i0 + (unsigned long long)v * (16 / sizeof(T));
            if (e + 16 / sizeof(T) <= len) *reinterpret_cast<uint4 *>(row + e) = ov[v];
            else
                for (int q = 0; q < (int)(16 / sizeof(T)); q++) if (e + q < len) row[e + q] = out[v * (16 / sizeof(T)) + q];
        }
#pragma unroll
        for (int v = 0; v < NV; v++) raw[v] = nraw[v];
        xb = nxb;
    }
    __shared__ double redmx[4];
    for (int o = 32; o; o >>= 1) { const double t = __shfl_xor(mx, o); mx = mx < t ? t : mx; }
    if (lane == 0) redmx[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) mx = mx < redmx[w] ? redmx[w] : mx;
        rowmax[r] = (unsigned long long)__double_as_longlong(mx);
    }
}

// ---------------------------------------------------------------- lag recurrences
// effects.echo :3531  o[i] = clamp(o[i] + o[i - L] * mult)  — chain r: i = r + k*L
template <typename T>
__global__ __launch_bounds__(256) void k_echo(T *data, RowMeta m, unsigned long long L, double mult) {
    const unsigned r = blockIdx.y;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    T *row = data + base;
    for (unsigned long long c = (unsigned long long)blockIdx.x * 256 + threadIdx.x; c < L && c < len; c += (unsigned long long)gridDim.x * 256) {
        double prev = (double)row[c];
        for (unsigned long long i = c + L; i < len; i += L) {
            prev = lua_clamp((double)row[i] + prev * mult, -1, 1);
            row[i] = (T)prev;
        }
    }
}
// one reverb comb :3560-3568, accumulated into sum in comb order
template <typename T>
__global__ __launch_bounds__(256) void k_comb(const T *data, RowMeta m, double *sum, unsigned long long L, double mult, int first) {
    const unsigned r = blockIdx.y;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    const T *row = data + base;
    double *srow = sum + base;
    for (unsigned long long c = (unsigned long long)blockIdx.x * 256 + threadIdx.x; c < L && c < len; c += (unsigned long long)gridDim.x * 256) {
        double comb = (double)row[c];
        srow[c] = (first ? 0.0 : srow[c]) + comb;
        for (unsigned long long i = c + L; i < len; i += L) {
            comb = (double)row[i] + comb * mult;
            srow[i] = (first ? 0.0 : srow[i]) + comb;
        }
    }
}
// wet/dry mix :3571
template <typename T>
__global__ __launch_bounds__(256) void k_wetdry(const T *data, RowMeta m, double *sum, double wet, double dry) {
    const unsigned r = blockIdx.y;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < len; i += (unsigned long long)gridDim.x * 256)
        sum[base + i] = sum[base + i] * wet + (double)data[base + i] * dry;
}
// in-place all-pass :3574-3575: sum[i] = sum[i] - 0.131*sum[i-S] + 0.131*sum[i+20-S], i >= S+2 (1-based).
// Element i depends on i-S and i-S+20, both at least S-20 behind: blocks of S-20 elements are internally independent
// (S == 20: the second one is the element itself, blocks of S).
__global__ __launch_bounds__(1024) void k_allpass(double *sum, RowMeta m, long long S) {
    const unsigned r = blockIdx.x;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    double *s = sum + base;  // s[i-1] = sum[i]
    const long long n = (long long)len, W = S > 20 ? S - 20 : S;  // S == 20 (rates of 224.1 … 235.2 Hz): sum[i + 20 - S] is sum[i] itself, only i - S lies behind
    // Only this workgroup touches the row, and __syncthreads() orders its global stores for its own threads.  (With an agent-scope
    // __threadfence() in front of every barrier each of the ~110 block steps wrote the whole L2 back: 119 ms for 2048 rows of ten seconds.)
    if (threadIdx.x == 0) s[S] = s[S] - 0.131 * s[0];  // sum[S+1] -= 0.131 * sum[1]
    __syncthreads();
    for (long long b0 = S + 2; b0 <= n; b0 += W) {
        const long long b1 = (b0 + W - 1 < n) ? b0 + W - 1 : n;
        for (long long i = b0 + threadIdx.x; i <= b1; i += blockDim.x) s[i - 1] = s[i - 1] - 0.131 * s[i - S - 1] + 0.131 * s[i + 20 - S - 1];
        __syncthreads();
    }
}
// :3576-3577
template <typename T>
__global__ __launch_bounds__(256) void k_allpass_out(T *data, RowMeta m, const double *sum, long long S) {
    const unsigned r = blockIdx.y;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    const double *s = sum + base;
    for (long long i = S + 1 + (long long)blockIdx.x * 256 + threadIdx.x; i <= (long long)len; i += (long long)gridDim.x * 256) {
        double v = (i == S + 1) ? s[S] - 0.131 * s[0] : s[i - 1] - 0.131 * s[i - S - 1] + 0.131 * s[i + 20 - S - 1];
        data[base + i - 1] = (T)lua_clamp(v, -1, 1);
    }
}


// effects.reverb :3546-3580 on an F32 audio in ONE pass over the row (round 3; the seven launches above moved ~150 B per sample through a
// scratch of doubles: 25.8 ms for 1024 ten-second stereo streams).  Everything in the effect is a lag recurrence:
//   comb_k[i] = o[i] + mult_k comb_k[i - L_k]                                  (four lags L_k around `delay`)
//   m[i]      = ((((0 + comb_1) + comb_2) + comb_3) + comb_4)[i] wet + o[i] dry
//   p[i]      = m[i] - 0.131 p[i - S] + 0.131 p[i + 20 - S]   (i >= S + 2; the in-place all-pass reads entries it has already rewritten)
//   o[i]      = clamp(p[i] - 0.131 p[i - S] + 0.131 p[i + 20 - S])             (the same two taps once more)
// so a block of W <= min(L_k, S - 20) consecutive samples depends only on samples before the block.  One workgroup of 1024 threads owns a
// row and walks it block by block; the only state is the last L_k values of each comb and the last S values of p, and those live in LDS
// as rings (slot = index mod lag: an element reads the slot of its predecessor one lag back and overwrites it with itself): four f32
// comb rings + one fp64 ring for p, 111 KiB at 48 kHz with the default delay — one workgroup per CU.  HBM sees each sample once in, once out.
// The comb history is kept in f32 (the storage type of the audio; with fp64 rings the state would not fit a CU): every stored value
// re-enters scaled by mult_k < 1, so the rounding stays a few 1e-8 — the F32 contract is 1e-6 RMS; F64 audios keep the launches above.
struct ReverbParams {
    unsigned L[4];      // comb lags
    unsigned S;         // all-pass lag
    unsigned ept;       // elements per thread and block (block = ept * 1024 samples)
    double mult[4];
    double wet, dry;
};
__global__ __launch_bounds__(1024) void k_reverb_f32(float *data, RowMeta m, const ReverbParams R) {
    extern __shared__ double rv_sm[];
    const unsigned r = blockIdx.x, tid = threadIdx.x;
    unsigned long long base, len;
    row_of(m, r, &base, &len);
    float *row = data + base;
    double *const pr = rv_sm;                                   // p ring, S doubles
    float *cr[4];
    cr[0] = reinterpret_cast<float *>(rv_sm + ((R.S + 1) & ~1u));
    for (int k = 1; k < 4; k++) cr[k] = cr[k - 1] + R.L[k - 1];
    // (history before the row: the combs start from o itself (:3561-3564: 0 * mult is 0), p is not read before index S + 1)
    for (unsigned i = tid; i < R.S; i += 1024) pr[i] = 0.0;
    for (int k = 0; k < 4; k++) for (unsigned i = tid; i < R.L[k]; i += 1024) cr[k][i] = 0.f;
    unsigned pos[4] = {tid, tid, tid, tid}, ps = tid;          // index mod lag; every lag is at least 1024
    __syncthreads();
    constexpr int EMAX = 8;
    for (unsigned long long b0 = 0; b0 < len; b0 += (unsigned long long)R.ept * 1024) {
        double A[EMAX], Bv[EMAX];
        float x[EMAX];
        // ---- what the block needs from before it: the row itself and the two all-pass taps (read before anyone overwrites them)
        {
            unsigned q = ps;
#pragma unroll
            for (int e = 0; e < EMAX; e++) {
                if (e < (int)R.ept) {
                    const unsigned long long i0 = b0 + (unsigned)e * 1024 + tid;
                    x[e] = i0 < len ? row[i0] : 0.f;
                    unsigned qb = q + 20; qb -= qb >= R.S ? R.S : 0u;
                    A[e] = pr[q];      // p[i - S]
                    Bv[e] = pr[qb];    // p[i + 20 - S]
                    q += 1024; q -= q >= R.S ? R.S : 0u;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EMAX; e++) {
            if (e < (int)R.ept) {
                const unsigned long long i0 = b0 + (unsigned)e * 1024 + tid;   // 0-based; the Lua index is i0 + 1
                const double xv = (double)x[e];
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double c = xv + (double)cr[k][pos[k]] * R.mult[k];   // :3566 (:3562 while the ring still holds zeros)
                    if (i0 < len) cr[k][pos[k]] = (float)c;
                    acc = acc + c;                                             // :3563 / :3568, in comb order
                    pos[k] += 1024; pos[k] -= pos[k] >= R.L[k] ? R.L[k] : 0u;
                }
                const double mv = acc * R.wet + xv * R.dry;                    // :3571
                double pv = mv;
                if (i0 == R.S) pv = mv - 0.131 * A[e];                          // :3574
                else if (i0 > R.S) pv = mv - 0.131 * A[e] + 0.131 * Bv[e];     // :3575
                if (i0 < len) {
                    pr[ps] = pv;
                    if (i0 == R.S) row[i0] = (float)lua_clamp(pv - 0.131 * A[e], -1, 1);                       // :3576
                    else if (i0 > R.S) row[i0] = (float)lua_clamp(pv - 0.131 * A[e] + 0.131 * Bv[e], -1, 1);   // :3577
                }
                ps += 1024; ps -= ps >= R.S ? R.S : 0u;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- host side
static unsigned xblocks(const aukit_audio *a, unsigned per_thread = 4) {
    uint64_t mx = 1;
    for (uint64_t l : a->len) mx = std::max(mx, l);
    uint64_t b = (mx + 256ull * per_thread - 1) / (256ull * per_thread);
    return (unsigned)std::min<uint64_t>(std::max<uint64_t>(b, 1), 1024);
}
static uint64_t audio_bytes(const aukit_audio *a) {
    uint64_t e = 0;
    for (uint64_t l : a->len) e += l * (uint64_t)a->channels;
    return e * dtype_size(a->dtype);
}
#define AUKIT_FLOAT_ONLY(a)                                                                              \
    if ((a)->dtype != AUKIT_F64 && (a)->dtype != AUKIT_F32) return fail(AUKIT_E_ARG, "this operation needs an AUKIT_F64 / AUKIT_F32 audio")
#define AUKIT_DISPATCH_T(a, CALL)                        \
    do {                                                 \
        if ((a)->dtype == AUKIT_F64) { using T = double; CALL; } \
        else { using T = float; CALL; }                  \
    } while (0)

template <int OP>
static int run_map(aukit_ctx *ctx, aukit_audio *a, const MapArgs &A, const char *name) {
    if (a->n == 0) return AUKIT_OK;
    dim3 grid(xblocks(a), a->n * a->channels);
    int rc = ctx_begin_kernel(ctx);
    if (rc) return rc;
    AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_map<T, OP>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<T *>(a->dev), meta_of(a), A));
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, name, 2 * audio_bytes(a));
}

static int fx_onepole(aukit_ctx *ctx, aukit_audio *a, double coef, bool highpass);

// everything owed on the rows themselves: a resample (flac_tail.hip), and behind it a deferred effects.highpass / lowpass (aukit_audio::lazy_fx) —
// both in one pass where k_rs_onepole serves the shape, else the ordinary resample kernel and then the ordinary filter
int lazy_resolve(aukit_ctx *ctx, aukit_audio *a) {
    if (!a->lazy_rs) return AUKIT_OK;
    if (a->lazy_fx) {
        aukit_ctx *use = ctx ? ctx : (ctx_is_live(a->lazy_ctx, a->lazy_ctx_id) ? a->lazy_ctx : nullptr);
        if (!use) return fail(AUKIT_E_ARG, "audio has a deferred filter and no context to run it with");
        const int fx = a->lazy_fx;
        const double coef = a->lazy_fx_coef;
        int lrc = AUKIT_OK;
        if (lazy_onepole_try(use, a, coef, fx == 2, &lrc)) return lrc;
        a->lazy_fx = 0;
        if ((lrc = lazy_materialize(use, a))) return lrc;
        a->rowmax_valid = false;
        return fx_onepole(use, a, coef, fx == 2);
    }
    return lazy_materialize(ctx, a);
}

int audio_flush(aukit_ctx *ctx, const aukit_audio *ca) {
    aukit_audio *a = const_cast<aukit_audio *>(ca);
    if (a && a->lazy_rs) { int lrc = lazy_resolve(ctx, a); if (lrc) return lrc; }   // an owed resample (and filter) first (flac_tail.hip)
    if (!a || !a->pend_norm) return AUKIT_OK;
    if (!ctx) ctx = ctx_is_live(a->pend_ctx, a->pend_ctx_id) ? a->pend_ctx : nullptr;
    if (!ctx) return fail(AUKIT_E_ARG, "audio has a deferred map and no context to apply it with");
    { int orc = owner_ready(ctx, a->pend_ctx, a->pend_ctx_id); if (orc) return orc; }   // the pass that left the rows and their maxima ran on pend_ctx's stream
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    a->pend_norm = false;
    MapArgs A{};
    A.a0 = a->pend_peak;
    A.rowmax = reinterpret_cast<const double *>(a->d_rowmax);
    A.independent = a->pend_independent;
    a->rowmax_valid = false;  // the maxima described the samples before the map
    return run_map<MAP_SCALE_ROWMAX>(ctx, a, A, "k_map<normalize>");
}

// effects.normalize  :3430-3458.  Two things make config 5's tail (highpass → normalize → mono) three passes over the rows instead of five:
// the peak search reuses the per-row maxima the pass before it left behind (k_onepole; had_rowmax), and the scaling itself is deferred
// (aukit_audio::pend_norm) to whoever reads the samples next — Audio:mono applies it while it reads, anything else materialises it first.
static int fx_normalize(aukit_ctx *ctx, aukit_audio *a, double peak, int independent, bool had_rowmax) {
    if (a->n == 0) return AUKIT_OK;
    int rc = audio_rowmax_ensure(a);
    if (rc) return rc;
    if (!had_rowmax || getenv("AUKIT_NO_TAIL_FUSION")) {
        const size_t cnt = (size_t)a->n * a->channels;
        AUKIT_HIP_CHECK(hipMemsetAsync(a->d_rowmax, 0, cnt * 8, ctx->stream));
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        dim3 grid(xblocks(a, 32), a->n * a->channels);
        AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_rowmax<T>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const T *>(a->dev), meta_of(a),
                                               reinterpret_cast<unsigned long long *>(a->d_rowmax)));
        AUKIT_HIP_CHECK(hipGetLastError());
        if ((rc = ctx_end_kernel(ctx, "k_rowmax", audio_bytes(a)))) return rc;
    }
    a->pend_norm = true;
    a->pend_peak = peak;
    a->pend_independent = independent;
    a->pend_ctx = ctx; a->pend_ctx_id = ctx->id;
    if (getenv("AUKIT_NO_TAIL_FUSION")) return audio_flush(ctx, a);  // A/B: five passes
    return AUKIT_OK;
}

static int fx_onepole(aukit_ctx *ctx, aukit_audio *a, double coef, bool highpass) {
    if (a->n == 0) return AUKIT_OK;
    int rc = audio_rowmax_ensure(a);
    if (rc) return rc;
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    dim3 grid(a->n * a->channels);
    unsigned long long *rm = reinterpret_cast<unsigned long long *>(a->d_rowmax);
    if (highpass) AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_onepole<T, true>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<T *>(a->dev), meta_of(a), coef, rm));
    else AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_onepole<T, false>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<T *>(a->dev), meta_of(a), coef, rm));
    AUKIT_HIP_CHECK(hipGetLastError());
    a->rowmax_valid = true;  // max |x| of every row as stored: a following effects.normalize needs no pass of its own to find its peak
    return ctx_end_kernel(ctx, highpass ? "k_onepole<highpass>" : "k_onepole<lowpass>", 2 * audio_bytes(a));
}

static int fx_reverb(aukit_ctx *ctx, aukit_audio *a, double delay, double decay, double wet, double dry) {
    static const double dshift[4] = {0, -11.73, 19.31, -7.97}, cshift[4] = {0, 0.1313, 0.2743, 0.31};  // :3536-3537
    if (a->n == 0) return AUKIT_OK;
    const long long S = (long long)std::floor(0.08927 * a->rate);  // :3573
    long long lag[4];
    for (int k = 0; k < 4; k++) {
        double sf = std::floor((delay + dshift[k]) / 1000 * a->rate);
        if (sf < 1) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        lag[k] = (long long)sf;
    }
    for (uint64_t l : a->len)
        if ((long long)l < S + 1 || S < 20) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
    int rc;
    if (a->dtype == AUKIT_F32 && !getenv("AUKIT_NO_FUSED_REVERB")) {  // one pass, state in LDS (k_reverb_f32)
        const long long minlag = std::min(std::min(std::min(lag[0], lag[1]), std::min(lag[2], lag[3])), S - 20);
        const size_t lds = (((size_t)S + 1) & ~(size_t)1) * 8 + (size_t)(lag[0] + lag[1] + lag[2] + lag[3]) * 4;
        if (minlag >= 1024 && lds <= 150 * 1024 && S < 0x7FFFFFFFll) {
            ReverbParams R;
            for (int k = 0; k < 4; k++) { R.L[k] = (unsigned)lag[k]; R.mult[k] = decay - cshift[k]; }
            R.S = (unsigned)S;
            R.ept = (unsigned)std::min<long long>(minlag / 1024, 8);
            R.wet = wet; R.dry = dry;
            static thread_local int attr_dev = -1;
            if (attr_dev != ctx->device) {
                AUKIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_reverb_f32), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                attr_dev = ctx->device;
            }
            if ((rc = ctx_begin_kernel(ctx))) return rc;
            hipLaunchKernelGGL(k_reverb_f32, dim3(a->n * a->channels), dim3(1024), lds, ctx->stream, reinterpret_cast<float *>(a->dev), meta_of(a), R);
            AUKIT_HIP_CHECK(hipGetLastError());
            return ctx_end_kernel(ctx, "k_reverb_f32", 2 * audio_bytes(a));
        }
    }
    rc = ctx->tmp_buf.ensure((size_t)a->total * 8);
    if (rc) return rc;
    double *sum = reinterpret_cast<double *>(ctx->tmp_buf.p);
    dim3 grid(xblocks(a), a->n * a->channels);
    for (int k = 0; k < 4; k++) {
        AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_comb<T>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const T *>(a->dev), meta_of(a), sum,
                                               (unsigned long long)lag[k], decay - cshift[k], k == 0 ? 1 : 0));
        AUKIT_HIP_CHECK(hipGetLastError());
    }
    AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_wetdry<T>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const T *>(a->dev), meta_of(a), sum, wet, dry));
    hipLaunchKernelGGL(k_allpass, dim3(a->n * a->channels), dim3(1024), 0, ctx->stream, sum, meta_of(a), S);
    AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_allpass_out<T>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<T *>(a->dev), meta_of(a), sum, S));
    AUKIT_HIP_CHECK(hipGetLastError());
    ctx->last_kernel = "reverb(k_comb x4, k_wetdry, k_allpass, k_allpass_out)";
    return AUKIT_OK;
}

}  // namespace aukit

using namespace aukit;

extern "C" {

int aukit_mono(aukit_ctx *ctx, const aukit_audio *in, aukit_audio **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_FLOAT_ONLY(in);
    if (*out == in) return fail(AUKIT_E_ARG, "mono cannot run in place");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    if (in->lazy_rs && in->lazy_fx && in->channels == 2 && in->dtype == AUKIT_F32 && !(in->pend_norm && in->pend_independent) && in->n && !getenv("AUKIT_NO_MONO_FUSION")) {
        // resample -> highpass / lowpass [-> normalize] -> mono with everything before the mono still owed (BASELINE config 5's tail): one pass from the
        // decoder's rows to the MEAN of the filtered channels — the stereo rows are never written.  A normalize in between (not `independent`: one
        // multiplier per stream, :3439-3444) commutes with the mean up to rounding (mult * (l + r) / 2 for (mult l + mult r) / 2: the F32 tolerance
        // path, 1e-7 of full scale) and stays owed on the RESULT, whose row maxima receive the channels' (k_rs_onepole<..., 2>, flac_tail.hip)
        aukit_audio *o = *out;
        int rc = audio_prepare(ctx, &o, in->n, 1, in->rate, AUKIT_F32, in->len.data());
        if (rc) return rc;
        *out = o;
        int lrc = AUKIT_OK;
        if (lazy_onepole_try(ctx, const_cast<aukit_audio *>(in), in->lazy_fx_coef, in->lazy_fx == 2, &lrc, o)) {
            if (lrc) return lrc;
            o->pend_norm = false;
            if (in->pend_norm) { o->pend_norm = true; o->pend_peak = in->pend_peak; o->pend_independent = 0; o->pend_ctx = ctx; o->pend_ctx_id = ctx->id; }
            return AUKIT_OK;
        }
    }
    if (in->lazy_rs) { int lrc = lazy_resolve(ctx, const_cast<aukit_audio *>(in)); if (lrc) return lrc; }
    if (in->pend_norm && in->channels > AUKIT_MAX_CHANNELS) { int frc = audio_flush(ctx, in); if (frc) return frc; }   // (k_mono<NORM> keeps a multiplier per channel in registers: eight)
    aukit_audio *o = *out;
    int rc = audio_prepare(ctx, &o, in->n, 1, in->rate, in->dtype, in->len.data());
    if (rc) return rc;
    *out = o;
    if (in->n == 0) return AUKIT_OK;
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    dim3 grid(xblocks(in), in->n);
    if (in->pend_norm)  // the deferred effects.normalize of the input is applied as the rows are read (the input itself stays deferred)
        AUKIT_DISPATCH_T(in, hipLaunchKernelGGL((k_mono<T, true>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const T *>(in->dev), meta_of(in), reinterpret_cast<T *>(o->dev),
                                                meta_of(o), reinterpret_cast<const double *>(in->d_rowmax), in->pend_peak, in->pend_independent));
    else
        AUKIT_DISPATCH_T(in, hipLaunchKernelGGL((k_mono<T, false>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const T *>(in->dev), meta_of(in), reinterpret_cast<T *>(o->dev),
                                                meta_of(o), nullptr, 1.0, 0));
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, in->pend_norm ? "k_mono<normalize>" : "k_mono", audio_bytes(in) + audio_bytes(o));
}

int aukit_mix(aukit_ctx *ctx, const aukit_audio *const *audios, int count, double amplifier, aukit_audio **out) {
    if (!ctx || !audios || !out || count < 1 || !audios[0]) return fail(AUKIT_E_ARG, "null argument");
    const aukit_audio *self = audios[0];
    AUKIT_FLOAT_ONLY(self);
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    for (int a = 0; a < count; a++) AUKIT_FLUSH(ctx, audios[a]);
    int cn = self->channels;
    std::vector<uint64_t> lens(self->len);
    for (int a = 1; a < count; a++) {
        if (!audios[a] || audios[a]->n != self->n) return fail(AUKIT_E_ARG, "all audios of a mix must hold the same number of streams");
        if (audios[a]->dtype != self->dtype) return fail(AUKIT_E_ARG, "all audios of a mix must share a dtype");
        if (audios[a]->rate != self->rate) return fail(AUKIT_E_ARG, "mix: resample to a common sample rate first (Audio:mix does this with the default interpolation)");
        cn = std::max(cn, audios[a]->channels);
        for (uint32_t s = 0; s < self->n; s++) lens[s] = std::max(lens[s], audios[a]->len[s]);
    }
    for (int a = 0; a < count; a++)
        if (*out == audios[a]) return fail(AUKIT_E_ARG, "mix cannot run in place");
    aukit_audio *o = *out;
    int rc = audio_prepare(ctx, &o, self->n, cn, self->rate, self->dtype, lens.data());
    if (rc) return rc;
    *out = o;
    if (self->n == 0) return AUKIT_OK;
    uint64_t bytes = audio_bytes(o);
    dim3 grid(xblocks(o), o->n * o->channels);
    if (count > 8) {   // the sources' table in device memory (up to eight travel as kernel arguments)
        std::vector<MixEnt> ents((size_t)count);
        for (int a = 0; a < count; a++) { ents[a].data = audios[a]->dev; ents[a].m = meta_of(audios[a]); bytes += audio_bytes(audios[a]); }
        if ((rc = upload_table(ctx, ctx->misc_buf, ents.data(), ents.size() * sizeof(MixEnt)))) return rc;
        if ((rc = ctx_begin_kernel(ctx))) return rc;
        AUKIT_DISPATCH_T(o, hipLaunchKernelGGL((k_mix_many<T>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const MixEnt *>(ctx->misc_buf.p), count, reinterpret_cast<T *>(o->dev), meta_of(o), amplifier));
        AUKIT_HIP_CHECK(hipGetLastError());
        return ctx_end_kernel(ctx, "k_mix_many", bytes);
    }
    MixSrc S;
    S.count = count;
    for (int a = 0; a < count; a++) { S.data[a] = audios[a]->dev; S.m[a] = meta_of(audios[a]); bytes += audio_bytes(audios[a]); }
    if ((rc = ctx_begin_kernel(ctx))) return rc;
    AUKIT_DISPATCH_T(o, hipLaunchKernelGGL((k_mix<T>), grid, dim3(256), 0, ctx->stream, S, reinterpret_cast<T *>(o->dev), meta_of(o), amplifier));
    AUKIT_HIP_CHECK(hipGetLastError());
    return ctx_end_kernel(ctx, "k_mix", bytes);
}

int aukit_encode_pcm(aukit_ctx *ctx, const aukit_audio *in, int bit_depth, int data_type, int interleaved, aukit_audio **out) {
    if (!ctx || !in || !out) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_FLOAT_ONLY(in);
    AUKIT_FLUSH(ctx, in);
    if (bit_depth != 8 && bit_depth != 16 && bit_depth != 24 && bit_depth != 32) return fail(AUKIT_E_ARG, "bad argument #2 (invalid bit depth)");
    if (data_type < 0 || data_type > 2) return fail(AUKIT_E_ARG, "bad argument #3 (invalid data type)");
    if (data_type == AUKIT_FLOAT && bit_depth != 32) return fail(AUKIT_E_ARG, "bad argument #2 (float audio must have 32-bit depth)");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<uint64_t> lens(in->n);
    for (uint32_t s = 0; s < in->n; s++) lens[s] = in->len[s] * (uint64_t)in->channels;
    aukit_audio *o = *out;
    int rc = audio_prepare(ctx, &o, in->n, 1, in->rate, AUKIT_F64, lens.data());
    if (rc) return rc;
    *out = o;
    if (in->n == 0) return AUKIT_OK;
    const double maxValue = std::ldexp(1.0, bit_depth - 1);
    dim3 grid(xblocks(o), in->n);
    AUKIT_DISPATCH_T(in, hipLaunchKernelGGL((k_encode_pcm<T>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const T *>(in->dev), meta_of(in),
                                            reinterpret_cast<double *>(o->dev), meta_of(o), maxValue, data_type == AUKIT_UNSIGNED ? maxValue : 0.0,
                                            data_type == AUKIT_FLOAT ? 1 : 0, interleaved));
    AUKIT_HIP_CHECK(hipGetLastError());
    ctx->last_kernel = "k_encode_pcm";
    return AUKIT_OK;
}

int aukit_effect(aukit_ctx *ctx, aukit_audio *a, int id, const double *args, int nargs) {
    if (!ctx || !a) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_FLOAT_ONLY(a);
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    if (a->lazy_rs && !a->lazy_fx && !a->pend_norm && nargs >= 1 && args && (id == AUKIT_FX_HIGHPASS || id == AUKIT_FX_LOWPASS)) {   // resample + filter in one pass (flac_tail.hip)
        int lrc = AUKIT_OK;
        const double coef = id == AUKIT_FX_HIGHPASS ? 1 / (2 * M_PI * (args[0] / a->rate) + 1) : 1 - std::exp(-(args[0] / a->rate) * 2 * M_PI);   // :3607 / :3589
        if (a->channels == 2 && a->dtype == AUKIT_F32 && a->n && !getenv("AUKIT_NO_MONO_FUSION")) {
            // two channels: the filter is owed as well — Audio:mono may be the next reader (aukit_mono above); anyone else pays it first (lazy_resolve)
            a->lazy_fx = id == AUKIT_FX_HIGHPASS ? 2 : 1;
            a->lazy_fx_coef = coef;
            ctx->last_kernel = "(filter deferred)";
            return AUKIT_OK;
        }
        if (lazy_onepole_try(ctx, a, coef, id == AUKIT_FX_HIGHPASS, &lrc)) return lrc;
    }
    if (id == AUKIT_FX_NORMALIZE && a->lazy_rs && a->lazy_fx && !a->pend_norm) {
        // the peak search needs the FILTERED rows' maxima: the pass that pays the filter leaves them (k_rs_onepole), so the normalize is owed behind it
        int nrc = audio_rowmax_ensure(a);
        if (nrc) return nrc;
        auto argn = [&](int i, double def) { return (args && i < nargs) ? args[i] : def; };
        a->pend_norm = true;
        a->pend_peak = argn(0, 1.0);
        a->pend_independent = argn(1, 0.0) != 0;
        a->pend_ctx = ctx; a->pend_ctx_id = ctx->id;
        ctx->last_kernel = "(normalize deferred)";
        return AUKIT_OK;
    }
    AUKIT_FLUSH(ctx, a);  // every effect reads the samples
    const bool had_rowmax = a->rowmax_valid;
    a->rowmax_valid = false;  // ... and rewrites them (k_onepole leaves new maxima behind)
    auto arg = [&](int i, double def) { return (args && i < nargs) ? args[i] : def; };
    auto need = [&](int k) { return nargs >= k && args; };
    MapArgs A{};
    switch (id) {
    case AUKIT_FX_AMPLIFY:
        if (!need(1)) return fail(AUKIT_E_ARG, "bad argument #2 (expected number, got nil)");
        if (args[0] == 1) return AUKIT_OK;  // :3359
        A.a0 = args[0];
        return run_map<MAP_AMPLIFY>(ctx, a, A, "k_map<amplify>");
    case AUKIT_FX_INVERT:
        return run_map<MAP_INVERT>(ctx, a, A, "k_map<invert>");
    case AUKIT_FX_FADE: {
        if (!need(4)) return fail(AUKIT_E_ARG, "bad argument #5 (expected number, got nil)");
        const double startTime = args[0], startAmp = args[1], endTime = args[2], endAmp = args[3];
        if (startAmp == 1 && endAmp == 1) return AUKIT_OK;  // :3400
        const double start = startTime * a->rate, limit = endTime * a->rate;
        if (start <= limit) {  // the loop runs at least once: ch[i] must exist for every visited i (Q17)
            if (start != std::floor(start) || start < 1) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
            const double last = start + std::floor(limit - start);
            for (uint64_t l : a->len)
                if (last > (double)l) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        }
        A.a0 = start;
        A.a1 = limit;
        A.a2 = (endAmp - startAmp) / ((endTime - startTime) * a->rate);  // :3405
        A.a3 = startAmp;
        return run_map<MAP_FADE>(ctx, a, A, "k_map<fade>");
    }
    case AUKIT_FX_NORMALIZE:
        return fx_normalize(ctx, a, arg(0, 1.0), arg(1, 0.0) != 0, had_rowmax);
    case AUKIT_FX_CENTER: {
        if (a->rate != std::floor(a->rate) || a->rate < 1) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        if (a->n == 0) return AUKIT_OK;
        uint64_t mx = 1;
        for (uint64_t l : a->len) mx = std::max(mx, l);
        unsigned nw = (unsigned)std::min<uint64_t>((mx + (uint64_t)a->rate - 1) / (uint64_t)a->rate, 4096);
        dim3 grid(std::max(nw, 1u), a->n * a->channels);
        AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_center<T>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<T *>(a->dev), meta_of(a), (unsigned long long)a->rate));
        AUKIT_HIP_CHECK(hipGetLastError());
        ctx->last_kernel = "k_center";
        return AUKIT_OK;
    }
    case AUKIT_FX_TRIM:
        return fail(AUKIT_E_LUA, "bad argument #1 to 'sub' (string expected, got table)");  // :3495 (Q17)
    case AUKIT_FX_DELAY: {
        if (!need(1)) return fail(AUKIT_E_ARG, "bad argument #2 (expected number, got nil)");
        const double sd = std::floor(args[0] * a->rate);
        if (sd < 0) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        int rc = ctx->tmp_buf.ensure((size_t)a->total * dtype_size(a->dtype) + 64);
        if (rc) return rc;
        if (a->total) AUKIT_HIP_CHECK(hipMemcpyAsync(ctx->tmp_buf.p, a->dev, (size_t)a->total * dtype_size(a->dtype), hipMemcpyDeviceToDevice, ctx->stream));
        A.a0 = arg(1, 0.5);
        A.aux = ctx->tmp_buf.p;
        A.lag = sd > 1.8e19 ? ~0ull : (unsigned long long)sd;
        return run_map<MAP_DELAY>(ctx, a, A, "k_map<delay>");
    }
    case AUKIT_FX_ECHO: {
        const double sd = std::floor(arg(0, 1.0) * a->rate);
        if (sd < 0) return fail(AUKIT_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        if (a->n == 0) return AUKIT_OK;
        if (sd == 0) return fail(AUKIT_E_UNSUPPORTED, "echo with a zero-sample delay");
        const unsigned long long L = sd > 1.8e19 ? ~0ull : (unsigned long long)sd;
        dim3 grid((unsigned)std::min<unsigned long long>((L + 255) / 256, 1024), a->n * a->channels);
        AUKIT_DISPATCH_T(a, hipLaunchKernelGGL((k_echo<T>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<T *>(a->dev), meta_of(a), L, arg(1, 0.5)));
        AUKIT_HIP_CHECK(hipGetLastError());
        ctx->last_kernel = "k_echo";
        return AUKIT_OK;
    }
    case AUKIT_FX_REVERB:
        return fx_reverb(ctx, a, arg(0, 100.0), arg(1, 0.3), arg(2, 1.0), arg(3, 0.0));
    case AUKIT_FX_LOWPASS:
        if (!need(1)) return fail(AUKIT_E_ARG, "bad argument #2 (expected number, got nil)");
        return fx_onepole(ctx, a, 1 - std::exp(-(args[0] / a->rate) * 2 * M_PI), false);  // :3589
    case AUKIT_FX_HIGHPASS:
        if (!need(1)) return fail(AUKIT_E_ARG, "bad argument #2 (expected number, got nil)");
        return fx_onepole(ctx, a, 1 / (2 * M_PI * (args[0] / a->rate) + 1), true);          // :3607
    case AUKIT_FX_SPEED: {
        if (!need(1)) return fail(AUKIT_E_ARG, "bad argument #2 (expected number, got nil)");
        if (args[0] == 1) return AUKIT_OK;  // :3379
        const double rate = a->rate;
        aukit_audio *tmp = nullptr;
        a->rate = rate * args[0];  // :3381
        int rc = aukit_resample(ctx, a, rate, (int)arg(1, AUKIT_INTERP_LINEAR), &tmp);
        a->rate = rate;
        if (rc) { if (tmp) aukit_audio_free(tmp); return rc; }
        // audio.data = new.data: adopt the resampled buffers
        std::swap(a->dev, tmp->dev); std::swap(a->cap_bytes, tmp->cap_bytes); std::swap(a->d_meta, tmp->d_meta); std::swap(a->meta_cap, tmp->meta_cap);
        a->len = tmp->len; a->row_off = tmp->row_off; a->row_stride = tmp->row_stride; a->total = tmp->total;
        a->version++;
        AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        aukit_audio_free(tmp);
        return AUKIT_OK;
    }
    }
    return fail(AUKIT_E_ARG, "unknown effect id %d", id);
}

}  // extern "C"
