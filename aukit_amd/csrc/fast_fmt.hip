// fast_fmt.hip — the wave-private f32 resampler (fast2.hip) for every sample format the specialised kernels leave over: interleaved PCM of
// 1-4 bytes per sample, either byte order, signed / 8-bit unsigned / float, one or two channels, and two-channel G.711 —
//   aukit.pcm(data, bits, type, channels, rate, bigEndian):resample(new_rate)  (aukit.lua:1049-1171 + :653-675)  and
//   aukit.g711(data, ulaw, channels, rate):resample(new_rate)                   (:1361-1384)
// with AUKIT_F32 storage, decode and resample in one launch.  (Round 2 unpacked these formats to f32 rows and ran the row kernel on them —
// 258 G out-samples/s on 24-bit stereo — or left them to the reference-order kernel: float strings 203 G, G.711 stereo 372 G.)
//
// A source element is a FRAME of C samples of B bytes; frames of 3 or 6 bytes do not divide a 16-byte vector, so the window's raw vectors
// are parked in LDS as they arrive and every lane picks its samples out of them (two aligned dwords + v_alignbyte: any byte offset, any
// B <= 4), converts, and writes one f32 table per channel.  From there on it is k_fast_wave_s16x2: one position per lane and row, each
// channel evaluated with it.  Tiles are 512 outputs per channel: raw window + two tables stay below 8 KiB per wave for 24-bit stereo.
//
// Samples beyond [-1, 1] (float strings can hold anything): where the reference's ROUNDED position x misses an integer the exact rational
// position hits (or the other way round) the reference clamps what this kernel copies — invisible inside [-1, 1], arbitrarily large outside.
// The kernel therefore raises a flag when it stages such a sample, and the reference-order kernel, queued right behind with the flag as
// its launch condition (ResampleParams::only_if), redoes the call; without the flag it returns at once.
#include <algorithm>
#include "fast_stream_body.h"   // interp_qr_raw, prev_lane / last_lane: the stream.pcm epilogue

namespace aukit {

constexpr int WTF = 512;  // outputs per wave tile and channel = 8 rows of 64

enum { FMT_SIGNED = 0, FMT_UNSIGNED8 = 1, FMT_FLOAT = 2, FMT_ULAW = 3, FMT_ALAW = 4, FMT_UNSIGNED = 5 /* 16 / 24 / 32-bit unsigned: (s - 128) / (s < 128 and 2^(b-1) or 2^(b-1) - 1), Q4 — the stream path only */ };
struct FmtParams {
    int kind, big_endian;
    float scale_pos, scale_neg;   // signed: 1 / (2^(bits-1) - 1), 1 / 2^(bits-1)   (:1133)
    float g711_scale;
    int *flag;                    // FMT_FLOAT: set when a staged sample is not inside [-1, 1]
    // fp64 arithmetic (AUKIT_OPT_EXACT_MATH = 1): s / (2^(bits-1) - 1) as a correctly rounded double quotient (div_rcp), s * 2^-(bits-1) exact
    double dpos, drcp, dneg, dg711, dinv_b;
};

AUKIT_DEV int g711_int(unsigned byte, int ulaw) {   // aukit.lua:1371-1379 up to the division by 0x2000
    unsigned b = byte ^ (ulaw ? 0xFFu : 0x55u);
    int m = b & 15, e = (b >> 4) & 7;
    if (!ulaw && e == 0) m = m * 4 + 2;
    else m = (m * 2 + 33) << e;
    if (ulaw) m -= 33;
    const bool neg = ((b & 0x80) != 0) == (ulaw != 0);
    return neg ? -m : m;
}

template <int B, typename AT>
AUKIT_DEV AT fmt_sample(unsigned raw, const FmtParams &M, bool &wild) {
    // `raw`: the sample's B bytes in memory order, lowest address in the low byte
    constexpr bool F64 = sizeof(AT) == 8;
    int v;
    if constexpr (B == 1) {
        if (M.kind == FMT_ULAW || M.kind == FMT_ALAW) {
            if constexpr (F64) return (AT)((double)g711_int(raw & 0xFF, M.kind == FMT_ULAW) * M.dg711);
            else return (AT)g711_f32b(raw & 0xFF, M.kind == FMT_ULAW, M.g711_scale);
        }
        v = M.kind == FMT_UNSIGNED8 ? (int)(raw & 0xFF) - 128 : (int)(signed char)(raw & 0xFF);   // unsigned: s - 128, with s < 128 deciding the divisor — the same test as v < 0 (Q4)
    } else {
        unsigned u = raw;
        if (M.big_endian) {
            if constexpr (B == 2) u = ((raw & 0xFF) << 8) | ((raw >> 8) & 0xFF);
            else if constexpr (B == 3) u = ((raw & 0xFF) << 16) | (raw & 0xFF00) | ((raw >> 16) & 0xFF);
            else u = __builtin_bswap32(raw);
        }
        if constexpr (B == 4) {
            if (M.kind == FMT_FLOAT) {
                const float f = __uint_as_float(u);
                wild = wild || !(fabsf(f) <= 1.0f);
                return (AT)f;
            }
        }
        if (M.kind == FMT_UNSIGNED) {   // :1152 (Q4: 128 is subtracted whatever the depth)
            const float uf = (float)u;
            return (AT)((uf - 128.0f) * (u < 128u ? M.scale_neg : M.scale_pos));
        }
        v = (int)(u << (32 - 8 * B)) >> (32 - 8 * B);
    }
    if constexpr (F64) return (AT)(v < 0 ? (double)v * M.dneg : div_rcp((double)v, M.dpos, M.drcp));   // :1133
    else return (AT)((float)v * (v < 0 ? M.scale_neg : M.scale_pos));
}

// one output in fp64: the Catmull-Rom polynomial (:261-266) as a Horner form on fx = rem * RN(1/b), every tap, product and sum a double
// (the arithmetic of wave_f64.hip's table-less variant); rounded to f32, then :667-668
template <int INTERP>
AUKIT_DEV float interp_qr_f64(const double *tab, unsigned q, unsigned rem, double inv_b) {
    const double fx = (double)rem * inv_b;
    const double p1 = tab[q], p2 = tab[q + 1];
    double v;
    if constexpr (INTERP == AUKIT_INTERP_LINEAR) v = __builtin_fma(p2 - p1, fx, p1);
    else {
        const double p0 = tab[(int)q - 1], p3 = tab[q + 2];
        const double c3 = __builtin_fma(1.5, p1 - p2, 0.5 * (p3 - p0));
        const double c2 = __builtin_fma(-0.5, p3, __builtin_fma(2.0, p2, __builtin_fma(-2.5, p1, p0)));
        const double c1 = 0.5 * (p2 - p0);
        v = __builtin_fma(__builtin_fma(__builtin_fma(c3, fx, c2), fx, c1), fx, p1);
    }
    const float c = __builtin_amdgcn_fmed3f((float)v, -1.0f, 1.0f);
    return rem == 0 ? (float)p1 : c;
}
template <int INTERP, typename AT>
AUKIT_DEV float interp_any(const FastParams &F, const FmtParams &M, const AT *tab, unsigned q, unsigned rem) {
    if constexpr (sizeof(AT) == 8) return interp_qr_f64<INTERP>(tab, q, rem, M.dinv_b);
    else return interp_qr<SRC_AUDIO_F32, INTERP>(F, tab, q, rem);
}

template <int FB, int HL, int HR>
AUKIT_DEV WaveTile describe_fmt(const ResampleParams &P, const FastParams &F, unsigned t) {
    unsigned sidx, tin;
    if (P.tiles_per_seg) { sidx = t / P.tiles_per_seg; tin = t - sidx * P.tiles_per_seg; }
    else { sidx = as_const(P.tile_seg)[t]; tin = t - as_const(P.seg_tile0)[sidx]; }
    const Seg sg = load_seg(P.segs, sidx);
    WaveTile w;
    const unsigned o0 = tin * (unsigned)WTF;
    w.cnt = o0 < sg.n_out ? min((unsigned)WTF, sg.n_out - o0) : 0u;
    const unsigned td = tin * F.wd;                    // (o0 * a) = (tin * wc + td / b) * b + td % b
    const unsigned tq = td / F.b;
    const unsigned kb = tin * F.wc + tq;
    w.r0 = td - tq * F.b;
    const unsigned klast = w.cnt ? (w.r0 + (w.cnt - 1) * F.a) / F.b : 0u;
    w.k_lo = 1 + (int)kb - HL;
    w.n_stage = (int)klast + 1 + HL + HR;
    w.w_lo = sg.w_lo;
    w.w_hi = sg.w_hi;
    w.base = P.src + (size_t)as_const(P.src_off)[sg.stream] + (long long)FB * sg.src_base;
    const unsigned char *a0 = w.base + (long long)FB * w.k_lo;
    w.al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
    w.head = (int)(a0 - w.al);                         // BYTES into the first vector (frames need not divide it)
    w.nvec = (w.head + w.n_stage * FB + 15) / 16;
    w.orow = reinterpret_cast<float *>(P.out) + sg.out_off + o0;
    return w;
}

// EPI 0: Audio:resample (:666-668), one row per channel.  EPI 1: aukit.stream.pcm's chunk samples (:2395-2403: the interpolated sample is not
// clamped, 2-tap low-pass on the raw sample before it, x 127 | 128, clamp) per channel; EPI 2: the same on the channels' mean (`mono`, :2368).
template <int B, int C, int INTERP, int NV, typename AT, int EPI = 0>
__global__ __launch_bounds__(256) void k_fast_wave_fmt(const ResampleParams P, const FastParams F, const FmtParams M) {
    extern __shared__ float smf[];
    constexpr int FB = B * C;
    constexpr int HL = (INTERP == AUKIT_INTERP_CUBIC ? 1 : 0) + (EPI ? 1 : 0), HR = INTERP == AUKIT_INTERP_CUBIC ? 2 : 1;   // the stream epilogue reaches one tap further left
    constexpr int TC = EPI == 2 ? 1 : C;                  // tables (= output rows)
    constexpr int RAWW = NV * 64 * 4 + 4;               // dwords of raw window per wave (+ one vector: the second dword of the last sample)
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr unsigned AW = sizeof(AT) / 4;                 // dwords per table entry
    unsigned *const raw = reinterpret_cast<unsigned *>(smf) + wave * (unsigned)(RAWW + TC * F.cap * AW);
    AT *const tab0 = reinterpret_cast<AT *>(raw + RAWW);
    const unsigned nwaves = gridDim.x * 4u;
    const unsigned lane_a = (unsigned)lane * F.a;
    bool wild = false;

    unsigned t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave);
    if (t >= P.n_tiles) return;
    uint4 pre[NV];
    WaveTile cur = describe_fmt<FB, HL, HR>(P, F, t);
    issue_loads<NV>(P, cur, lane, pre);
    auto global_sample = [&](const unsigned char *q) {  // a sample straight from memory (edges, vectors at the rim of the allocation)
        unsigned u = 0;
#pragma unroll
        for (int b = 0; b < B; b++) u |= (unsigned)q[b] << (8 * b);
        return fmt_sample<B, AT>(u, M, wild);
    };
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        // ---- the raw vectors, as they are
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = lane + 64 * i;
            if (v < cur.nvec) *reinterpret_cast<uint4 *>(raw + 4 * v) = pre[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- samples → one f32 table per channel
        const int ne = cur.n_stage * C;
        auto lds_sample = [&](unsigned off) {
            const unsigned w0 = raw[off >> 2], w1 = raw[(off >> 2) + 1];
            const unsigned u = B == 4 ? __builtin_amdgcn_alignbyte(w1, w0, off & 3) : (__builtin_amdgcn_alignbyte(w1, w0, off & 3) & ((1u << (8 * (B & 3))) - 1u));
            return fmt_sample<B, AT>(u, M, wild);
        };
        if constexpr (EPI == 2) {   // self[i] = ((0 + read()) + read()) / channels  (:2368), in the table's arithmetic
#pragma unroll 4
            for (int f = lane; f < cur.n_stage; f += 64) {
                const unsigned off = (unsigned)cur.head + (unsigned)f * FB;
                tab0[f] = (lds_sample(off) + lds_sample(off + B)) * (AT)0.5;
            }
        } else {
#pragma unroll 4
            for (int e = lane; e < ne; e += 64) {
                const int f = C == 1 ? e : e >> 1, c = C == 1 ? 0 : e & 1;
                tab0[c * F.cap + f] = lds_sample((unsigned)cur.head + (unsigned)e * B);
            }
        }
        {
            const unsigned char *lo = cur.al, *hi = cur.al + 16 * (size_t)cur.nvec;
            if (lo < P.safe_lo || hi > P.safe_hi) {  // wave-uniform, rare: vectors that straddle the allocation were zero-filled
                for (int e = lane; e < ne; e += 64) {
                    const unsigned char *q = cur.al + cur.head + (size_t)e * B;
                    const unsigned char *v0 = (const unsigned char *)((uintptr_t)q & ~(uintptr_t)15), *v1 = (const unsigned char *)((uintptr_t)(q + B - 1) & ~(uintptr_t)15);
                    const bool ok0 = v0 >= P.safe_lo && v0 + 16 <= P.safe_hi, ok1 = v1 >= P.safe_lo && v1 + 16 <= P.safe_hi;
                    if (!(ok0 && ok1)) {
                        const int f = C == 1 ? e : e >> 1, c = C == 1 ? 0 : e & 1;
                        const AT v = (q >= P.safe_lo && q + B <= P.safe_hi) ? global_sample(q) : (AT)0;
                        if constexpr (EPI == 2) {   // (rare: redo the frame's mean from memory)
                            if (c == 0) {
                                const unsigned char *q1 = q + B;
                                const AT v1 = (q1 >= P.safe_lo && q1 + B <= P.safe_hi) ? global_sample(q1) : (AT)0;
                                tab0[f] = (v + v1) * (AT)0.5;
                            }
                        } else tab0[c * F.cap + f] = v;
                    }
                }
            }
            // nil fall-backs of interpolate.{linear,cubic} (aukit.lua:259, :264) = replicated edge samples
            const int k_hi = cur.k_lo + cur.n_stage - 1;
            auto edge = [&](const unsigned char *q, int c) -> AT {
                if constexpr (EPI == 2) return (global_sample(q) + global_sample(q + B)) * (AT)0.5;
                else return global_sample(q + c * B);
            };
            if (cur.k_lo < cur.w_lo) {
                const unsigned char *q = cur.base + (long long)FB * cur.w_lo;
#pragma unroll
                for (int c = 0; c < TC; c++) {
                    const AT ev = edge(q, c);
                    for (int idx = lane; idx < cur.w_lo - cur.k_lo; idx += 64) tab0[c * F.cap + idx] = ev;
                }
            }
            if (k_hi > cur.w_hi) {
                const unsigned char *q = cur.base + (long long)FB * cur.w_hi;
                const int first = cur.w_hi + 1 - cur.k_lo;
#pragma unroll
                for (int c = 0; c < TC; c++) {
                    const AT ev = edge(q, c);
                    for (int idx = lane; idx < k_hi - cur.w_hi; idx += 64) tab0[c * F.cap + first + idx] = ev;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the tile's segment: channel c's row is c * out_stride elements after the first one
        unsigned sidx;
        if (P.tiles_per_seg) sidx = t / P.tiles_per_seg; else sidx = as_const(P.tile_seg)[t];
        const unsigned ostride = ((const AUKIT_CONST_AS Seg *)(P.segs + sidx))->out_stride;
        const unsigned tn = t + nwaves;
        const bool more = tn < P.n_tiles;
        WaveTile nxt = cur;
        if (more) {  // wave-uniform
            nxt = describe_fmt<FB, HL, HR>(P, F, tn);
            issue_loads<NV>(P, nxt, lane, pre);  // in flight while this tile is interpolated
        }
        const AT *tabL = tab0 + HL, *tabR = tab0 + F.cap + HL;  // tab[q] = d[1 + kb + q]
        float *orowL = cur.orow, *orowR = cur.orow + ostride;
        if constexpr (EPI != 0) {
            // aukit.stream.pcm (the body of k_fast_wave_stream, fast_stream_body.h, per table): the raw sample before the tile's first output is
            // re-evaluated from the window (position n = r0 - a, one table step back when that is negative); the first output of an iterator
            // call has ls = 0 behind it (:2397)
            const bool first = (P.tiles_per_seg ? t % P.tiles_per_seg : t - as_const(P.seg_tile0)[as_const(P.tile_seg)[t]]) == 0;
            const float alpha = F.alpha;
            float carry[TC];
#pragma unroll
            for (int c = 0; c < TC; c++) {
                const float *tb = reinterpret_cast<const float *>(c ? tabR : tabL);
                carry[c] = first ? 0.f : (cur.r0 >= F.a ? interp_row<INTERP, false>(F, tb, cur.r0 - F.a) : interp_row<INTERP, false>(F, tb - 1, cur.r0 + F.b - F.a));
            }
            auto finish = [&](float sv, float &cy) {
                const float prev = prev_lane(sv, cy);
                cy = last_lane(sv);
                const float ns = fmaf(alpha, sv - prev, prev);                                                 // :2401
                return __builtin_amdgcn_fmed3f(ns * (ns < 0.f ? 128.f : 127.f), -128.f, 127.f);              // :2402
            };
            if ((cur.cnt & 63u) == 0) {   // whole rows: (q, rem) advanced by additions
                const unsigned n0 = cur.r0 + lane_a;
                unsigned q = __umulhi(n0, F.magic);
                unsigned rem = n0 - q * F.b;
                const int rows = (int)(cur.cnt >> 6);
                for (int r = 0; r < WTF / 64; r++) {
                    if (r >= rows) break;
                    orowL[r * 64 + lane] = finish(interp_qr_raw<INTERP>(F, reinterpret_cast<const float *>(tabL), q, rem), carry[0]);
                    if constexpr (TC == 2) orowR[r * 64 + lane] = finish(interp_qr_raw<INTERP>(F, reinterpret_cast<const float *>(tabR), q, rem), carry[TC - 1]);
                    rem += F.dr64;
                    q += F.dq64;
                    const bool wrap = rem >= F.b;
                    rem -= wrap ? F.b : 0u;
                    q += wrap ? 1u : 0u;
                }
            } else {
                for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
                    const unsigned j = rb + lane;
                    const unsigned n = cur.r0 + (j < cur.cnt ? j : cur.cnt - 1) * F.a;
                    const float vl = finish(interp_row<INTERP, false>(F, reinterpret_cast<const float *>(tabL), n), carry[0]);
                    if (j < cur.cnt) orowL[j] = vl;
                    if constexpr (TC == 2) {
                        const float vr = finish(interp_row<INTERP, false>(F, reinterpret_cast<const float *>(tabR), n), carry[TC - 1]);
                        if (j < cur.cnt) orowR[j] = vr;
                    }
                }
            }
        } else
        if (cur.cnt == (unsigned)WTF) {
            const unsigned n0 = cur.r0 + lane_a;
            unsigned q = __umulhi(n0, F.magic);
            unsigned rem = n0 - q * F.b;
#pragma unroll
            for (int r = 0; r < WTF / 64; r++) {
                orowL[r * 64 + lane] = interp_any<INTERP, AT>(F, M, tabL, q, rem);
                if constexpr (C == 2) orowR[r * 64 + lane] = interp_any<INTERP, AT>(F, M, tabR, q, rem);
                rem += F.dr64;
                q += F.dq64;
                const bool wrap = rem >= F.b;
                rem -= wrap ? F.b : 0u;
                q += wrap ? 1u : 0u;
            }
        } else {
            for (unsigned rb = 0; rb < cur.cnt; rb += 64) {
                const unsigned j = rb + lane;
                const unsigned n = cur.r0 + (j < cur.cnt ? j : cur.cnt - 1) * F.a;
                const unsigned q = __umulhi(n, F.magic);
                const unsigned rem = n - q * F.b;
                const float vl = interp_any<INTERP, AT>(F, M, tabL, q, rem);
                if (j < cur.cnt) orowL[j] = vl;
                if constexpr (C == 2) {
                    const float vr = interp_any<INTERP, AT>(F, M, tabR, q, rem);
                    if (j < cur.cnt) orowR[j] = vr;
                }
            }
        }
        if (!more) break;
        cur = nxt;
        t = tn;
    }
    if constexpr (B == 4) {
        if (M.flag && __builtin_amdgcn_ballot_w64(wild) != 0 && lane == 0) atomicOr(M.flag, 1);
    }
}

template <int B, int C, int INTERP, typename AT, int EPI>
static void launch_fmt_nv(aukit_ctx *ctx, int nv, const ResampleParams &P, const FastParams &F, const FmtParams &M, size_t lds, unsigned grid) {
    if (nv == 1) hipLaunchKernelGGL((k_fast_wave_fmt<B, C, INTERP, 1, AT, EPI>), dim3(grid), dim3(256), lds, ctx->stream, P, F, M);
    else if (nv == 8) hipLaunchKernelGGL((k_fast_wave_fmt<B, C, INTERP, 8, AT, EPI>), dim3(grid), dim3(256), lds, ctx->stream, P, F, M);
    else hipLaunchKernelGGL((k_fast_wave_fmt<B, C, INTERP, 4, AT, EPI>), dim3(grid), dim3(256), lds, ctx->stream, P, F, M);
}
template <int B, int C>
static void launch_fmt_i(aukit_ctx *ctx, int interp, bool f64, int epi, int nv, const ResampleParams &P, const FastParams &F, const FmtParams &M, size_t lds, unsigned grid) {
    const bool lin = interp == AUKIT_INTERP_LINEAR;
    if (epi == 1) {   // stream.pcm, f32 arithmetic only
        if (lin) launch_fmt_nv<B, C, AUKIT_INTERP_LINEAR, float, 1>(ctx, nv, P, F, M, lds, grid);
        else launch_fmt_nv<B, C, AUKIT_INTERP_CUBIC, float, 1>(ctx, nv, P, F, M, lds, grid);
    } else if (epi == 2) {
        if constexpr (C == 2) {
            if (lin) launch_fmt_nv<B, C, AUKIT_INTERP_LINEAR, float, 2>(ctx, nv, P, F, M, lds, grid);
            else launch_fmt_nv<B, C, AUKIT_INTERP_CUBIC, float, 2>(ctx, nv, P, F, M, lds, grid);
        }
    } else if (f64) {
        if (lin) launch_fmt_nv<B, C, AUKIT_INTERP_LINEAR, double, 0>(ctx, nv, P, F, M, lds, grid);
        else launch_fmt_nv<B, C, AUKIT_INTERP_CUBIC, double, 0>(ctx, nv, P, F, M, lds, grid);
    } else {
        if (lin) launch_fmt_nv<B, C, AUKIT_INTERP_LINEAR, float, 0>(ctx, nv, P, F, M, lds, grid);
        else launch_fmt_nv<B, C, AUKIT_INTERP_CUBIC, float, 0>(ctx, nv, P, F, M, lds, grid);
    }
}

bool fast_eligible(int src_kind, int interp, double old_rate, double new_rate, FastParams &F);

// The fused decode + resample of an interleaved PCM / G.711 string of one or two channels on the f32 tolerance path.  Returns false
// (nothing launched) when the shape is not served.  *needs_exact: a float string — the caller queues the reference-order kernel behind
// this one with P.only_if = the flag this launch may raise.
bool fast_fmt_try(aukit_ctx *ctx, const aukit_codec_desc *d, int interp, double new_rate, const std::vector<Seg> &segs, ResampleParams &P,
                  uint64_t algorithmic_bytes, int *rc, const int **only_if, int epi, double alpha) {
    *only_if = nullptr;
    if (ctx->exact_math > 1 || getenv("AUKIT_NO_FAST_FMT")) return false;
    const bool f64 = ctx->exact_math == 1;   // fp64 arithmetic, f32 store
    if (epi && f64) return false;            // the stream epilogue exists in f32 arithmetic only
    if (epi == 2 && d->channels != 2) return false;
    const int C = d->channels;
    if (C != 1 && C != 2) return false;
    FmtParams M;
    memset(&M, 0, sizeof M);
    int B;
    if (d->codec == AUKIT_CODEC_G711) {
        B = 1;
        M.kind = P.ulaw ? FMT_ULAW : FMT_ALAW;
        M.g711_scale = (float)P.g711_scale;
        M.dg711 = P.g711_scale;
    } else if (d->codec == AUKIT_CODEC_PCM) {
        B = d->bit_depth / 8;
        if (B < 1 || B > 4 || P.planar) return false;
        if (d->data_type == AUKIT_FLOAT) { if (B != 4) return false; M.kind = FMT_FLOAT; }
        else if (d->data_type == AUKIT_SIGNED) M.kind = FMT_SIGNED;
        else if (d->data_type == AUKIT_UNSIGNED && B == 1) M.kind = FMT_UNSIGNED8;
        else if (d->data_type == AUKIT_UNSIGNED && epi) M.kind = FMT_UNSIGNED;   // stream.pcm does not clamp its interpolated sample (Q2): samples beyond ±1 are no obstacle there
        else return false;  // unsigned 16 / 24 / 32-bit: Q4 maps them to [-1, 3): the reference-order kernel keeps them for Audio:resample
        M.big_endian = d->big_endian ? 1 : 0;
        const double full = (double)(1ull << (8 * B - 1));
        M.scale_pos = (float)(1.0 / (full - 1));
        M.scale_neg = (float)(1.0 / full);
        M.dpos = full - 1; M.drcp = 1.0 / (full - 1); M.dneg = 1.0 / full;
    } else return false;
    FastParams F;
    if (!fast_eligible(SRC_AUDIO_F32, interp, d->sample_rate, new_rate, F)) return false;
    for (const Seg &g : segs)
        if (g.w_hi < g.w_lo && g.n_out) return false;
    const int FB = B * C;
    F.epi = epi;
    F.alpha = (float)alpha;
    const int hl = (interp == AUKIT_INTERP_CUBIC ? 1 : 0) + (epi ? 1 : 0), hr = interp == AUKIT_INTERP_CUBIC ? 2 : 1;
    const int win = (int)(((unsigned long long)(WTF - 1) * F.a) / F.b) + 2 + hl + hr;  // staged frames per wave tile (upper bound)
    const int nvec = (win * FB + 15 + 15) / 16;
    const int nv = nvec <= 64 ? 1 : (nvec <= 256 ? 4 : (nvec <= 512 ? 8 : 0));
    if (!nv) return false;  // strong down-sampling
    uint64_t max_tiles = 0;
    for (const Seg &g : segs) max_tiles = std::max<uint64_t>(max_tiles, (g.n_out + WTF - 1) / WTF);
    F.wc = (unsigned)(((unsigned long long)WTF * F.a) / F.b);
    F.wd = (unsigned)(((unsigned long long)WTF * F.a) % F.b);
    if ((double)max_tiles * (double)F.wd >= 4294967296.0 || ((double)max_tiles + 1) * (double)F.wc >= 2147483648.0) return false;
    if (((double)F.b + (double)WTF * (double)F.a) * (double)F.b >= 4294967296.0) return false;
    F.cap = (win + 8 + 15) & ~15;
    F.dq64 = (unsigned)((64ull * F.a) / F.b);
    F.dr64 = (unsigned)((64ull * F.a) % F.b);
    M.dinv_b = 1.0 / (double)F.b;
    const size_t lds = 4 * ((size_t)(nv * 64 * 4 + 4) + (size_t)(epi == 2 ? 1 : C) * F.cap * (f64 ? 2 : 1)) * 4;
    if (lds > 64 * 1024) return false;
    if (M.kind == FMT_FLOAT && !epi) {
        if ((*rc = ctx->fmt_flag.ensure(64))) return true;
        M.flag = reinterpret_cast<int *>(ctx->fmt_flag.p);
    }
    if ((*rc = plan_tiles_sized(ctx, segs, WTF, P))) return true;
    *rc = AUKIT_OK;
    if (P.n_tiles == 0) return true;
    unsigned per_cu = 64;   // measured 4 / 8 / 16 / 32 / 64 on 1024 ten-second streams: float mono 497 / 549 / 555 / 569 / 600 G out-samples/s, 24-bit stereo 629 / 651 / 660 / 649 / 667
    if (const char *e = getenv("AUKIT_FMT_PER_CU")) { const int v = atoi(e); if (v >= 1) per_cu = (unsigned)v; }   // tuning knob
    const unsigned grid = std::min<unsigned>((P.n_tiles + 3) / 4, (unsigned)ctx->num_cus * per_cu);
    if ((*rc = ctx_begin_kernel(ctx))) return true;
    if (M.flag && hipMemsetAsync(M.flag, 0, 4, ctx->stream) != hipSuccess) { *rc = fail(AUKIT_E_HIP, "hipMemsetAsync failed"); return true; }
#define AUKIT_FMT(BB, CC) launch_fmt_i<BB, CC>(ctx, interp, f64, epi, nv, P, F, M, lds, grid)
    if (C == 1) { if (B == 1) AUKIT_FMT(1, 1); else if (B == 2) AUKIT_FMT(2, 1); else if (B == 3) AUKIT_FMT(3, 1); else AUKIT_FMT(4, 1); }
    else { if (B == 1) AUKIT_FMT(1, 2); else if (B == 2) AUKIT_FMT(2, 2); else if (B == 3) AUKIT_FMT(3, 2); else AUKIT_FMT(4, 2); }
#undef AUKIT_FMT
    if (hipGetLastError() != hipSuccess) { *rc = fail(AUKIT_E_HIP, "k_fast_wave_fmt launch failed"); return true; }
    static thread_local char nm[96];
    static const char *kn[] = {"signed", "unsigned", "float", "ulaw", "alaw", "unsigned"};
    snprintf(nm, sizeof nm, "k_fast_wave_fmt<%s%d%s,%dch,%s,nv%d%s%s>", kn[M.kind], 8 * B, M.big_endian ? "be" : "", C, interp == AUKIT_INTERP_LINEAR ? "linear" : "cubic", nv, f64 ? ",f64" : "",
             epi == 1 ? ",stream_pcm" : (epi == 2 ? ",stream_pcm_mono" : ""));
    *rc = ctx_end_kernel(ctx, nm, algorithmic_bytes);
    if (M.flag) *only_if = M.flag;
    return true;
}

}  // namespace aukit
