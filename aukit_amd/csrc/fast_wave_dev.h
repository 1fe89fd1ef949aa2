// fast_wave_dev.h — device helpers of the wave-private f32 resampler, shared by its translation units: fast2.hip (Audio:resample
// epilogue — the headline kernel) and fast_stream.hip (stream.pcm epilogue).  The epilogues live in separate translation units on
// purpose: adding the second epilogue as a template parameter of the one kernel changed the register allocation of the headline
// instantiation and cost it 18 % (A/B on one box, tools/ab_lib.sh) although its source path was untouched.
#pragma once
#include "resample.h"

namespace aukit {

#ifndef AUKIT_WT
#define AUKIT_WT 1024
#endif
constexpr int WT = AUKIT_WT;  // outputs per wave tile = 16 rows of 64

template <int SRC> struct SrcTraits;
template <> struct SrcTraits<SRC_PCM_S16LE_MONO> { static constexpr int BYTES = 2, SPV = 8; };
template <> struct SrcTraits<SRC_G711_MONO> { static constexpr int BYTES = 1, SPV = 16; };
template <> struct SrcTraits<SRC_PCM8_MONO> { static constexpr int BYTES = 1, SPV = 16; };  // 8-bit PCM: s / (s < 0 and 128 or 127), or (s - 128) / (s < 128 and 128 or 127) for unsigned (Q4)
template <> struct SrcTraits<SRC_I8> { static constexpr int BYTES = 1, SPV = 16; };         // raw int8 rows (the DFPWM decoder's samples, stream.dfpwm): the value itself
template <> struct SrcTraits<SRC_AUDIO_F32> { static constexpr int BYTES = 4, SPV = 4; };
template <> struct SrcTraits<SRC_I32> { static constexpr int BYTES = 4, SPV = 4; };  // integer rows (FLAC): v * 2^-depth, exact in f32 for |v| < 2^24

// 8-bit PCM sample → f32 (tolerance path: the quotient by 127 as a multiplication, one ulp of f32 from the reference's double division)
AUKIT_DEV float pcm8_f32(unsigned byte, int is_unsigned) {
    const int v = is_unsigned ? (int)byte - 128 : (int)(signed char)byte;  // unsigned: s - 128, with s < 128 deciding the divisor — the same test as v < 0
    return (float)v * (v < 0 ? 1.0f / 128.0f : 1.0f / 127.0f);
}
AUKIT_DEV float g711_f32b(unsigned byte, int ulaw, float scale) {
    unsigned b = byte ^ (ulaw ? 0xFFu : 0x55u);
    int m = b & 15, e = (b >> 4) & 7;
    if (!ulaw && e == 0) m = m * 4 + 2;
    else m = (m * 2 + 33) << e;
    if (ulaw) m -= 33;
    bool neg = ((b & 0x80) != 0) == (ulaw != 0);
    return (float)(neg ? -m : m) * scale;
}

// Read-only descriptor tables (segments, tile maps, stream offsets) are read through the CONSTANT address space: with a
// wave-uniform index that is an s_load (SMEM, lgkmcnt) wherever it stands.  As plain global loads hipcc must assume that the
// kernel's own output stores may alias them and emits VMEM loads once a store precedes them — and on gfx9 waiting for a VMEM
// load (vmcnt) also waits for every store issued before it: the per-tile descriptor fetch drained the wave's whole store queue.
#define AUKIT_CONST_AS __attribute__((address_space(4)))
template <typename T> AUKIT_DEV const AUKIT_CONST_AS T *as_const(const T *p) { return (const AUKIT_CONST_AS T *)p; }
AUKIT_DEV Seg load_seg(const Seg *segs, unsigned sidx) {
    const AUKIT_CONST_AS unsigned long long *q = (const AUKIT_CONST_AS unsigned long long *)(segs + sidx);
    union { unsigned long long u[5]; Seg s; } x;
#pragma unroll
    for (int i = 0; i < 5; i++) x.u[i] = q[i];
    return x.s;
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
    else { sidx = as_const(P.tile_seg)[t]; tin = t - as_const(P.seg_tile0)[sidx]; }
    const Seg sg = load_seg(P.segs, sidx);
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
    if constexpr (SRC == SRC_AUDIO_F32 || SRC == SRC_I32) w.base = P.src + 4 * (size_t)as_const(P.src_off)[sg.stream] + 4 * sg.src_base;
    else w.base = P.src + (size_t)as_const(P.src_off)[sg.stream] + (long long)T::BYTES * sg.src_base;
    const unsigned char *a0 = w.base + (long long)T::BYTES * w.k_lo;
    w.al = (const unsigned char *)((uintptr_t)a0 & ~(uintptr_t)15);
    w.head = (int)(a0 - w.al) / T::BYTES;
    w.nvec = (w.head + w.n_stage + T::SPV - 1) / T::SPV;
    w.orow = reinterpret_cast<float *>(P.out) + sg.out_off + o0;
    return w;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NV>
AUKIT_DEV void issue_loads(const ResampleParams &P, const WaveTile &w, int lane, uint4 (&pre)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int v = lane + 64 * i;
        const unsigned char *p = w.al + 16 * (size_t)v;
        pre[i] = make_uint4(0, 0, 0, 0);
        if (v < w.nvec && p >= P.safe_lo && p + 16 <= P.safe_hi) {
            const u32x4 r = *(const __attribute__((address_space(1))) u32x4 *)p;  // one global_load_dwordx4 (a uint4 deref goes through a generic reference: flat_load)
            pre[i] = make_uint4(r.x, r.y, r.z, r.w);
        }
    }
}

// Where the wave waits for the next tile's loads.  hipcc waits for `pre` where it is first used — at the top of the tile loop, right
// after the sixteen row stores of the tile before.  On gfx9 loads and stores share vmcnt and return out of order with respect to
// each other, so that wait can only be vmcnt(0): every wave sat out the write latency of the stores it had just issued before it
// touched loads that had landed long ago (a hand-counted vmcnt(16) there returns garbage — tried).  The kernels therefore keep a
// tile's sixteen results in registers, "use" the NEXT tile's loads here (an empty asm that takes them as operands: hipcc puts its
// vmcnt(0) in front of it — the loads were issued before the tile's arithmetic, a whole tile of work ago, and the only stores still
// counted are the tile-before's, a tile older still) and only then issue the stores, which drain while the next tile is staged and
// evaluated.  The results pass through the same statements as in/out operands, so the stores (which need them) stay below; no
// "memory" clobber (with one hipcc loses the address space of the kernel-argument pointers and turns every load into flat_load).
// hipcc's waitcnt pass merges the states of every edge into the tile loop's header: `pre` must be "used" (waited for) on ALL of them —
// before the loop for the first tile, and on the partial-tile path — or its vmcnt(0) lands at the top of the loop after all.
template <int NV>
AUKIT_DEV void pre_landed(uint4 (&pre)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i++) asm volatile("" : "+v"(pre[i].x), "+v"(pre[i].y), "+v"(pre[i].z), "+v"(pre[i].w));
}
template <int NV>
AUKIT_DEV void loads_landed(uint4 (&pre)[NV], float (&res)[16]) {
    static_assert(NV == 1 || NV == 2 || NV == 4, "NV");
    if constexpr (NV == 1) asm volatile("" : "+v"(pre[0].x), "+v"(pre[0].y), "+v"(pre[0].z), "+v"(pre[0].w), "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]));
    else if constexpr (NV == 2)
        asm volatile("" : "+v"(pre[0].x), "+v"(pre[0].y), "+v"(pre[0].z), "+v"(pre[0].w), "+v"(pre[1].x), "+v"(pre[1].y), "+v"(pre[1].z), "+v"(pre[1].w), "+v"(res[0]), "+v"(res[1]), "+v"(res[2]),
                     "+v"(res[3]));
    else {
        asm volatile("" : "+v"(pre[0].x), "+v"(pre[0].y), "+v"(pre[0].z), "+v"(pre[0].w), "+v"(pre[1].x), "+v"(pre[1].y), "+v"(pre[1].z), "+v"(pre[1].w));
        asm volatile("" : "+v"(pre[2].x), "+v"(pre[2].y), "+v"(pre[2].z), "+v"(pre[2].w), "+v"(pre[3].x), "+v"(pre[3].y), "+v"(pre[3].z), "+v"(pre[3].w), "+v"(res[0]), "+v"(res[1]), "+v"(res[2]),
                     "+v"(res[3]));
    }
    asm volatile("" : "+v"(res[4]), "+v"(res[5]), "+v"(res[6]), "+v"(res[7]), "+v"(res[8]), "+v"(res[9]), "+v"(res[10]), "+v"(res[11]), "+v"(res[12]), "+v"(res[13]), "+v"(res[14]), "+v"(res[15]));
}

// results that must not be stored before this point (the statement is ordered after every asm volatile in front of it, and the stores need its outputs)
template <int N>
AUKIT_DEV void hold_results(float (&res)[N]) {
    static_assert(N % 8 == 0, "N");
#pragma unroll
    for (int i = 0; i < N; i += 8)
        asm volatile("" : "+v"(res[i]), "+v"(res[i + 1]), "+v"(res[i + 2]), "+v"(res[i + 3]), "+v"(res[i + 4]), "+v"(res[i + 5]), "+v"(res[i + 6]), "+v"(res[i + 7]));
}

template <int SRC>
AUKIT_DEV float sample_at(const ResampleParams &P, const FastParams &F, const unsigned char *q) {
    if constexpr (SRC == SRC_PCM_S16LE_MONO) {
        short s = (short)(q[0] | q[1] << 8);
        return (float)s * (s < 0 ? F.scale_neg : F.scale_pos);
    } else if constexpr (SRC == SRC_G711_MONO) {
        return g711_f32b(*q, P.ulaw, (float)P.g711_scale);
    } else if constexpr (SRC == SRC_PCM8_MONO) {
        return pcm8_f32(*q, P.data_type == AUKIT_UNSIGNED);
    } else if constexpr (SRC == SRC_I8) {
        return (float)(signed char)*q;
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
        } else if constexpr (SRC == SRC_PCM8_MONO) {
            const unsigned ww[4] = {u.x, u.y, u.z, u.w};
            const int us = P.data_type == AUKIT_UNSIGNED;
            float4 *o = reinterpret_cast<float4 *>(sm + 16 * v);
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = make_float4(pcm8_f32(ww[e] & 0xFF, us), pcm8_f32((ww[e] >> 8) & 0xFF, us), pcm8_f32((ww[e] >> 16) & 0xFF, us), pcm8_f32(ww[e] >> 24, us));
        } else if constexpr (SRC == SRC_I8) {
            const unsigned ww[4] = {u.x, u.y, u.z, u.w};
            float4 *o = reinterpret_cast<float4 *>(sm + 16 * v);
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = make_float4((float)(signed char)(ww[e] & 0xFF), (float)(signed char)((ww[e] >> 8) & 0xFF), (float)(signed char)((ww[e] >> 16) & 0xFF), (float)(signed char)(ww[e] >> 24));
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

// The full-tile path of k_fast_wave: the wave is VALU-issue-bound (profiles/: 32 VALU instructions per 64 outputs, two of them
// quarter-rate integer multiplies), so the position arrives as (q, rem) kept by additions, fx is one multiply (|error| <= 6e-8,
// the tolerance path allows 1e-6 RMS), the clamp is one v_med3, and for sources whose samples lie in [-1, 1] the `x % 1 == 0`
// copy (:666) needs no select: rem == 0 makes fx == 0 and the Horner form returns p1 exactly.
template <int SRC, int INTERP>
AUKIT_DEV float interp_qr(const FastParams &F, const float *tab, unsigned q, unsigned rem) {
    const float fx = (float)rem * F.inv_b;
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
    const float c = __builtin_amdgcn_fmed3f(v, -1.0f, 1.0f);  // aukit.lua:667-668
    if constexpr (SRC == SRC_PCM_S16LE_MONO || SRC == SRC_G711_MONO || SRC == SRC_PCM8_MONO) return c;
    else return rem == 0 ? p1 : c;
}

template <int INTERP, bool CLAMP = true>
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
    if constexpr (!CLAMP) return rem == 0 ? p1 : v;            // stream.pcm uses the interpolated sample as is (aukit.lua:2397-2400)
    else return rem == 0 ? p1 : fminf(fmaxf(v, -1.0f), 1.0f);  // aukit.lua:667-668
}

}  // namespace aukit
