// resample_dev.h — device-side position arithmetic and interpolate.{none,linear,cubic,sinc} in the reference's
// fp64 operation order (aukit.lua:253-282), shared by resample.hip and codecs.hip.
#pragma once
#include "resample.h"

namespace aukit {

template <int INTERP> struct HaloOf { static constexpr int L = 0, R = 0; };
template <> struct HaloOf<AUKIT_INTERP_LINEAR> { static constexpr int L = 0, R = 1; };
template <> struct HaloOf<AUKIT_INTERP_CUBIC> { static constexpr int L = 1, R = 2; };

AUKIT_DEV double pos_of(const ResampleParams &P, unsigned o) {
    double n = (double)(P.pos_mul > 1 ? (unsigned long long)o * (unsigned)P.pos_mul : (unsigned long long)o);  // (i - 1)
    double q = P.exact_rcp ? div_rcp(n, P.ratio, P.rcp) : n / P.ratio;
    return q + 1.0;
}

// ------------------------------------------------------------------ one interpolated sample
// `tab` points at the LDS slot of table index k_lo for this channel.  Returns
// `if x % 1 == 0 then d[x] else interp(d, x)`; *isint tells the caller which branch was taken.
// SKEW16: the table is stored with one pad slot per 16 entries (slot(i) = i + i/16) so that lanes which each own 16
// consecutive entries write to distinct LDS banks (k_ima_stream).
// TAB_T: double, or float for tables whose samples are exact in f32 (G.711: multiples of 1/64) — converted on read.
// CV: what a table entry means — the identity, or (k_ima_stream_f32) the int16 predictor p kept as an exact float, read as the
// reference's sample p / (p < 0 and 128 or 127) (aukit.lua:2812)
struct CvIdentity { static AUKIT_DEV double cv(double v) { return v; } };
struct CvIma8 { static AUKIT_DEV double cv(double p) { return p < 0 ? p * (1.0 / 128) : div_rcp(p, 127.0, 1.0 / 127.0); } };
template <int INTERP, bool SKEW16 = false, typename TAB_T = double, typename CV = CvIdentity>
AUKIT_DEV double eval_at(const ResampleParams &P, const Seg &sg, const TAB_T *tabp, int k_lo, unsigned o, bool *isint) {
    struct { const TAB_T *p; AUKIT_DEV double operator[](int i) const { return CV::cv((double)(SKEW16 ? p[i + (i >> 4)] : p[i])); } } tab{tabp};
    double x = pos_of(P, o);
    double ffx = floor(x);
    int k = (int)ffx;
    k = k < sg.w_lo ? sg.w_lo : (k > sg.w_hi ? sg.w_hi : k);  // host guarantees w_lo <= k <= w_hi; keeps LDS reads in range
    int idx = k - k_lo;
    *isint = (x == ffx);  // x % 1 == 0
    double p1 = tab[idx];
    if (*isint) return p1;
    double fx = x - ffx;
    if constexpr (INTERP == AUKIT_INTERP_NONE) {
        return p1;  // data[math.floor(x)]  :254-256
    } else if constexpr (INTERP == AUKIT_INTERP_LINEAR) {
        int i1 = (k + 1 <= sg.w_hi) ? idx + 1 : idx;  // data[ffx+1] or data[ffx]
        return linear_exact(p1, tab[i1], fx);
    } else if constexpr (INTERP == AUKIT_INTERP_CUBIC) {
        int i0 = (k - 1 >= sg.w_lo) ? idx - 1 : idx;      // p0 or p1
        int i2 = (k + 1 <= sg.w_hi) ? idx + 1 : idx;      // p2 or p1
        int i3 = (k + 2 <= sg.w_hi) ? idx + 2 : i2;       // p3 or p2 or p1
        return cubic_exact(tab[i0], p1, tab[i2], tab[i3], fx);
    } else {  // sinc  :267-281 (sin() is libm-dependent: tolerance-level parity only)
        double sum = 0;
        const double pi = 3.14159265358979323846;
        for (int n = -P.sinc_w; n <= P.sinc_w; n++) {
            int w = k + n;
            int sft = 0;
            if (P.sinc_hole) {   // reference index w <= -1 is the row's entry w + 1 (the block before, shifted by the hole at 0)
                if (w == 0) continue;
                if (w < 0) { sft = 1; w++; }
            }
            if (w >= sg.w_lo && w <= sg.w_hi) {
                double d = tab[idx + n + sft];
                double px = pi * (fx - n);
                if (px == 0) sum = sum + d;
                else sum = sum + d * sin(px) / px;
            }
        }
        return sum;
    }
}

// interpolate.sinc (:267-281) on a table whose indices lo .. hi hold numbers (everything else is nil and skipped): the block tables of stream.flac
// and stream.qoa, indices -1 and 0 being the two samples kept from the block before (:3170-3171, :3312).  `tap(i)` reads entry i.
template <class Tap> AUKIT_DEV double sinc_at(const Tap &tap, int k, int lo, int hi, double fx, int w) {
    double sum = 0;
    const double pi = 3.14159265358979323846;
    for (int m = -w; m <= w; m++) {
        const int idx = k + m;
        if (idx >= lo && idx <= hi) {
            const double d = tap(idx);
            const double px = pi * (fx - m);
            if (px == 0) sum = sum + d;
            else sum = sum + d * sin(px) / px;
        }
    }
    return sum;
}

}  // namespace aukit
