/* ork_check.c — host-side checks of the exact-arithmetic shortcuts the HIP kernels rely on (test infrastructure only).
 * The kernels replace fp64 divisions by q = fma(fma(-d, n*r, n), r, n*r) with r = RN(1/d) (Markstein's correction
 * step).  These functions compare that expression with the true IEEE quotient over whole ranges. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static inline double div_rcp(double a, double b, double r) {
    double q0 = a * r;
    double e = fma(-b, q0, a);
    return fma(e, r, q0);
}
/* number of integers n in [0, count) for which the shortcut differs from n / d */
uint64_t ork_check_div_rcp(double d, uint64_t count) {
    const double r = 1.0 / d;
    uint64_t bad = 0;
    for (uint64_t n = 0; n < count; n++)
        if (div_rcp((double)n, d, r) != (double)n / d) bad++;
    return bad;
}
/* correctly rounded fx^3 via a double-double product vs long-double reference (returns mismatches) */
static inline double pow3_rn(double fx) {
    double hi = fx * fx, lo = fma(fx, fx, -hi);
    double p = hi * fx, pl = fma(hi, fx, -p);
    double t = fma(lo, fx, pl);
    return p + t;
}
uint64_t ork_check_pow3(uint64_t count, uint64_t seed) {
    uint64_t bad = 0, s = seed ? seed : 88172645463325252ULL;
    for (uint64_t i = 0; i < count; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double fx = (double)(s >> 11) / 9007199254740992.0;
        __float128 ex = (__float128)fx * fx * fx;  /* 113-bit: exact enough to decide the rounding of a 53-bit result except in 2^-60 cases */
        double ref = (double)ex;
        if (pow3_rn(fx) != ref) bad++;
    }
    return bad;
}
