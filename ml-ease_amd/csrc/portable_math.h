/* portable_math.h -- exp and log1p from IEEE +,-,*,/ and integer bit operations only.
 *
 * Used by the ORDER-FAITHFUL VERIFICATION MODE only (MLX_FAITHFUL=1): there the HIP path sums in
 * the reference's sequential order, and the one thing left that could differ from the CPU oracle is the last bit of the
 * elementary functions (OCML on the device, glibc / StrictMath on the host). Both sides of that test therefore evaluate
 * these two functions instead (the oracle is compiled a second time with -DORC_PORTABLE_MATH including this very
 * file), which makes "HIP == oracle, bit for bit" a statement about summation order alone. Accuracy, measured against glibc by
 * tests/native/pm_ulp_sweep.c (tests/test_oracle.py): pm_exp within 1 ulp over [-745, 709], pm_log1p within 3 ulp over (-1, 1e308)
 * (Java's Math.exp / Math.log1p are specified to 1 ulp: against a JVM the reference-order contract is exact up to that). Every
 * operation is a single correctly rounded IEEE operation with contraction off, so device and host agree exactly.
 * Used by the reference-order numerics only, never by the fast contract's kernels.
 */
#ifndef MLX_PORTABLE_MATH_H
#define MLX_PORTABLE_MATH_H
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define PM_FN static __host__ __device__ __forceinline__
#else
#define PM_FN static inline
#endif

PM_FN double pm_from_bits(uint64_t b) { double d; memcpy(&d, &b, sizeof d); return d; }
PM_FN uint64_t pm_to_bits(double d) { uint64_t b; memcpy(&b, &d, sizeof b); return b; }

/* 2^k for k in [-1022, 1023] */
PM_FN double pm_pow2(int k) { return pm_from_bits((uint64_t)(k + 1023) << 52); }

PM_FN double pm_exp(double x)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (x != x) return x;
    if (x > 709.782712893384) return pm_from_bits(0x7FF0000000000000ULL);      /* +inf */
    if (x < -745.2) return 0.0;
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;   /* the fdlibm split: k * LN2_HI is exact for |k| < 2^20 */
    const double LN2_LO = 1.90821492927058770002e-10;
    const double t = x * LOG2E;
    const long long k = (long long)(t + (t >= 0 ? 0.5 : -0.5));                /* round to nearest, ties away */
    const double kd = (double)k;
    const double r = (x - kd * LN2_HI) - kd * LN2_LO;                          /* |r| <= 0.3466 + eps */
    /* exp(r) = sum r^i / i!, Horner, degree 13 (remainder < 2e-17 relative) */
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    /* scale by 2^k in two steps so that results in the subnormal range round once */
    const int k1 = (int)(k / 2), k2 = (int)(k - k / 2);
    return (p * pm_pow2(k1)) * pm_pow2(k2);
}

PM_FN double pm_log1p(double u)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (u != u) return u;
    if (u <= -1.0) return (u == -1.0) ? -pm_from_bits(0x7FF0000000000000ULL) : (0.0 / 0.0);
    const double au = u < 0 ? -u : u;
    if (au < 5.551115123125783e-17) return u;                                  /* |u| < 2^-54 */
    const double f = 1.0 + u;
    if (((pm_to_bits(f) >> 52) & 0x7FF) == 0x7FF) return f;                    /* +inf stays +inf (round 6: this line used to turn it into a NaN) */
    const double c = (u - (f - 1.0)) / f;                                      /* correction for the rounding of 1 + u */
    /* f = m * 2^e, m in [sqrt(1/2), sqrt(2)) */
    uint64_t b = pm_to_bits(f);
    int e = (int)((b >> 52) & 0x7FF) - 1023;
    b = (b & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double m = pm_from_bits(b);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    /* log(m) = 2 s (1 + s^2/3 + s^4/5 + ... + s^24/25), |s| <= 0.1716 */
    double q = 1.0 / 25.0;
    q = q * s2 + 1.0 / 23.0;
    q = q * s2 + 1.0 / 21.0;
    q = q * s2 + 1.0 / 19.0;
    q = q * s2 + 1.0 / 17.0;
    q = q * s2 + 1.0 / 15.0;
    q = q * s2 + 1.0 / 13.0;
    q = q * s2 + 1.0 / 11.0;
    q = q * s2 + 1.0 / 9.0;
    q = q * s2 + 1.0 / 7.0;
    q = q * s2 + 1.0 / 5.0;
    q = q * s2 + 1.0 / 3.0;
    q = q * s2 + 1.0;
    const double lm = 2.0 * s * q;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double ed = (double)e;
    return ed * LN2_HI + ((lm + c) + ed * LN2_LO);
}

#endif /* MLX_PORTABLE_MATH_H */
