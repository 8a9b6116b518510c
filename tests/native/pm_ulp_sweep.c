/* Direct accuracy sweep of ml-ease_amd/csrc/portable_math.h against the host libm, in units in the last place: the
 * reference-order contract ships pm_exp / pm_log1p in BOTH the kernels and the oracle twin, so "bit-identical to the twin" pins the
 * summation order but not these two functions. Built and run by tests/test_oracle.py.
 *   pm_ulp_sweep <points per region>
 * exp: x in [-745, 709] (uniform, plus a dense band around 0 and the logistic range [-40, 40]); log1p: u in (-1, 1e308) (log-uniform
 * over both signs, plus (-1, -0.5], the band around 0, and u = exp(-z) for z in [0, 40] -- what row_eval feeds it).
 * Prints: function, region, points, max ulp error, where. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "portable_math.h"

static double ulp_err(double got, double want)
{
    if (got == want) return 0.0;
    if (isnan(got) || isnan(want) || isinf(got) || isinf(want)) return (isnan(got) && isnan(want)) ? 0.0 : 1e300;
    int e;
    frexp(want, &e);
    double u = ldexp(1.0, e - 53);
    if (u < 4.9406564584124654e-324) u = 4.9406564584124654e-324;      /* subnormal results: the fixed spacing 2^-1074 */
    return fabs(got - want) / u;
}
static uint64_t rs = 88172645463325252ULL;
static double urand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 200000;
    struct { const char *name; double lo, hi; int logspace; } er[] = {
        {"exp [-745,709]", -745.0, 709.0, 0}, {"exp [-40,40]", -40.0, 40.0, 0}, {"exp [-1,1]", -1.0, 1.0, 0}, {"exp [-1e-3,1e-3]", -1e-3, 1e-3, 0}};
    double worst = 0;
    for (unsigned r = 0; r < sizeof er / sizeof er[0]; r++) {
        double mx = 0, at = 0;
        for (long i = 0; i < n; i++) {
            const double x = er[r].lo + (er[r].hi - er[r].lo) * urand();
            const double e = ulp_err(pm_exp(x), exp(x));
            if (e > mx) { mx = e; at = x; }
        }
        printf("%-28s points %8ld max_ulp %.3f at %.17g\n", er[r].name, n, mx, at);
        if (mx > worst) worst = mx;
    }
    printf("exp_max_ulp %.3f\n", worst);
    double worst_l = 0;
    const char *ln[] = {"log1p +[1e-320,1e308] log", "log1p -[1e-320,0.5] log", "log1p (-1,-0.5]", "log1p [-1e-3,1e-3]", "log1p exp(-z), z in [0,40]", "log1p [0.5,4]"};
    for (int r = 0; r < 6; r++) {
        double mx = 0, at = 0;
        for (long i = 0; i < n; i++) {
            double u;
            switch (r) {
            case 0: u = pow(10.0, -320.0 + 628.0 * urand()); break;
            case 1: u = -pow(10.0, -320.0 + (log10(0.5) + 320.0) * urand()); break;
            case 2: u = -0.5 - 0.5 * urand(); if (u <= -1.0) u = -0.9999999999999999; break;
            case 3: u = -1e-3 + 2e-3 * urand(); break;
            case 4: u = exp(-40.0 * urand()); break;
            default: u = 0.5 + 3.5 * urand(); break;
            }
            const double e = ulp_err(pm_log1p(u), log1p(u));
            if (e > mx) { mx = e; at = u; }
        }
        printf("%-28s points %8ld max_ulp %.3f at %.17g\n", ln[r], n, mx, at);
        if (mx > worst_l) worst_l = mx;
    }
    printf("log1p_max_ulp %.3f\n", worst_l);
    /* the special values both sides must agree on exactly */
    const double sp[] = {0.0, -0.0, 709.782712893384, 709.79, -745.2, -745.13, -746.0, INFINITY, -INFINITY};
    int bad = 0;
    for (unsigned i = 0; i < sizeof sp / sizeof sp[0]; i++) {
        const double a = pm_exp(sp[i]), b = exp(sp[i]);
        if (!((a == b) || ulp_err(a, b) <= 2.0)) { printf("exp special %g: %g vs %g\n", sp[i], a, b); bad++; }
    }
    if (!(pm_log1p(-1.0) == -INFINITY) || !isnan(pm_log1p(-1.5)) || !(pm_log1p(INFINITY) == INFINITY) || !(pm_log1p(0.0) == 0.0) || !isnan(pm_exp(NAN)) || !isnan(pm_log1p(NAN))) { printf("special values differ\n"); bad++; }
    printf("specials_bad %d\n", bad);
    return 0;
}
