// Host property test of csrc/mlx_seqfold.h (the segmented grid fold): the model of the wave algorithm against the plain loop
// `for (i) s += t[i]` on random and adversarial vectors -- every result must match BIT FOR BIT. Built and run by tests/test_seqfold.py.
//   seqfold_host <vectors per kind (short)> <vectors per kind (long)> <K> <seed>
// Prints one line per kind: vectors, mismatches, sub-blocks, failed checks, literal sub-blocks.
#include "mlx_seqfold.h"
#include <stdio.h>
#include <stdlib.h>
#include <random>
#include <vector>

static double seq(double s, const double *t, size_t n) { for (size_t i = 0; i < n; i++) s = s + t[i]; return s; }

int main(int argc, char **argv)
{
    const long nshort = argc > 1 ? atol(argv[1]) : 1000, nlong = argc > 2 ? atol(argv[2]) : 10;
    const int K = argc > 3 ? atoi(argv[3]) : SGF_K;
    std::mt19937_64 rng(argc > 4 ? (uint64_t)atoll(argv[4]) : 12345u);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    static const char *names[] = {"positive (a dot of squares)", "mixed signs (random walk)", "wide dynamic range", "few mantissa bits (ties)",
                                  "alternating cancellation", "powers of two", "drift with rare large negative steps", "tiny (near subnormal)",
                                  "one huge term", "signed zeros", "sum in [2^53, 2^54): every 1.0 a tie", "nan / inf inside", "sign flips through zero",
                                  "squares after a large start", "exact binade hits"};
    const int NK = 15;
    long total_fail = 0;
    std::vector<double> t;
    for (int kind = 0; kind < NK; kind++) {
        long fails = 0, viol = 0, lit = 0, blocks = 0, vecs = 0;
        for (long rep = 0; rep < nshort + nlong; rep++) {
            const bool lg = rep >= nshort;
            const size_t n = lg ? (size_t)(20000 + U(rng) * 60000) : (size_t)(1 + U(rng) * 300);
            t.resize(n);
            double s0 = 0.0;
            for (size_t i = 0; i < n; i++) {
                switch (kind) {
                case 0: { const double x = N(rng); t[i] = x * x; break; }
                case 1: t[i] = N(rng); break;
                case 2: t[i] = N(rng) * N(rng) * exp(8 * N(rng)); break;
                case 3: t[i] = (double)(int)(U(rng) * 8) * 0.5; break;
                case 4: t[i] = (i % 2 ? -1.0 : 1.0) * (1.0 + 1e-9 * U(rng)); break;
                case 5: t[i] = ldexp(1.0, -(int)(U(rng) * 60)); break;
                case 6: t[i] = (U(rng) < 0.01) ? -50.0 * U(rng) : U(rng); break;
                case 7: t[i] = U(rng) * U(rng) * 1e-300; break;
                case 8: t[i] = (i == n / 2) ? 1e300 : N(rng); break;
                case 9: t[i] = (U(rng) < 0.5 ? 0.0 : -0.0); break;
                case 10: t[i] = 1.0; break;
                case 11: t[i] = (U(rng) < 0.002) ? (U(rng) < 0.5 ? NAN : INFINITY) : U(rng); break;
                case 12: t[i] = sin(0.01 * (double)i) * (1.0 + U(rng)); break;
                case 13: { const double x = N(rng); t[i] = x * x * 1e-6; break; }
                default: t[i] = (i % 7 == 0) ? ldexp(1.0, (int)(U(rng) * 10)) : -ldexp(1.0, (int)(U(rng) * 8)); break;
                }
            }
            if (kind == 3) s0 = ldexp(1.0, 53);
            if (kind == 5) s0 = 1.0;
            if (kind == 9) s0 = -0.0;
            if (kind == 10) s0 = ldexp(1.0, 53);
            if (kind == 13) s0 = 12345.678;
            if (kind == 14) s0 = 1024.0;
            int v = 0, l = 0;
            const double a = sgf_model_fold(s0, t.data(), n, K, &v, &l), b = seq(s0, t.data(), n);
            vecs++; viol += v; lit += l; blocks += (long)((n + K - 1) / K);
            if (memcmp(&a, &b, 8) != 0) {
                if (fails < 5) printf("MISMATCH kind %d (%s) rep %ld n %zu: %.17g vs %.17g\n", kind, names[kind], rep, n, a, b);
                fails++;
            }
        }
        printf("kind %2d %-40s vectors %8ld mismatches %ld sub-blocks %10ld failed-checks %9ld literal %10ld\n", kind, names[kind], vecs, fails, blocks, viol, lit);
        total_fail += fails;
    }
    printf("TOTAL mismatches %ld\n", total_fail);
    return total_fail != 0;
}
