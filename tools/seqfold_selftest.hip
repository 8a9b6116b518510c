// GPU self-test of csrc/mlx_seqfold.h: the wave code (sgf_wave_fold: DPP scan, ballots, the literal sub-blocks) against the plain
// sequential loops on the host, bit for bit -- plain sums `s += t[i]` and the euclideanNorm form `sum = c[i] + sum * m[i]`.
// One wave per vector. Built by csrc/Makefile, run by tests/test_gpu_parity.py (pytest -m gpu). Exit code 0 = every result identical.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <random>
#include <vector>
#include "../ml-ease_amd/csrc/mlx_seqfold.h"

template <bool MUL>
__global__ void __launch_bounds__(64) k_fold(const double *__restrict__ t, const double *__restrict__ m, const long *__restrict__ off, const double *__restrict__ s0, double *__restrict__ out, long long *__restrict__ cyc)
{
#pragma clang fp contract(off)
    const int v = blockIdx.x, lane = threadIdx.x;
    const long b0 = off[v], n = off[v + 1] - b0;
    double s = s0[v];
    int hostile = 0;
    long long ctot = 0, nch = 0;
    for (long base = 0; base < n; base += 64 * SGF_K) {
        double x[SGF_K], mm[SGF_K];
        bool hm = false;
#pragma unroll
        for (int i = 0; i < SGF_K; i++) {
            const long j = base + (long)lane * SGF_K + i;
            x[i] = j < n ? t[b0 + j] : -0.0;
            mm[i] = (MUL && j < n) ? m[b0 + j] : 1.0;
            hm = hm || (mm[i] != 1.0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long c0 = clock64();
        s = sgf_wave_fold<SGF_K, MUL>(s, x, mm, hm, hostile);
        ctot += clock64() - c0;
        nch++;
    }
    if (lane == 0) { out[v] = s; cyc[2 * v] = ctot; cyc[2 * v + 1] = nch; }
}

int main()
{
    std::mt19937_64 rng(2026);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    const int NK = 15, PER = 40;
    std::vector<double> t, m, s0;
    std::vector<long> off(1, 0);
    for (int kind = 0; kind < NK; kind++)
        for (int rep = 0; rep < PER; rep++) {
            const size_t n = rep < PER / 2 ? (size_t)(1 + U(rng) * 3000) : (size_t)(20000 + U(rng) * 60000);
            double st = 0.0;
            for (size_t i = 0; i < n; i++) {
                double x;
                switch (kind) {
                case 0: { const double z = N(rng); x = z * z; break; }
                case 1: x = N(rng); break;
                case 2: x = N(rng) * N(rng) * exp(8 * N(rng)); break;
                case 3: x = (double)(int)(U(rng) * 8) * 0.5; break;
                case 4: x = (i % 2 ? -1.0 : 1.0) * (1.0 + 1e-9 * U(rng)); break;
                case 5: x = ldexp(1.0, -(int)(U(rng) * 60)); break;
                case 6: x = (U(rng) < 0.01) ? -50.0 * U(rng) : U(rng); break;
                case 7: x = U(rng) * U(rng) * 1e-300; break;
                case 8: x = (i == n / 2) ? 1e300 : N(rng); break;
                case 9: x = (U(rng) < 0.5 ? 0.0 : -0.0); break;
                case 10: x = 1.0; break;
                case 11: x = (U(rng) < 0.002) ? (U(rng) < 0.5 ? NAN : INFINITY) : U(rng); break;
                case 12: x = sin(0.01 * (double)i) * (1.0 + U(rng)); break;
                case 13: { const double z = N(rng); x = z * z * 1e-6; break; }
                default: x = (i % 7 == 0) ? ldexp(1.0, (int)(U(rng) * 10)) : -ldexp(1.0, (int)(U(rng) * 8)); break;
                }
                t.push_back(x);
                // the multipliers of the norm form: 1.0 almost everywhere, a "scale change" (m < 1, c = 1) now and then
                m.push_back(U(rng) < 0.002 ? U(rng) * U(rng) : 1.0);
            }
            if (kind == 3 || kind == 10) st = ldexp(1.0, 53);
            if (kind == 5) st = 1.0;
            if (kind == 9) st = -0.0;
            if (kind == 13) st = 12345.678;
            if (kind == 14) st = 1024.0;
            s0.push_back(st);
            off.push_back((long)t.size());
        }
    const int nv = (int)s0.size();
    double *dt, *dm, *ds, *dout;
    long *doff;
    long long *dcyc;
    hipMalloc(&dcyc, (size_t)s0.size() * 16);
    hipMalloc(&dt, t.size() * 8); hipMalloc(&dm, m.size() * 8); hipMalloc(&ds, nv * 8); hipMalloc(&dout, nv * 8); hipMalloc(&doff, off.size() * 8);
    hipMemcpy(dt, t.data(), t.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dm, m.data(), m.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(ds, s0.data(), nv * 8, hipMemcpyHostToDevice); hipMemcpy(doff, off.data(), off.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> got(nv);
    long bad = 0;
    for (int mul = 0; mul < 2; mul++) {
        if (mul) hipLaunchKernelGGL(k_fold<true>, dim3(nv), dim3(64), 0, 0, dt, dm, doff, ds, dout, dcyc);
        else hipLaunchKernelGGL(k_fold<false>, dim3(nv), dim3(64), 0, 0, dt, dm, doff, ds, dout, dcyc);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
        hipMemcpy(got.data(), dout, nv * 8, hipMemcpyDeviceToHost);
        std::vector<long long> cy(2 * (size_t)nv);
        hipMemcpy(cy.data(), dcyc, (size_t)nv * 16, hipMemcpyDeviceToHost);
        for (int kind = 0; kind < NK; kind++) {                 // shader-clock cycles per chunk of 1 024 terms, the long vectors of each kind
            long long c = 0, n = 0;
            for (int rep = PER / 2; rep < PER; rep++) { c += cy[2 * (size_t)(kind * PER + rep)]; n += cy[2 * (size_t)(kind * PER + rep) + 1]; }
            printf("form %d kind %2d: %7.0f cycles per chunk\n", mul, kind, n ? (double)c / (double)n : 0.0);
        }
        for (int v = 0; v < nv; v++) {
            double s = s0[v];
            for (long j = off[v]; j < off[v + 1]; j++) s = mul ? t[j] + s * m[j] : s + t[j];
            // (the norm form with c = 1 where m != 1 is what euclideanNorm does; any c works for the test)
            if (memcmp(&s, &got[v], 8) != 0) {
                if (bad < 10) printf("MISMATCH form %d vector %d (kind %d, n %ld): device %.17g host %.17g\n", mul, v, v / PER, off[v + 1] - off[v], got[v], s);
                bad++;
            }
        }
    }
    printf("seqfold selftest: %d vectors x 2 forms, %ld mismatches\n", nv, bad);
    return bad ? 1 : 0;
}
