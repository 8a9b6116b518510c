// f64_rate_probe.hip -- issue rate and dependent latency of the fp64 VALU operations the reference-order folds are made of (round 6:
// a clean 1 024-term chunk of sgf_wave_fold costs the folding wave ~3 500 cycles for ~300 instructions; which of them are slow?).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/f64_rate_probe.hip -o tools/f64_rate_probe
// One wave, shader clock (clock64): N independent accumulators per operation -> cycles per instruction at N = 1 (latency) and N = 8
// (issue rate); then 1, 2, 4 waves on ONE SIMD's worth of a workgroup (a 256-thread workgroup puts one wave on each SIMD, a 1024-thread
// one four) to see what waves sharing a SIMD cost each other.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 2000
template <int OP, int N>
__global__ void k_rate(double *out, long long *cyc, double seed)
{
#pragma clang fp contract(off)
    double a[N];
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = seed + i * 0.125 + threadIdx.x * 1e-3;
    const double b = seed * 0.5 + 1.0, c = 1.0000001;
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (OP == 0) a[i] = a[i] + b;
            else if (OP == 1) a[i] = a[i] * c;
            else if (OP == 2) a[i] = __builtin_fma(a[i], c, b);
            else if (OP == 3) a[i] = fmax(a[i], b + it);
            else if (OP == 4) a[i] = fmin(a[i], b - it);
            else if (OP == 5) a[i] = (a[i] < b + it) ? a[i] + 1.0 : a[i];     // compare + select + add
            else if (OP == 6) a[i] = fabs(a[i]) - b;                           // add with a source modifier
            else if (OP == 7) { a[i] = (a[i] + b) - b; }                       // two dependent adds (the magic-constant rounding)
        }
    }
    const long long t1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int OP, int N>
static double run(int threads, double *out, long long *cyc)
{
    hipLaunchKernelGGL((k_rate<OP, N>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.5);
    hipLaunchKernelGGL((k_rate<OP, N>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.5);
    CK(hipDeviceSynchronize());
    long long c; CK(hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost));
    const int per = (OP == 5 || OP == 7) ? 2 : 1;        // VALU fp64-class instructions per "operation" of the loop (5: cmp + add; 7: two adds)
    return (double)c / ((double)ITERS * N * per);
}
int main()
{
    double *out; long long *cyc;
    CK(hipMalloc(&out, 8 * 4096)); CK(hipMalloc(&cyc, 64));
    const char *names[8] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_min_f64", "v_cmp+cndmask+add (per 2)", "v_add_f64 |src|", "(x+M)-M (per add)"};
    printf("{\"shader_cycles_per_instruction\": {\n");
#define ROW(OP) printf(" \"%s\": {\"1_chain_1_wave\": %.2f, \"8_chains_1_wave\": %.2f, \"8_chains_4_waves_4_simds\": %.2f, \"8_chains_16_waves_4_per_simd\": %.2f}%s\n", names[OP], \
        run<OP, 1>(64, out, cyc), run<OP, 8>(64, out, cyc), run<OP, 8>(256, out, cyc), run<OP, 8>(1024, out, cyc), OP == 7 ? "" : ",")
    ROW(0); ROW(1); ROW(2); ROW(3); ROW(4); ROW(5); ROW(6); ROW(7);
    printf("}}\n");
    return 0;
}
