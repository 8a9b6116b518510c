// mfma_f64_probe.hip -- practical ceiling of v_mfma_f64_16x16x4_f64 on this part: 16 independent accumulator tiles per
// wave, operands in registers, no memory traffic. Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(double *out, int iters, double seed)
{
    d4_t acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (d4_t){0, 0, 0, 0};
    double a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = seed + threadIdx.x * 1e-3 + i; b[i] = seed - threadIdx.x * 1e-3 - i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n < 4; n++) acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m * 4 + n], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES> void run(int wgs, int iters)
{
    double *d; hipMalloc(&d, sizeof(double) * wgs * 64 * WAVES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WAVES>, dim3(wgs), dim3(64 * WAVES), 0, 0, d, 10, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WAVES>, dim3(wgs), dim3(64 * WAVES), 0, 0, d, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * WAVES * iters * 16 * 2048.0;
    printf("waves/WG %d, WGs %d: %.3f ms, %.1f TFLOP/s\n", WAVES, wgs, ms, flop / ms / 1e9);
    hipFree(d);
}
int main()
{
    run<4>(256, 20000);      // 1 wave per SIMD
    run<4>(512, 20000);      // 2 waves per SIMD
    run<8>(256, 20000);
    run<4>(1024, 20000);     // 4 waves per SIMD
    return 0;
}
