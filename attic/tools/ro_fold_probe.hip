// Probe (GPU): cycles per term of the sequential fold of csrc/mlx_ro_kernels.h -- the asm loop ro_fold32, a plain C loop, and the
// dependent v_add_f64 chain alone (terms in registers). One workgroup, six folding lanes like pass B of k_ro_step.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ml-ease_amd/csrc attic/tools/ro_fold_probe.hip -o /tmp/ro_fold_probe && /tmp/ro_fold_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <mutex>
#include <type_traits>
template <typename T> __device__ __forceinline__ T gld(const T *p) { return *p; }
template <typename T> __device__ __forceinline__ void gst(T *p, T v) { *p = v; }
#include "mlx_types.h"
#define MLX_RO_PROBE
#include "mlx_ro_kernels.h"

__global__ void __launch_bounds__(256) k_probe(double *out, long long *cyc, int reps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RoLds &sh = *reinterpret_cast<RoLds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < RO_NF; k++) for (int i = tid; i < RO_CHP; i += 256) sh.C[0][k][i] = 1.0 / (double)(1 + i + 7 * k);
    __syncthreads();
    if (wave == 0 && lane < 6) {
        const double *cp = &sh.C[0][lane][0];
        double s = 0.0;
        long long t0 = clock64();
        for (int r = 0; r < reps; r++) s = ro_fold32(s, cp, 32);
        long long t1 = clock64();
        double s2 = 0.0;
        for (int r = 0; r < reps; r++) for (int i = 0; i < 1024; i++) s2 = s2 + cp[i];
        long long t2 = clock64();
        double s3 = 0.0, x = cp[lane];
        for (int r = 0; r < reps * 1024; r += 16) {
#pragma unroll
            for (int u = 0; u < 16; u++) { s3 = s3 + x; asm volatile("" : "+v"(s3)); }
        }
        long long t3 = clock64();
        if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
        out[lane] = s; out[8 + lane] = s2; out[16 + lane] = s3;
    }
}

int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 24 * sizeof(double)); hipMalloc(&cyc, 3 * sizeof(long long));
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RoLds));
    const int reps = 64;
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), sizeof(RoLds), 0, out, cyc, reps);
    hipDeviceSynchronize();
    double h[24]; long long c[3];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    const double n = 1024.0 * reps;
    printf("clock64 ticks per term: ro_fold32 %.2f   plain loop %.2f   add chain in registers %.2f   (sums equal: %d)\n", c[0] / n, c[1] / n, c[2] / n,
           h[0] == h[8]);
    int khz = 0, ckhz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0); hipDeviceGetAttribute(&ckhz, hipDeviceAttributeClockRate, 0);
    printf("wall clock %d kHz, shader clock %d kHz (clock64 = s_memtime: shader-clock domain on gfx9)\n", khz, ckhz);
    return 0;
}
