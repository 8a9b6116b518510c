// GPU self-test of ml-ease_amd/csrc/mlx_wave.h: every VALU butterfly against its __shfl_xor form, bit for bit, on random doubles
// (incl. zeros, denormals, huge values and a NaN round for max). Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off
// tools/wave_selftest.hip -o tools/wave_selftest (the Makefile in csrc does it); exit code 0 = all equal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include "../ml-ease_amd/csrc/mlx_wave.h"

__device__ __forceinline__ double ref_wave_sum(double x) { for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64); return x; }
__device__ __forceinline__ double ref_wave_max(double x) { for (int m = 32; m >= 1; m >>= 1) x = fmax(x, __shfl_xor(x, m, 64)); return x; }
__device__ __forceinline__ double ref_group8(double x) { for (int m = 4; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64); return x; }

__global__ void k_test(const double *in, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double x = in[i];
    out[0 * n + i] = ref_wave_sum(x);   out[1 * n + i] = mlx_wave_allreduce_sum(x);
    out[2 * n + i] = ref_wave_max(x);   out[3 * n + i] = mlx_wave_allreduce_max(x);
    out[4 * n + i] = ref_group8(x);     out[5 * n + i] = mlx_group8_allreduce_sum(x);
    out[6 * n + i] = __shfl_xor(x, 4, 64); out[7 * n + i] = mlx_xor4(x);
    out[8 * n + i] = __shfl_xor(x, 8, 64); out[9 * n + i] = mlx_xor8(x);
    out[10 * n + i] = __shfl_xor(x, 2, 64); out[11 * n + i] = mlx_xor2(x);
    out[12 * n + i] = __shfl_xor(x, 1, 64); out[13 * n + i] = mlx_xor1(x);
    // the swap pairs: {a, b} == {x, x of lane i ^ m} as bit patterns
    double a, b;
    const long long xi = __double_as_longlong(x);
    mlx_swap32(x, a, b);
    const long long p32 = __double_as_longlong(__shfl_xor(x, 32, 64)), a32 = __double_as_longlong(a), b32 = __double_as_longlong(b);
    out[14 * n + i] = 1.0; out[15 * n + i] = ((a32 == xi && b32 == p32) || (a32 == p32 && b32 == xi)) ? 1.0 : 0.0;
    mlx_swap16(x, a, b);
    const long long p16 = __double_as_longlong(__shfl_xor(x, 16, 64)), a16 = __double_as_longlong(a), b16 = __double_as_longlong(b);
    out[16 * n + i] = 1.0; out[17 * n + i] = ((a16 == xi && b16 == p16) || (a16 == p16 && b16 == xi)) ? 1.0 : 0.0;
}

int main()
{
    const int n = 64 * 1024;
    std::vector<double> h(n);
    std::mt19937_64 rng(12345);
    std::normal_distribution<double> nd(0.0, 1.0);
    for (int i = 0; i < n; i++) {
        double v = nd(rng);
        const int kind = (i / 64) % 8;
        if (kind == 1) v *= 1e300; else if (kind == 2) v *= 1e-310; else if (kind == 3 && (i % 5) == 0) v = 0.0;
        else if (kind == 4) v = (double)(float)v * 1e8; else if (kind == 5 && (i % 64) == 17) v = __builtin_nan("");
        h[i] = v;
    }
    double *din, *dout;
    if (hipMalloc(&din, n * sizeof(double)) != hipSuccess || hipMalloc(&dout, 18 * (size_t)n * sizeof(double)) != hipSuccess) { printf("no device memory\n"); return 2; }
    (void)hipMemcpy(din, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
    std::vector<double> o(18 * (size_t)n);
    if (hipMemcpy(o.data(), dout, o.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
    const char *names[9] = {"wave sum", "wave max", "group-of-8 sum", "xor 4", "xor 8", "xor 2", "xor 1", "swap32 pair", "swap16 pair"};
    int bad = 0;
    for (int t = 0; t < 9; t++) {
        int diff = 0;
        for (int i = 0; i < n; i++) {
            uint64_t a, b;
            memcpy(&a, &o[(2 * t) * (size_t)n + i], 8); memcpy(&b, &o[(2 * t + 1) * (size_t)n + i], 8);
            const bool both_nan = (o[(2 * t) * (size_t)n + i] != o[(2 * t) * (size_t)n + i]) && (o[(2 * t + 1) * (size_t)n + i] != o[(2 * t + 1) * (size_t)n + i]);
            if (a != b && !both_nan) diff++;
        }
        printf("%-16s %s (%d of %d lanes differ)\n", names[t], diff ? "DIFFERS" : "bit-identical", diff, n);
        bad += diff != 0;
    }
    return bad ? 1 : 0;
}
