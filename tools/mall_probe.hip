// mall_probe.hip -- how fast is a streaming read served by the 256 MiB Infinity Cache, compared with HBM? (round 6: would a second read of
// a dense tile band right behind the first be cheaper than the first?)  hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o tools/mall_probe
// For each buffer size: one untimed pass (fills the cache), then REPS timed passes over the same buffer -- a buffer below the cache
// size is then served on-die. Also: a leader / trailer pair, the trailer re-reading what the leader read DIST bytes earlier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) k_read(const f4 *__restrict__ p, size_t n16, float *out)
{
    f4 a = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const size_t j = i + u * 256; v[u] = j < n16 ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : a; }
#pragma unroll
        for (int u = 0; u < 4; u++) a += v[u];
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main()
{
    const size_t MAXB = (size_t)4 << 30;
    f4 *buf; float *out;
    CK(hipMalloc(&buf, MAXB)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, MAXB));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int REPS = 10;
    printf("{\"same_buffer_repeated\": [\n");
    const size_t sizes[] = {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 4096};
    for (int nt = 0; nt < 2; nt++)
        for (size_t s : sizes) {
            const size_t bytes = s << 20, n16 = bytes / 16;
            const int grid = 256 * 8;
            if (nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, buf, n16, out); else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, buf, n16, out);
            hipEventRecord(e0, 0);
            for (int r = 0; r < REPS; r++) { if (nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, buf, n16, out); else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, buf, n16, out); }
            hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf(" {\"MiB\": %zu, \"nontemporal\": %d, \"GB_per_s\": %.1f, \"us_per_pass\": %.1f},\n", s, nt, bytes * REPS / (ms * 1e6), ms * 1e3 / REPS);
        }
    printf(" null],\n\"band_pairs\": [\n");
    // band pairs: read band b (leader), then band b again (trailer), b = 0 .. over a 4 GiB buffer: what two passes per band cost against two passes over everything
    for (size_t band : {64, 128, 192, 256, 512}) {
        const size_t bb = band << 20, n16 = bb / 16, nb = MAXB / bb;
        hipEventRecord(e0, 0);
        for (size_t b = 0; b < nb; b++) {
            hipLaunchKernelGGL(k_read<false>, dim3(2048), dim3(256), 0, 0, buf + b * n16, n16, out);
            hipLaunchKernelGGL(k_read<false>, dim3(2048), dim3(256), 0, 0, buf + b * n16, n16, out);
        }
        hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" {\"band_MiB\": %zu, \"two_reads_of_4GiB_ms\": %.3f, \"GB_per_s_counting_both_reads\": %.1f},\n", band, ms, 2.0 * MAXB / (ms * 1e6));
    }
    printf(" null]}\n");
    return 0;
}
