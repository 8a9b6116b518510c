// lds_gather_probe.hip -- what does a 64-lane ds_read_b64 gather cost as a function of its bank conflicts? (round 6: the column pass's
// random gathers of the staged coefficients are its largest item, `profiles/r6_notes.md` section 3.)
//   hipcc --offload-arch=gfx950 -O2 tools/lds_gather_probe.hip -o tools/lds_gather_probe
// One workgroup of 1024 threads (16 waves, as the pass), 128 KB of LDS; every lane gathers ITERS x 8 doubles from addresses of a fixed
// pattern: mult = how many lanes of a 32-lane half share one 8-byte bank pair (1 = conflict-free ... 32 = all on one), or random.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITERS 512
__global__ void __launch_bounds__(1024) k_gather(const int *__restrict__ idx, double *out, long long *cyc, int nwaves_active)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = i * 0.5;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ix[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ix[k] = idx[k * 64 + lane];
    double a = 0.0;
    __syncthreads();
    const long long t0 = wall_clock64();
    if (wave < nwaves_active)
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int k = 0; k < 8; k++) a += lds[(ix[k] + 32 * it * (k + 1)) & 16383];      // (+ a multiple of 32: the bank pair of every lane stays what the pattern says)
        }
    __syncthreads();
    const long long t1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main()
{
    int *d_idx; double *d_out; long long *d_cyc;
    CK(hipMalloc(&d_idx, 8 * 64 * 4)); CK(hipMalloc(&d_out, 1024 * 8)); CK(hipMalloc(&d_cyc, 8));
    CK(hipFuncSetAttribute((const void *)k_gather, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    printf("{\"ns_per_wave_gather_instruction\": [\n");
    for (int nw : {1, 16}) {
        for (int mult : {1, 2, 4, 8, 32, 0}) {
            std::vector<int> idx(8 * 64);
            srand(7 + mult);
            for (int k = 0; k < 8; k++)
                for (int l = 0; l < 64; l++) {
                    const int h = l & 31;                     // position inside the 32-lane half
                    if (mult == 0) idx[k * 64 + l] = rand() & 16383;
                    else idx[k * 64 + l] = (h / mult) + 32 * ((h % mult) * 7 + k * 3 + (l >> 5) * 11);      // bank pair = h / mult: `mult` lanes of the half share it (different rows)
                }
            CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_gather, dim3(1), dim3(1024), 131072, 0, d_idx, d_out, d_cyc, nw);
            hipLaunchKernelGGL(k_gather, dim3(1), dim3(1024), 131072, 0, d_idx, d_out, d_cyc, nw);
            CK(hipDeviceSynchronize());
            long long c; CK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
            // wall clock 100 MHz: 10 ns per tick; instructions issued by the CU: nw waves x ITERS x 8
            printf(" {\"waves\": %d, \"lanes_per_bank_pair\": \"%s%d\", \"ns_per_instruction_per_CU\": %.2f},\n", nw, mult == 0 ? "random/" : "", mult, 10.0 * c / ((double)nw * ITERS * 8));
        }
    }
    printf(" null]}\n");
    return 0;
}
