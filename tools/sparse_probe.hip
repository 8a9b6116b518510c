// sparse_probe.hip -- A/B probe for the CSR row pass: gathers of the fp64 vector from L2 (library r1) vs from an
// LDS-staged hot prefix of the vector ("hot-first" local feature order).
// Build: hipcc -O3 --offload-arch=gfx950 tools/sparse_probe.hip -o tools/sparse_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Args { const int *rp; const int *ci; const double *v; const double *wd; double *coef; int l, n, nnz_per, rows_per_blk, hot; long cistride, vstride, lstride; };

template <int G> __device__ __forceinline__ double group_sum(double x) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// A: library r1 structure: 8 lanes per row, 4 entries per lane per round, gathers from global (L2)
__global__ void __launch_bounds__(256) k_A(Args a) {
    constexpr int G = 8, RU = 4, GPB = 256 / G;
    const int p = blockIdx.y, b = blockIdx.x;
    const int *rp = a.rp, *ci = a.ci + (long)p * a.cistride;
    const double *v = a.v + (long)p * a.vstride, *wd = a.wd + (long)p * a.lstride;
    double *coef = a.coef + (long)p * a.lstride;
    const int gid = threadIdx.x / G, gl = threadIdx.x % G;
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    for (int base = r0; base < r1; base += GPB) {
        const int row = base + gid; const bool valid = row < r1; const int rowc = min(row, r1 - 1);
        const int k0 = rp[rowc], k1 = valid ? rp[rowc + 1] : k0;
        const double w = wd[rowc];
        double s = 0;
        for (int kb = k0; kb < k1; kb += G * RU) {
            int idx[RU]; double vv[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) idx[u] = ci[min(kb + gl + u * G, k1 - 1)];
#pragma unroll
            for (int u = 0; u < RU; u++) vv[u] = v[idx[u]];
#pragma unroll
            for (int u = 0; u < RU; u++) s += (kb + gl + u * G) < k1 ? vv[u] : 0.0;
        }
        s = group_sum<G>(s);
        if (valid && gl == 0) coef[row] = w * s;
    }
}

// A4: like A but RB row-groups per lane in flight (rp, then all index loads, then all gathers): 4x the memory-level parallelism
template <int RB, int NT>
__global__ void __launch_bounds__(NT) k_A4(Args a) {
    constexpr int G = 8, RU = 3, GPB = NT / G;
    const int p = blockIdx.y, b = blockIdx.x;
    const int *rp = a.rp, *ci = a.ci + (long)p * a.cistride;
    const double *v = a.v + (long)p * a.vstride, *wd = a.wd + (long)p * a.lstride;
    double *coef = a.coef + (long)p * a.lstride;
    const int gid = threadIdx.x / G, gl = threadIdx.x % G;
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    for (int base = r0; base < r1; base += GPB * RB) {
        int k0[RB], k1[RB]; double w[RB], s[RB];
#pragma unroll
        for (int q = 0; q < RB; q++) { const int row = base + q * GPB + gid; const int rowc = min(row, r1 - 1);
            k0[q] = rp[rowc]; k1[q] = row < r1 ? rp[rowc + 1] : k0[q]; w[q] = wd[rowc]; s[q] = 0; }
        int idx[RB][RU]; double vv[RB][RU];
#pragma unroll
        for (int q = 0; q < RB; q++)
#pragma unroll
            for (int u = 0; u < RU; u++) idx[q][u] = ci[min(k0[q] + gl + u * G, max(k1[q] - 1, k0[q]))];
#pragma unroll
        for (int q = 0; q < RB; q++)
#pragma unroll
            for (int u = 0; u < RU; u++) vv[q][u] = v[idx[q][u]];
#pragma unroll
        for (int q = 0; q < RB; q++) {
#pragma unroll
            for (int u = 0; u < RU; u++) s[q] += (k0[q] + gl + u * G) < k1[q] ? vv[q][u] : 0.0;
            // rows longer than G*RU entries: remaining rounds (not hit by this data)
            for (int kb = k0[q] + G * RU; kb < k1[q]; kb += G) { const int k = kb + gl; if (k < k1[q]) s[q] += v[ci[k]]; }
            s[q] = group_sum<G>(s[q]);
            const int row = base + q * GPB + gid;
            if (row < r1 && gl == 0) coef[row] = w[q] * s[q];
        }
    }
}

// B4: A4's row-group ILP + LDS-staged hot prefix (cold entries from global)
template <int RB, int NT, int ABL = 0>
__global__ void __launch_bounds__(NT) k_B4(Args a) {
    extern __shared__ double vh[];
    constexpr int G = 8, RU = 3, GPB = NT / G;
    const int p = blockIdx.y, b = blockIdx.x;
    const int *rp = a.rp, *ci = a.ci + (long)p * a.cistride;
    const double *v = a.v + (long)p * a.vstride, *wd = a.wd + (long)p * a.lstride;
    double *coef = a.coef + (long)p * a.lstride;
    const int hot = min(a.hot, a.n);
    for (int j = threadIdx.x; j < hot; j += NT) vh[j] = v[j];
    __syncthreads();
    const int gid = threadIdx.x / G, gl = threadIdx.x % G;
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    for (int base = r0; base < r1; base += GPB * RB) {
        int k0[RB], k1[RB]; double w[RB], s[RB];
#pragma unroll
        for (int q = 0; q < RB; q++) { const int row = base + q * GPB + gid; const int rowc = min(row, r1 - 1);
            k0[q] = rp[rowc]; k1[q] = row < r1 ? rp[rowc + 1] : k0[q]; w[q] = wd[rowc]; s[q] = 0; }
        int idx[RB][RU]; double vv[RB][RU];
#pragma unroll
        for (int q = 0; q < RB; q++)
#pragma unroll
            for (int u = 0; u < RU; u++) {
                if (ABL == 2) { unsigned hsh = (unsigned)(k0[q] + gl + u * G) * 2654435761u; idx[q][u] = (hsh >> 8) % (unsigned)((hsh & 7) ? 8192 : a.n); }
                else idx[q][u] = ci[min(k0[q] + gl + u * G, max(k1[q] - 1, k0[q]))];
            }
#pragma unroll
        for (int q = 0; q < RB; q++)
#pragma unroll
            for (int u = 0; u < RU; u++) vv[q][u] = (ABL == 1) ? (double)idx[q][u] : ((idx[q][u] < hot) ? vh[idx[q][u]] : v[idx[q][u]]);
#pragma unroll
        for (int q = 0; q < RB; q++) {
#pragma unroll
            for (int u = 0; u < RU; u++) s[q] += (k0[q] + gl + u * G) < k1[q] ? vv[q][u] : 0.0;
            for (int kb = k0[q] + G * RU; kb < k1[q]; kb += G) { const int k = kb + gl; if (k < k1[q]) s[q] += v[ci[k]]; }
            s[q] = group_sum<G>(s[q]);
            const int row = base + q * GPB + gid;
            if (row < r1 && gl == 0) coef[row] = w[q] * s[q];
        }
    }
}

// B: 1024 threads, hot prefix of v staged in LDS (HOT doubles), cold entries from global
template <int G>
__global__ void __launch_bounds__(1024) k_B(Args a) {
    extern __shared__ double vh[];
    constexpr int RU = 4, GPB = 1024 / G;
    const int p = blockIdx.y, b = blockIdx.x;
    const int *rp = a.rp, *ci = a.ci + (long)p * a.cistride;
    const double *v = a.v + (long)p * a.vstride, *wd = a.wd + (long)p * a.lstride;
    double *coef = a.coef + (long)p * a.lstride;
    const int hot = min(a.hot, a.n);
    for (int j = threadIdx.x; j < hot; j += 1024) vh[j] = v[j];
    __syncthreads();
    const int gid = threadIdx.x / G, gl = threadIdx.x % G;
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    for (int base = r0; base < r1; base += GPB) {
        const int row = base + gid; const bool valid = row < r1; const int rowc = min(row, r1 - 1);
        const int k0 = rp[rowc], k1 = valid ? rp[rowc + 1] : k0;
        const double w = wd[rowc];
        double s = 0;
        for (int kb = k0; kb < k1; kb += G * RU) {
            int idx[RU]; double vv[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) idx[u] = ci[min(kb + gl + u * G, k1 - 1)];
#pragma unroll
            for (int u = 0; u < RU; u++) vv[u] = (idx[u] < hot) ? vh[idx[u]] : v[idx[u]];
#pragma unroll
            for (int u = 0; u < RU; u++) s += (kb + gl + u * G) < k1 ? vv[u] : 0.0;
        }
        s = group_sum<G>(s);
        if (valid && gl == 0) coef[row] = w * s;
    }
}

// C: thread per row (fixed-width friendly): all of a row's indices then all gathers in flight, LDS hot prefix
__global__ void __launch_bounds__(1024) k_C(Args a) {
    extern __shared__ double vh[];
    const int p = blockIdx.y, b = blockIdx.x;
    const int *rp = a.rp, *ci = a.ci + (long)p * a.cistride;
    const double *v = a.v + (long)p * a.vstride, *wd = a.wd + (long)p * a.lstride;
    double *coef = a.coef + (long)p * a.lstride;
    const int hot = min(a.hot, a.n);
    for (int j = threadIdx.x; j < hot; j += 1024) vh[j] = v[j];
    __syncthreads();
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    for (int row = r0 + threadIdx.x; row < r1; row += 1024) {
        const int k0 = rp[row], k1 = rp[row + 1];
        double s = 0;
        for (int kb = k0; kb < k1; kb += 8) {
            int idx[8]; double vv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) idx[u] = ci[min(kb + u, k1 - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) vv[u] = (idx[u] < hot) ? vh[idx[u]] : v[idx[u]];
#pragma unroll
            for (int u = 0; u < 8; u++) s += (kb + u) < k1 ? vv[u] : 0.0;
        }
        coef[row] = wd[row] * s;
    }
}

// D: thread per row, ALL of the row's indices in flight at once (MAXR slots), then all gathers; optional LDS hot prefix
template <int MAXR, bool USE_LDS, int NT>
__global__ void __launch_bounds__(NT) k_D(Args a) {
    extern __shared__ double vh[];
    const int p = blockIdx.y, b = blockIdx.x;
    const int *rp = a.rp, *ci = a.ci + (long)p * a.cistride;
    const double *v = a.v + (long)p * a.vstride, *wd = a.wd + (long)p * a.lstride;
    double *coef = a.coef + (long)p * a.lstride;
    const int hot = USE_LDS ? min(a.hot, a.n) : 0;
    if (USE_LDS) { for (int j = threadIdx.x; j < hot; j += NT) vh[j] = v[j]; __syncthreads(); }
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    for (int rowb = r0; rowb < r1; rowb += NT) {
        const int row = rowb + threadIdx.x; const bool valid = row < r1; const int rowc = min(row, r1 - 1);
        const int k0 = rp[rowc], k1 = valid ? rp[rowc + 1] : k0;
        const double w = wd[rowc];
        int idx[MAXR]; double vv[MAXR];
#pragma unroll
        for (int u = 0; u < MAXR; u++) idx[u] = ci[min(k0 + u, max(k1 - 1, k0))];
#pragma unroll
        for (int u = 0; u < MAXR; u++) vv[u] = (USE_LDS && idx[u] < hot) ? vh[idx[u]] : v[idx[u]];
        double s = 0;
#pragma unroll
        for (int u = 0; u < MAXR; u++) s += (k0 + u) < k1 ? vv[u] : 0.0;
        if (valid) coef[row] = w * s;
    }
}

int main(int argc, char **argv) {
    const int NP = argc > 1 ? atoi(argv[1]) : 256, L = argc > 2 ? atoi(argv[2]) : 39063, N = argc > 3 ? atoi(argv[3]) : 70000, K = 20;
    const int reps = 5;
    std::mt19937_64 rng(1);
    // Zipf(1.1) over 5000 levels x 20 fields, local ids ordered hot-first: id = rank of (field, level) by frequency
    std::vector<double> cdf(5000); { double s = 0; for (int r = 0; r < 5000; r++) { s += pow(r + 1.0, -1.1); cdf[r] = s; } for (auto &c : cdf) c /= s; }
    const long nnz = (long)L * K;
    std::vector<int> rp(L + 1), ci(nnz);
    for (int i = 0; i <= L; i++) rp[i] = i * K;
    std::uniform_real_distribution<double> U(0, 1);
    for (int i = 0; i < L; i++) {
        for (int f = 0; f < K; f++) { int lev = (int)(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin()); if (lev > 4999) lev = 4999;
            int id = lev * K + f;                 // hot-first: level-major => frequent levels get small ids
            ci[(long)i * K + f] = std::min(id, N - 1); }
        std::sort(ci.begin() + (long)i * K, ci.begin() + (long)(i + 1) * K);
    }
    long hot16 = 0; for (auto c : ci) hot16 += c < 16384;
    printf("NP=%d L=%d N=%d nnz/partition=%ld, share of entries with id<16384: %.3f\n", NP, L, N, nnz, (double)hot16 / nnz);
    int *d_rp, *d_ci; double *d_v, *d_wd, *d_coef;
    CK(hipMalloc(&d_rp, sizeof(int) * (L + 1))); CK(hipMalloc(&d_ci, sizeof(int) * nnz * NP)); CK(hipMalloc(&d_v, sizeof(double) * (long)N * NP));
    CK(hipMalloc(&d_wd, sizeof(double) * (long)L * NP)); CK(hipMalloc(&d_coef, sizeof(double) * (long)L * NP));
    CK(hipMemcpy(d_rp, rp.data(), sizeof(int) * (L + 1), hipMemcpyHostToDevice));
    for (int p = 0; p < NP; p++) CK(hipMemcpy(d_ci + (long)p * nnz, ci.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
    { std::vector<double> hv((long)N * NP), hw((long)L * NP); for (auto &q : hv) q = U(rng) - 0.5; for (auto &q : hw) q = U(rng) * 0.25;
      CK(hipMemcpy(d_v, hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_wd, hw.data(), sizeof(double) * hw.size(), hipMemcpyHostToDevice)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)NP * (nnz * 4.0 + 8.0 * L + 8.0 * N);
    struct V { const char *name; int id; int rpb; int hot; };
    std::vector<V> vs = {{"A 8 lanes/row, global gathers, 128 rows/blk (r1)", 0, 128, 0},
                         {"B4 LDS hot 8192 + 4 row-groups, 512 thr, 2048 rows/blk", 7, 2048, 8192},
                         {"B4 ablation: index loads only (no gathers)", 9, 2048, 8192},
                         {"B4 ablation: gathers only (synthetic indices)", 10, 2048, 8192},
                         {"B4 LDS hot 16384 + 4 row-groups, 512 thr, 4096 rows/blk", 7, 4096, 16384},
                         {"B8 LDS hot 8192 + 8 row-groups, 256 thr, 2048 rows/blk", 8, 2048, 8192},
                         {"B4 LDS hot 4096 + 4 row-groups, 512 thr, 2048 rows/blk", 7, 2048, 4096},
                         {"B G=8 LDS hot 16384, 4096 rows/blk", 1, 4096, 16384}, {"B G=8 LDS hot 16384, 8192 rows/blk", 1, 8192, 16384},
                         {"B G=8 LDS hot 18000, 4096 rows/blk", 1, 4096, 18000}, {"B G=8 hot 0 (global only), 4096 rows/blk", 1, 4096, 0},
                         {"C thread/row LDS hot 16384, 4096 rows/blk", 2, 4096, 16384}, {"C thread/row LDS hot 16384, 8192 rows/blk", 2, 8192, 16384},
                         {"D thread/row all-in-flight(24) LDS hot 16384, 4096 rows/blk", 3, 4096, 16384},
                         {"D thread/row all-in-flight(24) LDS hot 16384, 8192 rows/blk", 3, 8192, 16384},
                         {"D0 thread/row all-in-flight(24) global, 256 thr, 1024 rows/blk", 4, 1024, 0},
                         {"D0 thread/row all-in-flight(24) global, 256 thr, 256 rows/blk", 4, 256, 0},
                         {"A4 8 lanes/row x4 row-groups in flight, 256 thr, 512 rows/blk", 5, 512, 0},
                         {"A8 8 lanes/row x8 row-groups in flight, 256 thr, 1024 rows/blk", 6, 1024, 0}};
    std::vector<std::vector<float>> ms(vs.size());
    std::vector<double> ref((long)L), got((long)L);
    CK(hipFuncSetAttribute((const void *)k_B<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    CK(hipFuncSetAttribute((const void *)k_C, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    CK(hipFuncSetAttribute((const void *)k_B4<4, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    CK(hipFuncSetAttribute((const void *)k_B4<8, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    CK(hipFuncSetAttribute((const void *)k_B4<4, 512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    CK(hipFuncSetAttribute((const void *)k_B4<4, 512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    CK(hipFuncSetAttribute((const void *)k_D<24, true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    for (int r = 0; r < reps + 1; r++) for (size_t i = 0; i < vs.size(); i++) {
        Args a{d_rp, d_ci, d_v, d_wd, d_coef, L, N, K, vs[i].rpb, vs[i].hot, nnz, N, L};
        dim3 g((L + vs[i].rpb - 1) / vs[i].rpb, NP);
        CK(hipEventRecord(e0));
        if (vs[i].id == 0) hipLaunchKernelGGL(k_A, g, dim3(256), 0, 0, a);
        else if (vs[i].id == 1) hipLaunchKernelGGL(k_B<8>, g, dim3(1024), sizeof(double) * std::max(1, vs[i].hot), 0, a);
        else if (vs[i].id == 2) hipLaunchKernelGGL(k_C, g, dim3(1024), sizeof(double) * std::max(1, vs[i].hot), 0, a);
        else if (vs[i].id == 3) hipLaunchKernelGGL((k_D<24, true, 1024>), g, dim3(1024), sizeof(double) * std::max(1, vs[i].hot), 0, a);
        else if (vs[i].id == 4) hipLaunchKernelGGL((k_D<24, false, 256>), g, dim3(256), 8, 0, a);
        else if (vs[i].id == 7) hipLaunchKernelGGL((k_B4<4, 512>), g, dim3(512), sizeof(double) * vs[i].hot, 0, a);
        else if (vs[i].id == 8) hipLaunchKernelGGL((k_B4<8, 256>), g, dim3(256), sizeof(double) * vs[i].hot, 0, a);
        else if (vs[i].id == 9) hipLaunchKernelGGL((k_B4<4, 512, 1>), g, dim3(512), sizeof(double) * vs[i].hot, 0, a);
        else if (vs[i].id == 10) hipLaunchKernelGGL((k_B4<4, 512, 2>), g, dim3(512), sizeof(double) * vs[i].hot, 0, a);
        else if (vs[i].id == 5) hipLaunchKernelGGL((k_A4<4, 256>), g, dim3(256), 0, 0, a);
        else hipLaunchKernelGGL((k_A4<8, 256>), g, dim3(256), 0, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float m; CK(hipEventElapsedTime(&m, e0, e1));
        if (r > 0) ms[i].push_back(m);
        if (r == 0) { CK(hipMemcpy(got.data(), d_coef, sizeof(double) * L, hipMemcpyDeviceToHost));
            if (i == 0) ref = got; else { double md = 0; for (int k = 0; k < L; k++) md = std::max(md, fabs(got[k] - ref[k])); printf("  check %-50s max|diff| %.2e\n", vs[i].name, md); } }
    }
    for (size_t i = 0; i < vs.size(); i++) { std::sort(ms[i].begin(), ms[i].end());
        printf("%-52s median %.3f ms -> %.0f GB/s algorithmic, %.1f G gathers/s\n", vs[i].name, ms[i][reps / 2], bytes / (ms[i][reps / 2] * 1e-3) / 1e9, NP * (double)nnz / (ms[i][reps / 2] * 1e-3) / 1e9); }
    return 0;
}
