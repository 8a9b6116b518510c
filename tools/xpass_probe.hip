// xpass_probe.hip -- standalone A/B probe for the dense fused X-pass kernel (CG mode).
// Build: hipcc -O3 --offload-arch=gfx950 tools/xpass_probe.hip -o tools/xpass_probe
// Runs each variant interleaved over NP partitions of L x NF fp32 (default 64 x 15625 x 1000) and prints GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Args {
    const float *X; const double *v; const double *wd; double *parts; int l, nf, ld, rows_per_blk, nblk;
    long xstride, vstride, wdstride, pstride;   // per-partition strides (elements)
};

__device__ __forceinline__ double wave_allreduce_sum(double x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// ---- variant F: pure streaming read
__global__ void __launch_bounds__(256) k_stream(Args a) {
    const int p = blockIdx.y, b = blockIdx.x;
    const float *X = a.X + (long)p * a.xstride;
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    const float4 *base = reinterpret_cast<const float4 *>(X + (long)r0 * a.ld);
    const long n4 = (long)(r1 - r0) * a.ld / 4;
    float s = 0;
    for (long i = threadIdx.x; i < n4; i += 256 * 4) {
        float4 q0 = base[i];
        float4 q1 = (i + 256 < n4) ? base[i + 256] : make_float4(0, 0, 0, 0);
        float4 q2 = (i + 512 < n4) ? base[i + 512] : make_float4(0, 0, 0, 0);
        float4 q3 = (i + 768 < n4) ? base[i + 768] : make_float4(0, 0, 0, 0);
        s += q0.x + q0.y + q0.z + q0.w + q1.x + q1.y + q1.z + q1.w + q2.x + q2.y + q2.z + q2.w + q3.x + q3.y + q3.z + q3.w;
    }
    if (s == 123.456f) a.parts[(long)p * a.pstride + b] = s;
}

// ---- variant A: current library kernel (NV float4 per lane, U rows in flight, shuffle all-reduce per row)
template <int NV, int U, int MODE>   // MODE 0: full ; 1: no cross-lane reduce (ceiling) 
__global__ void __launch_bounds__(256) k_A(Args a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int p = blockIdx.y, b = blockIdx.x;
    if (b >= a.nblk) return;
    const int nf = a.nf, n = nf + 1, ld = a.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *__restrict__ X = a.X + (long)p * a.xstride;
    const double *__restrict__ v = a.v + (long)p * a.vstride;
    const double *__restrict__ wd = a.wd + (long)p * a.wdstride;
    double vr[NV][4];
#pragma unroll
    for (int c = 0; c < NV; c++) { const int col0 = (c * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) vr[c][e] = (col0 + e < nf) ? v[col0 + e] : 0.0; }
    const double vb = v[nf];
    double acc[NV][4];
#pragma unroll
    for (int c = 0; c < NV; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[c][e] = 0.0;
    double accb = 0.0;
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    for (int rb = r0 + wave * U; rb < r1; rb += 4 * U) {
        float4 x[U][NV];
#pragma unroll
        for (int u = 0; u < U; u++) { const int row = min(rb + u, r1 - 1); const float *xr = X + (long)row * ld;
#pragma unroll
            for (int c = 0; c < NV; c++) { const int col0 = (c * 64 + lane) * 4;
                x[u][c] = *reinterpret_cast<const float4 *>(xr + min(col0, ld - 4)); } }
        double t[U];
#pragma unroll
        for (int u = 0; u < U; u++) { double s = 0.0;
#pragma unroll
            for (int c = 0; c < NV; c++) { s += (double)x[u][c].x * vr[c][0]; s += (double)x[u][c].y * vr[c][1]; s += (double)x[u][c].z * vr[c][2]; s += (double)x[u][c].w * vr[c][3]; }
            t[u] = s; }
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int u = 0; u < U; u++) t[u] = wave_allreduce_sum(t[u]) + vb;
        }
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int c = 0; c < NV; c++) asm volatile("" : "+v"(x[u][c].x), "+v"(x[u][c].y), "+v"(x[u][c].z), "+v"(x[u][c].w));
        }
        double tm = t[0];
#pragma unroll
        for (int u = 1; u < U; u++) tm = (lane == u) ? t[u] : tm;
        const int myrow = rb + lane;
        double coef_m = 0.0;
        if (lane < U && myrow < r1) coef_m = wd[myrow] * tm;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const double cf = (MODE != 1) ? __shfl(coef_m, u, 64) : t[u];
            accb += cf;
#pragma unroll
            for (int c = 0; c < NV; c++) { acc[c][0] += (double)x[u][c].x * cf; acc[c][1] += (double)x[u][c].y * cf; acc[c][2] += (double)x[u][c].z * cf; acc[c][3] += (double)x[u][c].w * cf; }
        }
    }
    const int NC = NV * 256;
    double *red = smem, *redb = smem + 4 * NC;
#pragma unroll
    for (int c = 0; c < NV; c++) { const int col0 = (c * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) red[wave * NC + col0 + e] = acc[c][e]; }
    if (lane == 0) redb[wave] = accb;
    __syncthreads();
    double *outp = a.parts + (long)p * a.pstride + (long)b * n;
    for (int j = threadIdx.x; j < nf; j += 256) outp[j] = ((red[j] + red[NC + j]) + red[2 * NC + j]) + red[3 * NC + j];
    if (threadIdx.x == 0) outp[nf] = ((redb[0] + redb[1]) + redb[2]) + redb[3];
}

// ---- variant C: reduce-scatter butterfly over U=4 rows (7 exchanges instead of 24), readlane broadcast,
// X kept as fp32 in registers (converted at use), launch bounds for 2 waves/SIMD.
__device__ __forceinline__ double readlane_d(double x, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
template <int NV, int MINW>
__global__ void __launch_bounds__(256, MINW) k_C(Args a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int U = 4;
    const int p = blockIdx.y, b = blockIdx.x;
    if (b >= a.nblk) return;
    const int nf = a.nf, n = nf + 1, ld = a.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *__restrict__ X = a.X + (long)p * a.xstride;
    const double *__restrict__ v = a.v + (long)p * a.vstride;
    const double *__restrict__ wd = a.wd + (long)p * a.wdstride;
    double vr[NV][4];
#pragma unroll
    for (int c = 0; c < NV; c++) { const int col0 = (c * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) vr[c][e] = (col0 + e < nf) ? v[col0 + e] : 0.0; }
    const double vb = v[nf];
    double acc[NV][4];
#pragma unroll
    for (int c = 0; c < NV; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[c][e] = 0.0;
    double accb = 0.0;
    const int r0 = b * a.rows_per_blk, r1 = min(a.l, r0 + a.rows_per_blk);
    const bool hi32 = lane & 32, hi16 = lane & 16;
    const int myu = (hi32 ? 2 : 0) + (hi16 ? 1 : 0);      // the row of the batch whose total this lane ends up with
    for (int rb = r0 + wave * U; rb < r1; rb += 4 * U) {
        float4 x[U][NV];
#pragma unroll
        for (int u = 0; u < U; u++) { const int row = min(rb + u, r1 - 1); const float *xr = X + (long)row * ld;
#pragma unroll
            for (int c = 0; c < NV; c++) { const int col0 = (c * 64 + lane) * 4;
                x[u][c] = *reinterpret_cast<const float4 *>(xr + min(col0, ld - 4)); } }
        double t[U];
#pragma unroll
        for (int u = 0; u < U; u++) { double s = 0.0;
#pragma unroll
            for (int c = 0; c < NV; c++) { s += (double)x[u][c].x * vr[c][0]; s += (double)x[u][c].y * vr[c][1]; s += (double)x[u][c].z * vr[c][2]; s += (double)x[u][c].w * vr[c][3]; }
            t[u] = s; }
        // reduce-scatter: lanes with bit5 set keep rows 2,3 ; bit4 set keeps the odd row of the pair
        double k0 = hi32 ? t[2] : t[0], k1 = hi32 ? t[3] : t[1];
        double s0 = hi32 ? t[0] : t[2], s1 = hi32 ? t[1] : t[3];
        k0 += __shfl_xor(s0, 32, 64);
        k1 += __shfl_xor(s1, 32, 64);
        double kk = hi16 ? k1 : k0, ss = hi16 ? k0 : k1;
        kk += __shfl_xor(ss, 16, 64);
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) kk += __shfl_xor(kk, m, 64);
        const int myrow = rb + myu;
        double coef_m = 0.0;
        if (myrow < r1) coef_m = wd[myrow] * (kk + vb);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const double cf = readlane_d(coef_m, ((u & 2) ? 32 : 0) + ((u & 1) ? 16 : 0));
            accb += cf;
#pragma unroll
            for (int c = 0; c < NV; c++) { acc[c][0] += (double)x[u][c].x * cf; acc[c][1] += (double)x[u][c].y * cf; acc[c][2] += (double)x[u][c].z * cf; acc[c][3] += (double)x[u][c].w * cf; }
        }
    }
    const int NC = NV * 256;
    double *red = smem, *redb = smem + 4 * NC;
#pragma unroll
    for (int c = 0; c < NV; c++) { const int col0 = (c * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) red[wave * NC + col0 + e] = acc[c][e]; }
    if (lane == 0) redb[wave] = accb;
    __syncthreads();
    double *outp = a.parts + (long)p * a.pstride + (long)b * n;
    for (int j = threadIdx.x; j < nf; j += 256) outp[j] = ((red[j] + red[NC + j]) + red[2 * NC + j]) + red[3 * NC + j];
    if (threadIdx.x == 0) outp[nf] = ((redb[0] + redb[1]) + redb[2]) + redb[3];
}

int main(int argc, char **argv) {
    int NP = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 15625, NF = argc > 3 ? atoi(argv[3]) : 1000;
    int RPB = argc > 4 ? atoi(argv[4]) : 256, reps = argc > 5 ? atoi(argv[5]) : 5;
    const int ld = (NF + 3) / 4 * 4, n = NF + 1;
    const int nblk = (L + RPB - 1) / RPB;
    float *X; double *v, *wd, *parts;
    const long xs = (long)L * ld;
    CK(hipMalloc(&X, sizeof(float) * xs * NP)); CK(hipMalloc(&v, sizeof(double) * n * NP)); CK(hipMalloc(&wd, sizeof(double) * L * NP));
    CK(hipMalloc(&parts, sizeof(double) * (long)nblk * n * NP));
    { std::vector<float> h(xs); srand(1); const bool gauss = getenv("PROBE_GAUSS") != nullptr;
      for (long i = 0; i < xs; i++) { if (gauss) { double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0); h[i] = (float)(sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2)); } else h[i] = (float)((rand() % 2001) - 1000) / 500.f; }
      for (int p = 0; p < NP; p++) CK(hipMemcpy(X + (long)p * xs, h.data(), sizeof(float) * xs, hipMemcpyHostToDevice));
      std::vector<double> hv(n * NP), hw((long)L * NP); for (auto &q : hv) q = (rand() % 1000) / 1000.0 - 0.5; for (auto &q : hw) q = (rand() % 1000) / 4000.0;
      CK(hipMemcpy(v, hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(wd, hw.data(), sizeof(double) * hw.size(), hipMemcpyHostToDevice)); }
    Args a{X, v, wd, parts, L, NF, ld, RPB, nblk, xs, n, L, (long)nblk * n};
    const double bytes = (double)NP * (4.0 * L * NF + 8.0 * L + 8.0 * n);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sh4 = (4 * 4 * 256 + 16) * sizeof(double);
    struct V { const char *name; int id; };
    std::vector<V> vs = {{"F stream-read", 0}, {"A NV4 U4 (library r1)", 1}, {"A NV4 U2", 2}, {"A NV4 U4 no-reduce ceiling", 3},
                         {"C butterfly NV4 U4 minw1", 4}, {"C butterfly NV4 U4 minw2", 5},
                         {"D NV4 U4 reconvert (fp32 regs)", 6}, {"D NV4 U8 reconvert", 7}};
    std::vector<std::vector<float>> ms(vs.size());
    std::vector<double> ref((long)nblk * n), got((long)nblk * n);
    for (int r = 0; r < reps + 1; r++) for (size_t i = 0; i < vs.size(); i++) {
        dim3 g(nblk, NP), blk(256);
        CK(hipEventRecord(e0));
        switch (vs[i].id) {
        case 0: hipLaunchKernelGGL(k_stream, g, blk, 0, 0, a); break;
        case 1: hipLaunchKernelGGL((k_A<4, 4, 0>), g, blk, sh4, 0, a); break;
        case 2: hipLaunchKernelGGL((k_A<4, 2, 0>), g, blk, sh4, 0, a); break;
        case 3: hipLaunchKernelGGL((k_A<4, 4, 1>), g, blk, sh4, 0, a); break;
        case 4: hipLaunchKernelGGL((k_C<4, 1>), g, blk, sh4, 0, a); break;
        case 5: hipLaunchKernelGGL((k_C<4, 2>), g, blk, sh4, 0, a); break;
        case 6: hipLaunchKernelGGL((k_A<4, 4, 2>), g, blk, sh4, 0, a); break;
        case 7: hipLaunchKernelGGL((k_A<4, 8, 2>), g, blk, sh4, 0, a); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float m; CK(hipEventElapsedTime(&m, e0, e1));
        if (r > 0) ms[i].push_back(m);
        if (r == 0 && (vs[i].id == 1 || vs[i].id >= 4)) {
            CK(hipMemcpy(got.data(), parts, sizeof(double) * got.size(), hipMemcpyDeviceToHost));
            if (vs[i].id == 1) ref = got;
            else { double md = 0, mx = 0; for (size_t k = 0; k < got.size(); k++) { md = std::max(md, fabs(got[k] - ref[k])); mx = std::max(mx, fabs(ref[k])); }
                   printf("  check %-28s max|diff| = %.3e (max|ref| %.3e)\n", vs[i].name, md, mx); }
        }
    }
    printf("NP=%d L=%d NF=%d rows/blk=%d blocks=%d x %d, algorithmic %.3f GB per launch\n", NP, L, NF, RPB, nblk, NP, bytes / 1e9);
    for (size_t i = 0; i < vs.size(); i++) {
        std::sort(ms[i].begin(), ms[i].end());
        printf("%-30s median %.3f ms  min %.3f ms  -> %.0f GB/s (median)\n", vs[i].name, ms[i][ms[i].size() / 2], ms[i][0], bytes / (ms[i][ms[i].size() / 2] * 1e-3) / 1e9);
    }
    return 0;
}
