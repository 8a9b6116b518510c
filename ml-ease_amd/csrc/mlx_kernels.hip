// mlx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the batched TRON solve and the
// ADMM consensus. No library calls, no CUDA shims: HIP C++ for MI355X only.
//
// Execution model (DESIGN.md "Tick machine"): every (partition, lambda) problem advances in
// lock-step TICKS. One tick = one pass over X for every unfinished problem (dense tiles: k_xpass_dense; sliced CSR
// partitions: [k_rowcold +] k_rowpass_lds, then k_colpass_lds) followed by the TRON/CG step (dense: k_tron_step, one
// workgroup per problem; CSR: k_step_head + k_step_a / b / c + k_step_commit over column chunks), which owns all of Tron's control flow
// (bw/Tron.java:30-179) for its problem and decides what the next pass computes:
//     PH_CG    pass computes  X' diag(wt*D) X d          (the Hv of llf/LogisticRegressionL2.java:231-248)
//     PH_EVAL  pass computes  loss(w_new), D(w_new), X' t(w_new)  (fun + grad fused, :156-225)
// Fusions relative to the reference (all exact in real arithmetic, fp64 throughout):
//   * Hv reads X once (row tile in registers: dot, scale, rank-1 accumulate) instead of Xv + XTv;
//   * fun and grad share one pass; after a rejected step the trial D is discarded (wd double buffer);
//   * grad(0)'s data term X' t0 is a per-partition constant (c0), computed once at upload.
// Arithmetic that is NOT a plain tree on purpose: the two n-long dots of a CG step on the CSR path (d.Hd, r.r) round their terms to the
// running sum's grid first, as the reference's sequential loop does (grid_of_sum below).
// Partial sums are functions of the partition alone (dense: per 256-row unit, added in pairs; CSR: per 64-row group), so results do
// not depend on the chunking mlx_finalize picks for a handle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <stdint.h>

#include "mlx_kernels.h"
#include "mlx_types.h"
#include "portable_math.h"
#include "mlx_wave.h"
#include "mlx_seqfold.h"

// Global-memory accesses said to be global. The kernels take their pointers out of PartDev / ProbDev records in memory, so to the
// compiler they are generic and every access became a FLAT instruction: 64-bit per-lane addresses, and -- what costs -- a FLAT load
// counts on vmcnt AND lgkmcnt, so in the kernels that gather from LDS a wait for LDS data also waited for every index pack in flight.
// gld / gst cast at the access (the variables stay plain pointers); -DMLX_NO_GLOBAL_AS: A/B build with the flat accesses.
#ifdef MLX_NO_GLOBAL_AS
#define MLX_GAS
#else
#define MLX_GAS __attribute__((address_space(1)))
#endif
template <typename T> __device__ __forceinline__ T gld(const T *p) { return *(const MLX_GAS T *)p; }
template <typename T> __device__ __forceinline__ T gld_nt(const T *p) { return __builtin_nontemporal_load((const MLX_GAS T *)p); }
template <typename T> __device__ __forceinline__ void gst(T *p, T v) { *(MLX_GAS T *)p = v; }
template <typename T> __device__ __forceinline__ void gst_nt(T *p, T v) { __builtin_nontemporal_store(v, (MLX_GAS T *)p); }

#define WAVE 64

// Timing experiments only (tools/ablate_build.sh -DMLX_PHASE_TIMING): thread 0 of every workgroup of the sparse passes adds
// the 100 MHz wall-clock ticks it spent in each phase to g_phase[]; read back with mlx_debug_phase_times().
#ifdef MLX_PHASE_TIMING
__device__ unsigned long long g_phase[16];
#ifdef MLX_SGF_STATS      /* (the fold's counters own all of g_phase[]: [3 pass + k], mlx_seqfold.h) */
#define PT_INIT
#define PT_MARK(k)
#else
#define PT_INIT unsigned long long pt_last = wall_clock64()
#define PT_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_phase[k], t_ - pt_last); pt_last = t_; } } while (0)
#endif
extern "C" int mlx_debug_phase_times(double *out16)
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h) != hipSuccess) return -1;
    for (int i = 0; i < 16; i++) out16[i] = (double)h[i] * 0.01;      // microseconds
    return 0;
}
#else
#define PT_INIT
#define PT_MARK(k)
#endif

#ifdef MLX_SMALL_PROFILE
// development build only (tools/small_profile.sh): shader-clock stamps of workgroup 0 around the phases of a k_solve_small tick
__device__ unsigned long long g_small_prof[8], g_small_prof2[16];
extern "C" void mlxk_small_prof_read(unsigned long long *out, int reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_prof), sizeof(unsigned long long) * 8);
    hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_small_prof2), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_small_prof), z, sizeof(unsigned long long) * 8); hipMemcpyToSymbol(HIP_SYMBOL(g_small_prof2), z, sizeof(z)); }
}
__device__ unsigned long long g_small_prof_t;
#define SPROF2(slot) do { if (blockIdx.x == 0 && T::tid() == 0 && threadIdx.x == 0) { const unsigned long long t_ = clock64(); g_small_prof2[slot] += t_ - g_small_prof_t; g_small_prof_t = t_; } } while (0)
#define SPROF(slot) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long t_ = clock64(); g_small_prof[slot] += t_ - tprev; tprev = t_; } } while (0)
#else
#define SPROF(slot) do { } while (0)
#define SPROF2(slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// wave / block reductions (deterministic trees; fp64)
// ------------------------------------------------------------------------------------------------
// (the butterflies themselves: mlx_wave.h -- v_permlane*_swap / DPP moves with the association of the __shfl_xor loops they replaced)
__device__ __forceinline__ double wave_allreduce_sum(double x) { return mlx_wave_allreduce_sum(x); }      // partners i^32, i^16, ..., i^1
__device__ __forceinline__ double wave_allreduce_max(double x) { return mlx_wave_allreduce_max(x); }
template <int G>
__device__ __forceinline__ double group_allreduce_sum(double x)      // G aligned lanes: partners i ^ G/2, ..., i ^ 1
{
    double a, b;
    if (G >= 64) { mlx_swap32(x, a, b); x = a + b; }
    if (G >= 32) { mlx_swap16(x, a, b); x = a + b; }
    if (G >= 16) x += mlx_xor8(x);
    if (G >= 8) x += mlx_xor4(x);
    if (G >= 4) x += mlx_xor2(x);
    if (G >= 2) x += mlx_xor1(x);
    return x;
}

// Sum of up to 3 values over the block, result broadcast to every thread. scratch: >= 3*16 doubles.
template <int NVAL>
__device__ __forceinline__ void block_allreduce_sum(double (&v)[NVAL], double *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NVAL; i++) v[i] = wave_allreduce_sum(v[i]);
    __syncthreads();   // scratch reuse
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NVAL; i++) scratch[i * 16 + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NVAL; i++) {
        double a = 0;
        for (int w = 0; w < nw; w++) a += scratch[i * 16 + w];
        v[i] = a;
    }
}
__device__ __forceinline__ double block_allreduce_max(double x, double *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    x = wave_allreduce_max(x);
    __syncthreads();
    if (lane == 0) scratch[wave] = x;
    __syncthreads();
    double a = scratch[0];
    for (int w = 1; w < nw; w++) a = fmax(a, scratch[w]);
    return a;
}

// bw/Tron.java:220-252 euclideanNorm: scale * sqrt(sum((v/scale)^2)) with scale = max|v|
// (the Java keeps a running scale; same value up to the last bits).
__device__ __forceinline__ double block_norm(const double *__restrict__ v, int n, double *scratch)
{
    double mx = 0;
    _Pragma("unroll 4") for (int j = threadIdx.x; j < n; j += blockDim.x) mx = fmax(mx, fabs(v[j]));
    mx = block_allreduce_max(mx, scratch);
    if (!(mx > 0)) return (mx == 0) ? 0.0 : mx;   // 0, or NaN propagates
    double a[1] = {0};
    _Pragma("unroll 4") for (int j = threadIdx.x; j < n; j += blockDim.x) { double t = fabs(v[j]) / mx; a[0] += t * t; }
    block_allreduce_sum<1>(a, scratch);
    return mx * sqrt(a[0]);
}

// Norm from a sum of squares gathered inside an update loop (saves two passes over the vector). For every vector whose
// squares neither overflow nor vanish, sqrt(sum v^2) equals euclideanNorm's scale*sqrt(sum (v/scale)^2) up to the last bits
// (a power-of-two scale makes them identical); outside that range fall back to the scaled two-pass form.
__device__ __forceinline__ double norm_from_sumsq(double ss, const double *__restrict__ v, int n, double *scratch)
{
    if (ss > 1e-280 && ss < 1e280) return sqrt(ss);
    __syncthreads();
    return block_norm(v, n, scratch);
}

// Who runs the TRON/CG step body: the whole workgroup (barriers, block reductions through LDS) or -- for the tiny problems of
// k_solve_small<.., XL> -- its FIRST WAVE alone: at n ~ 200 a CG step is five reductions and a dozen barriers of 16 waves for a
// few hundred flops, and one wave does the same elementwise work (column j on lane j % 64 in every loop, so values only cross
// lanes inside the reductions) with shuffles and no barrier at all.
struct BlockTeam {
    static __device__ __forceinline__ int tid() { return threadIdx.x; }
    static __device__ __forceinline__ int nt() { return blockDim.x; }
    static __device__ __forceinline__ void sync() { __syncthreads(); }
    template <int N> static __device__ __forceinline__ void allreduce(double (&v)[N], double *scratch) { block_allreduce_sum<N>(v, scratch); }
    static __device__ __forceinline__ double allreduce_max(double x, double *scratch) { return block_allreduce_max(x, scratch); }
};
struct WaveTeam {
    static __device__ __forceinline__ int tid() { return threadIdx.x & 63; }
    static __device__ __forceinline__ int nt() { return 64; }
    static __device__ __forceinline__ void sync() { __builtin_amdgcn_wave_barrier(); }      // (one wave: LDS operations complete in issue order)
    template <int N> static __device__ __forceinline__ void allreduce(double (&v)[N], double *)
    {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = wave_allreduce_sum(v[i]);
    }
    static __device__ __forceinline__ double allreduce_max(double x, double *) { return wave_allreduce_max(x); }
};
// Pointer type of the small solver's work vectors: generic, or -- when k_solve_small keeps them in LDS -- an LDS pointer, so that
// every access is a ds_read / ds_write and not a flat instruction that finds the LDS aperture at run time (the descriptor holds
// generic pointers; before this the one-workgroup solver issued ~800 flat loads and ~330 flat stores per tick-loop body).
typedef __attribute__((address_space(3))) double *lds_dptr;
template <typename T, typename P>
__device__ __forceinline__ double team_sum_array(P a, int cnt, double *scratch)
{
    double v[1] = {0.0};
    for (int i = T::tid(); i < cnt; i += T::nt()) v[0] += a[i];
    T::template allreduce<1>(v, scratch);
    return v[0];
}
// the same over PAIRS of a dense partition's per-unit partials (k_xpass_dense): a[] holds pair sums already (stored_pairs) or unit sums
template <typename T, typename P>
__device__ __forceinline__ double team_sum_pairs(P a, int nunits, bool stored_pairs, double *scratch)
{
    const int np = (nunits + 1) / 2;
    double v[1] = {0.0};
    for (int i = T::tid(); i < np; i += T::nt())
        v[0] += stored_pairs ? a[i] : ((2 * i + 1 < nunits) ? a[2 * i] + a[2 * i + 1] : a[2 * i]);
    T::template allreduce<1>(v, scratch);
    return v[0];
}
template <typename T, typename P>
__device__ __forceinline__ double team_norm(P v, int n, double *scratch)      // block_norm for a team
{
    double mx = 0;
    for (int j = T::tid(); j < n; j += T::nt()) mx = fmax(mx, fabs(v[j]));
    mx = T::allreduce_max(mx, scratch);
    if (!(mx > 0)) return (mx == 0) ? 0.0 : mx;
    double a[1] = {0};
    for (int j = T::tid(); j < n; j += T::nt()) { double t = fabs(v[j]) / mx; a[0] += t * t; }
    T::template allreduce<1>(a, scratch);
    return mx * sqrt(a[0]);
}
template <typename T, typename P>
__device__ __forceinline__ double team_norm_from_sumsq(double ss, P v, int n, double *scratch)
{
    if (ss > 1e-280 && ss < 1e280) return sqrt(ss);
    T::sync();
    return team_norm<T>(v, n, scratch);
}

// ------------------------------------------------------------------------------------------------
// row-wise scalar maps
// ------------------------------------------------------------------------------------------------
// fun + grad at one row (llf/LogisticRegressionL2.java:172-178, :211-215): z = x.w + offset
// PM: the portable exp/log1p of portable_math.h (verification mode only; the oracle's -DORC_PORTABLE_MATH twin uses the same)
template <bool PM = false>
__device__ __forceinline__ void row_eval(double z, int y, double wt, double &loss, double &wd, double &coef)
{
#pragma clang fp contract(off)
    const double yz = (double)y * z;
    if (yz >= 0) loss = wt * (PM ? pm_log1p(pm_exp(-yz)) : log1p(exp(-yz)));
    else loss = wt * (-yz + (PM ? pm_log1p(pm_exp(yz)) : log1p(exp(yz))));
    const double p = 1.0 / (1.0 + (PM ? pm_exp(-yz) : exp(-yz)));
    wd = wt * (p * (1.0 - p));          // weight[i] * D[i]  (:243 multiplies in this order)
    coef = wt * (p - 1.0) * (double)y;  // :215
}

// ------------------------------------------------------------------------------------------------
// dense X pass: one read of the fp32 tile per pass, fp64 accumulate
// ------------------------------------------------------------------------------------------------
// Block = 256 threads = 4 waves; a wave owns whole rows: lane l holds float4 #(c*64+l) of the row
// (c < NV), i.e. 16*NV bytes per lane per row, loaded as 1 KiB-per-instruction coalesced reads.
// Per row: partial dot -> wave all-reduce -> row coefficient -> rank-1 accumulate into the lane's
// 4*NV fp64 column accumulators. U rows are in flight per wave for ILP.
// The rows of a partition are cut into UNITS of rows_per_blk rows (256 for partitions of >= 4096 rows: a property of the partition
// alone); a workgroup owns units_per_wg = 2 consecutive units (512 rows: finer chunks cost one prologue each, coarser ones leave CUs
// idle in the tail, profiles/r1_notes.md) when the handle holds many problems, 1 when it holds few. X'c, the loss and the intercept sum
// are reduced PER UNIT (same wave / row assignment whatever the workgroup owns) and units are added in PAIRS (u0 + u1), (u2 + u3), ...:
// a two-unit workgroup stores its pair sum, one-unit workgroups store unit sums and the step forms the pairs (dense_pair_term). The
// sums therefore associate the same way in both layouts: a partition gives bit-identical results whether it shares its GPU with 7
// others or with 63 (1/2/4/8-GPU runs of one job).
template <int NV, int U, bool NT>
__global__ void __launch_bounds__(256)
k_xpass_dense(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = qlist[blockIdx.y];
    ProbDev &pr = probs[q];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    const int b = blockIdx.x;
    const int rpb = pa.rows_per_blk, upw = pa.units_per_wg;
    if (b * upw >= pa.n_units) return;
    const int nf = pa.n_feat, n = nf + 1;
    const int64_t ld = pa.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool cg = (phase == PH_CG);
    const double *__restrict__ v = cg ? pr.d : pr.w_new;
    const double *__restrict__ wdcur = pr.wd[pr.dsel];
    double *__restrict__ wdnew = pr.wd[pr.dsel ^ 1];
    const float *__restrict__ X = pa.X;

    double vr[NV][4];
#pragma unroll
    for (int c = 0; c < NV; c++) {
        const int col0 = (c * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) vr[c][e] = (col0 + e < nf) ? gld(v + col0 + e) : 0.0;
    }
    const double vb = gld(v + nf);
    const int NC = NV * 256;                 // padded columns per wave slice
    double *red = smem;                      // [4][NC]
    double *redb = smem + 4 * NC;            // [4] intercept, [4] loss
    double psum[NV], pbsum = 0.0, plsum = 0.0;   // sums over the workgroup's units of column j = tid + 256 q, the intercept, the loss
    const int nunits = pa.n_units;
    for (int ub = b * upw; ub < min((b + 1) * upw, nunits); ub++) {
    double acc[NV][4];
#pragma unroll
    for (int c = 0; c < NV; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[c][e] = 0.0;
    double accb = 0.0, lossacc = 0.0;

    const int r0 = ub * rpb;
    const int r1 = min(pa.l, r0 + rpb);
    for (int rb = r0 + wave * U; rb < r1; rb += 4 * U) {
        float4 x[U][NV];
#pragma unroll
        for (int u = 0; u < U; u++) {
            // Unconditional loads (clamped addresses): a predicated load makes hipcc wait vmcnt(0) after
            // every single load (16 serialized round trips per batch, measured 3.2 -> 5.8 TB/s). Rows past
            // r1 re-read row r1-1 and get coefficient 0; lanes past the row width re-read its last
            // float4 against vr == 0, and their accumulators are never stored.
            const int row = min(rb + u, r1 - 1);
            const float *__restrict__ xr = X + (int64_t)row * ld;
#pragma unroll
            for (int c = 0; c < NV; c++) {
                const int col0 = min((c * 64 + lane) * 4, (int)ld - 4);
                typedef float f4v __attribute__((ext_vector_type(4)));
                // NT (single lambda): the tile is read once per tick and must not displace the vectors in L2/MALL; with several
                // lambdas per partition the problems run side by side and share the tile through the caches
                const f4v t4 = NT ? gld_nt(reinterpret_cast<const f4v *>(xr + col0)) : gld(reinterpret_cast<const f4v *>(xr + col0));
                x[u][c] = make_float4(t4.x, t4.y, t4.z, t4.w);
            }
        }
        double t[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            double a = 0.0;
#pragma unroll
            for (int c = 0; c < NV; c++) {
                a += (double)x[u][c].x * vr[c][0];
                a += (double)x[u][c].y * vr[c][1];
                a += (double)x[u][c].z * vr[c][2];
                a += (double)x[u][c].w * vr[c][3];
            }
            t[u] = a;
        }
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = wave_allreduce_sum(t[u]) + vb;
        // Keep the tile as fp32 in registers and convert again for the accumulate phase: without this
        // barrier hipcc keeps the 16*U*NV converted doubles alive (256 VGPR + AGPR spill = 1 wave/SIMD).
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int c = 0; c < NV; c++)
                asm volatile("" : "+v"(x[u][c].x), "+v"(x[u][c].y), "+v"(x[u][c].z), "+v"(x[u][c].w));
        // lane u (< U) owns row rb+u for the scalar map, then broadcasts its coefficient
        double tm = t[0];
#pragma unroll
        for (int u = 1; u < U; u++) tm = (lane == u) ? t[u] : tm;
        const int myrow = rb + lane;
        double coef_m = 0.0;
        if (lane < U && myrow < r1) {
            if (cg) {
                coef_m = gld(wdcur + myrow) * tm;                    // wa[i] = weight*D * (X s)[i], :243
            } else {
                double loss, wdv;
                row_eval(tm + (double)gld(pa.off + myrow), (int)gld(pa.y + myrow), (double)gld(pa.wt + myrow), loss, wdv, coef_m);
                gst(wdnew + myrow, wdv);
                lossacc += loss;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const double cf = mlx_wave_bcast(coef_m, u);             // (lane u's value in scalar registers: v_readlane, no LDS crossbar)
            accb += cf;
#pragma unroll
            for (int c = 0; c < NV; c++) {
                acc[c][0] += (double)x[u][c].x * cf;
                acc[c][1] += (double)x[u][c].y * cf;
                acc[c][2] += (double)x[u][c].z * cf;
                acc[c][3] += (double)x[u][c].w * cf;
            }
        }
    }

    // cross-wave reduction in LDS, fixed wave order -> deterministic partials
#pragma unroll
    for (int c = 0; c < NV; c++) {
        const int col0 = (c * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) red[wave * NC + col0 + e] = acc[c][e];
    }
    lossacc = wave_allreduce_sum(lossacc);
    if (lane == 0) { redb[wave] = accb; redb[4 + wave] = lossacc; }
    __syncthreads();
    const bool first = (ub == b * upw);
#pragma unroll
    for (int q = 0; q < NV; q++) {
        const int j = threadIdx.x + 256 * q;
        const double val = ((red[j] + red[NC + j]) + red[2 * NC + j]) + red[3 * NC + j];
        psum[q] = first ? val : psum[q] + val;
    }
    if (threadIdx.x == 0) {
        const double bv = ((redb[0] + redb[1]) + redb[2]) + redb[3], lv = ((redb[4] + redb[5]) + redb[6]) + redb[7];
        pbsum = first ? bv : pbsum + bv;
        plsum = first ? lv : plsum + lv;
    }
    __syncthreads();                         // the next unit overwrites red[]
    }
    double *__restrict__ outp = pr.parts + (int64_t)b * n;
#pragma unroll
    for (int q = 0; q < NV; q++) {
        const int j = threadIdx.x + 256 * q;
        if (j < nf) gst(outp + j, psum[q]);
    }
    if (threadIdx.x == 0) { outp[nf] = pbsum; pr.lossp[b] = plsum; }
}

// XCD-aware work mapping of the sparse passes. Workgroup L of a 1-D grid is dispatched to XCD L % 8 (observed order;
// used for speed only, never for correctness). All chunks of one problem are given to ONE XCD (problem index % 8), so
// the randomly gathered fp64 vectors of the 1-3 problems an XCD works on at a time (d / w_new: 8*n_local bytes, coef:
// 8*l bytes) stay in that XCD's 4 MiB L2 instead of being re-fetched through the fabric by all eight L2s.
__device__ __forceinline__ bool xcd_map(int nq, int gx, int &pi, int &bx, int lead = 0 /* workgroups in front of the mapped ones (a multiple of 8) */)
{
    const int L = (int)blockIdx.x - lead, xcd = L & 7, s = L >> 3;
    pi = (s / gx) * 8 + xcd;
    bx = s % gx;
    return pi < nq;
}

// ------------------------------------------------------------------------------------------------
// sparse X pass, part 1: CSR rows -> row coefficients (gather of v)
// ------------------------------------------------------------------------------------------------
// G lanes per row, RU entries per lane per round: a round is ONE index load + ONE dependent gather per
// lane with G*RU entries of the row in flight (20-nnz rows: one round at G=8). All loads are unconditional
// with clamped addresses (a predicated load would be waited for individually, see k_xpass_dense).
#define RU 4
template <int G, bool HASVAL>
__global__ void __launch_bounds__(256)
k_rowpass_csr(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq, int gx)
{
    __shared__ double scratch[48];
    int pi_, bx_;
    if (!xcd_map(nq, gx, pi_, bx_)) return;
    const int q = qlist[pi_];
    ProbDev &pr = probs[q];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    const int b = bx_;
    if (b >= pa.nblk) return;
    const bool cg = (phase == PH_CG);
    const double *__restrict__ v = cg ? pr.d : pr.w_new;
    const double *__restrict__ wdcur = pr.wd[pr.dsel];
    double *__restrict__ wdnew = pr.wd[pr.dsel ^ 1];
    double *__restrict__ coef = pr.coef;
    const int32_t *__restrict__ rp = pa.rp;
    const int32_t *__restrict__ ci = pa.ci;
    const float *__restrict__ val = pa.val;
    const double vb = v[pa.n_feat];
    constexpr int GPB = 256 / G;
    const int gid = threadIdx.x / G, gl = threadIdx.x % G;
    const int r0 = b * pa.rows_per_blk;
    const int r1 = min(pa.l, r0 + pa.rows_per_blk);
    double red[2] = {0.0, 0.0};          // loss, sum of coef
    for (int base = r0; base < r1; base += GPB) {
        const int row = base + gid;
        const bool valid = row < r1;
        const int rowc = min(row, r1 - 1);
        const int k0 = rp[rowc];
        const int k1 = valid ? rp[rowc + 1] : k0;
        // row metadata early, so its latency overlaps the gathers
        const double wdv0 = cg ? wdcur[rowc] : 0.0;
        const float offv = cg ? 0.f : pa.off[rowc];
        const float wtv = cg ? 0.f : pa.wt[rowc];
        const int yv = cg ? 0 : (int)pa.y[rowc];
        double a = 0.0;
        for (int kb = k0; kb < k1; kb += G * RU) {
            int idx[RU];
            float xv[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const int kc = min(kb + gl + u * G, k1 - 1);
                idx[u] = ci[kc];
                if (HASVAL) xv[u] = val ? val[kc] : 1.0f;          // (a binary partition in a valued handle: x * 1.0 == x)
            }
            double vv[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) vv[u] = v[idx[u]];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const bool in = (kb + gl + u * G) < k1;
                const double term = HASVAL ? vv[u] * (double)xv[u] : vv[u];
                a += in ? term : 0.0;
            }
        }
        a = group_allreduce_sum<G>(a);
        if (valid && gl == 0) {
            const double t = a + vb;
            double cf;
            if (cg) {
                cf = wdv0 * t;
            } else {
                double loss, wdv;
                row_eval(t + (double)offv, yv, (double)wtv, loss, wdv, cf);
                wdnew[row] = wdv;
                red[0] += loss;
            }
            coef[row] = cf;
            red[1] += cf;
        }
    }
    block_allreduce_sum<2>(red, scratch);
    if (threadIdx.x == 0) { pr.lossp[b] = red[0]; pr.csump[b] = red[1]; }
}

// ------------------------------------------------------------------------------------------------
// sparse X pass, part 2: CSC column segments ("items", <= 512 entries) -> X' coef (gather of coef)
// ------------------------------------------------------------------------------------------------
// One round per item: G lanes x CU entries in flight. Short items (<= 64 entries: the long tail of
// rare features, mostly 1-3 entries) take an 8-lane group; long items (65..512) a whole wave.
#define CU 8
template <int G, bool HASVAL>
__global__ void __launch_bounds__(256)
k_colpass_items(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq, int gx)
{
    int pi_, bx_;
    if (!xcd_map(nq, gx, pi_, bx_)) return;
    const int q = qlist[pi_];
    ProbDev &pr = probs[q];
    if (pr.phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    constexpr bool LONG = (G == 64);
    const int nlist = LONG ? pa.n_long : pa.n_short;
    const int32_t *__restrict__ list = LONG ? pa.items_long : pa.items_short;
    const int slot = bx_ * (256 / G) + threadIdx.x / G;
    const int gl = threadIdx.x % G;
    if (bx_ * (256 / G) >= nlist) return;
    const bool valid = slot < nlist;
    const int item = list[min(slot, nlist - 1)];
    const double *__restrict__ coef = pr.coef;
    const int32_t *__restrict__ cri = pa.cri;
    const float *__restrict__ cval = pa.cval;
    const int k0 = pa.item_ptr[item];
    const int k1 = valid ? pa.item_ptr[item + 1] : k0;
    double a = 0.0;
    for (int kb = k0; kb < k1; kb += G * CU) {
        int idx[CU];
        float xv[CU];
#pragma unroll
        for (int u = 0; u < CU; u++) {
            const int kc = min(kb + gl + u * G, k1 - 1);
            idx[u] = cri[kc];
            if (HASVAL) xv[u] = cval ? cval[kc] : 1.0f;
        }
        double cc[CU];
#pragma unroll
        for (int u = 0; u < CU; u++) cc[u] = coef[idx[u]];
#pragma unroll
        for (int u = 0; u < CU; u++) {
            const bool in = (kb + gl + u * G) < k1;
            const double term = HASVAL ? cc[u] * (double)xv[u] : cc[u];
            a += in ? term : 0.0;
        }
    }
    a = group_allreduce_sum<G>(a);
    if (valid && gl == 0) pr.parts[pa.item_dst[item]] = a;     // real items only are listed
}

// ------------------------------------------------------------------------------------------------
// sparse X pass on the sliced-ELL copies: one THREAD per row / per column item, EVERY gather served by LDS.
// Measured (tools/sparse_probe.hip, profiles/r1_notes.md): lane-group kernels are bound by the texture addresser (~1
// divergent lane address per clock per CU) and by their semi-coalesced index reads; thread-per-row kernels that gather
// from global memory by the L2 random-request rate (a 64-byte line moves for every 8 bytes used: the 25 % "cold" gathers
// of the one-hot data cost as much as the 75 % served by a 4096-entry LDS copy). Here the gathered vector itself is cut
// into slices that fit in LDS and staged slice by slice (coalesced 16-byte loads), the index streams are uint16 (slice- /
// block-local ids: half the bytes of the dominant stream) in packs of four per work item, pack p of 64 consecutive work
// items one contiguous 512-byte load, every lane has LSU such loads + 4 LSU LDS reads in flight and there is no
// cross-lane reduction.
// Padding entries point at an LDS slot that holds 0.0 (x + 0.0 == x): no length masks, no per-item length loads.
// Each sum runs sequentially over the item's entries (ascending library column id; ascending row inside an item) with
// contraction off -- the same order as before the slicing, so results are unchanged bit for bit.
// ------------------------------------------------------------------------------------------------
#ifndef LSU
#define LSU 6          // 8-byte index loads (4 ids each) in flight per lane in the deep loop
#endif
typedef unsigned int u2v_t __attribute__((ext_vector_type(2)));
typedef float f4v_t __attribute__((ext_vector_type(4)));

// the four gathers + adds of one pack (ids q, values xv), in entry order
template <bool HASVAL>
__device__ __forceinline__ double pack_sum(double a, u2v_t q, f4v_t xv, const double *__restrict__ lds)
{
#pragma clang fp contract(off)
#if defined(MLX_ABLATE) && (MLX_ABLATE & 1)      /* timing experiments only (tools/ablate_build.sh): no LDS gathers */
    const double c0 = (double)(q.x & 0xFFFFu), c1 = (double)(q.x >> 16), c2 = (double)(q.y & 0xFFFFu), c3 = (double)(q.y >> 16);
#else
    const double c0 = lds[q.x & 0xFFFFu], c1 = lds[q.x >> 16], c2 = lds[q.y & 0xFFFFu], c3 = lds[q.y >> 16];
#endif
    a = a + (HASVAL ? c0 * (double)xv.x : c0);
    a = a + (HASVAL ? c1 * (double)xv.y : c1);
    a = a + (HASVAL ? c2 * (double)xv.z : c2);
    a = a + (HASVAL ? c3 * (double)xv.w : c3);
    return a;
}

template <bool NT>
__device__ __forceinline__ u2v_t pack_load(const uint16_t *__restrict__ idx, int base, int kk, int lane)
{
#if defined(MLX_ABLATE) && (MLX_ABLATE & 2)      /* timing experiments only: no index loads */
    u2v_t q; q.x = (unsigned)((lane * 37 + kk * 101 + base) & 0x3FFF) * 0x10001u; q.y = q.x + 0x00010001u; return q;
#else
#if defined(MLX_ABLATE) && (MLX_ABLATE & 128)     /* timing experiments only: every pack from the first 64 KB of the index array (cache-resident) */
    const u2v_t *__restrict__ ip = reinterpret_cast<const u2v_t *>(idx + (base & 0x3FFF)) + (kk & 7) * 64 + lane;
#else
    const u2v_t *__restrict__ ip = reinterpret_cast<const u2v_t *>(idx + base) + kk * 64 + lane;
#endif
    return NT ? gld_nt(ip) : gld(ip);
#endif
}
template <bool NT>
__device__ __forceinline__ f4v_t pack_load_val(const float *__restrict__ val, int base, int kk, int lane)
{
    if (val == nullptr) return (f4v_t){1.0f, 1.0f, 1.0f, 1.0f};     // a binary partition in a valued handle: x * 1.0 == x
    const f4v_t *__restrict__ vp = reinterpret_cast<const f4v_t *>(val + base) + kk * 64 + lane;
    return NT ? gld_nt(vp) : gld(vp);
}

// Packs kb .. L4-1 of ONE work item (a row's entries in one column slice / a column item), LSU packs in flight: the deep
// loop for long items. base is a multiple of 256 ids; pack p of the 64 lanes' items is contiguous.
template <bool HASVAL, bool NT>
__device__ __forceinline__ double sell_lds_sum(double a, const uint16_t *__restrict__ idx, const float *__restrict__ val, int base,
                                               int kb, int L4, int lane, const double *__restrict__ lds, int zslot)
{
#pragma clang fp contract(off)
    const unsigned zz = (unsigned)zslot | ((unsigned)zslot << 16);
    constexpr int U = HASVAL ? 4 : LSU;                     // valued entries carry a float4 per pack: fewer packs in flight
    for (int k = kb; k < L4; k += U) {
        u2v_t q[U];
        f4v_t xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kk = min(k + u, L4 - 1);              // unconditional clamped loads (a predicated load is waited for alone)
            q[u] = pack_load<NT>(idx, base, kk, lane);
            if (HASVAL) xv[u] = pack_load_val<NT>(val, base, kk, lane);
            if (k + u >= L4) { q[u].x = zz; q[u].y = zz; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) a = pack_sum<HASVAL>(a, q[u], xv[u], lds);
    }
    return a;
}

// The same for the UNSPLIT column items of the reference-order numerics (thousands of entries: one chain per item, so the loop is
// bound by the latency of its index loads unless many are in flight): U packs are added while the next U are being fetched.
template <bool HASVAL, bool NT>
__device__ __forceinline__ double sell_lds_sum_pipe(double a, const uint16_t *__restrict__ idx, const float *__restrict__ val, int base,
                                                    int kb, int L4, int lane, const double *__restrict__ lds, int zslot)
{
#pragma clang fp contract(off)
    const unsigned zz = (unsigned)zslot | ((unsigned)zslot << 16);
    constexpr int U = HASVAL ? 4 : 12;
    u2v_t q[U];
    f4v_t xv[U];
    auto fetch = [&](int k, u2v_t (&qq)[U], f4v_t (&xx)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kk = min(k + u, L4 - 1);
            qq[u] = pack_load<NT>(idx, base, kk, lane);
            if (HASVAL) xx[u] = pack_load_val<NT>(val, base, kk, lane);
            if (k + u >= L4) { qq[u].x = zz; qq[u].y = zz; }
        }
    };
    if (kb < L4) fetch(kb, q, xv);
    for (int k = kb; k < L4; k += U) {
        u2v_t qn[U];
        f4v_t xn[U];
        fetch(min(k + U, L4 - 1), qn, xn);                  // (behind the last batch: one re-read of the last pack, unused)
#pragma unroll
        for (int u = 0; u < U; u++) a = pack_sum<HASVAL>(a, q[u], xv[u], lds);
#pragma unroll
        for (int u = 0; u < U; u++) { q[u] = qn[u]; if (HASVAL) xv[u] = xn[u]; }
    }
    return a;
}

// The deep part of NI work items AT ONCE (reference-order column pass: unsplit column items, 3 .. 64 packs long in the slices behind the
// relayed ones). One item after the other, every batch of LSU packs pays a memory latency of its own -- 2-8 dependent latencies per
// slice, 8 slices per round: 53 of the 130 us a work unit took (profiles/r6_notes.md). Here batch k of ALL live items is in flight
// together (NI x U packs, one latency), the NI chains' adds interleave, and what is left of the one or two longest items once the others
// are through goes to the read-ahead loop. L4s must not increase with the item index (slices are sorted by length); bases / L4s are
// wave-uniform, so the per-item tests are scalar branches.
template <bool HASVAL, bool NT, int NI>
__device__ __forceinline__ void sell_lds_deep_multi(double (&a)[NI], const uint16_t *__restrict__ idx, const float *__restrict__ val,
                                                    const int (&base)[NI], const int (&L4)[NI], int kb, int lane,
                                                    const double *__restrict__ lds, int zslot)
{
#pragma clang fp contract(off)
    const unsigned zz = (unsigned)zslot | ((unsigned)zslot << 16);
    constexpr int U = HASVAL ? 1 : 2, TAIL = 2;              // items left to the read-ahead loop
    static_assert(NI > TAIL, "items");
    int k = kb;
    for (; k < L4[TAIL]; k += U) {
        u2v_t q[NI][U];
        f4v_t xv[NI][U];
#pragma unroll
        for (int i = 0; i < NI; i++)
            if (k < L4[i]) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int kk = min(k + u, L4[i] - 1);
                    q[i][u] = pack_load<NT>(idx, base[i], kk, lane);
                    if (HASVAL) xv[i][u] = pack_load_val<NT>(val, base[i], kk, lane);
                    if (k + u >= L4[i]) { q[i][u].x = zz; q[i][u].y = zz; }
                }
            }
#pragma unroll
        for (int i = 0; i < NI; i++)
            if (k < L4[i]) {
#pragma unroll
                for (int u = 0; u < U; u++) a[i] = pack_sum<HASVAL>(a[i], q[i][u], xv[i][u], lds);
            }
    }
#pragma unroll
    for (int i = 0; i < TAIL; i++)
        if (k < L4[i]) a[i] = sell_lds_sum_pipe<HASVAL, NT>(a[i], idx, val, base[i], k, L4[i], lane, lds, zslot);
}

// The first KP packs of NI work items at once (NI*KP loads in flight, ONE memory latency for all of them): what makes
// the short items -- the 0-3 entries a row has in a cold column slice, the rare features' column items -- cheap, where a
// loop over items pays a full dependent-load latency per item. bases / L4s are wave-uniform.
template <bool HASVAL, bool NT, int NI, int KP>
__device__ __forceinline__ void sell_lds_first(double (&a)[NI], const uint16_t *__restrict__ idx, const float *__restrict__ val,
                                               const int (&base)[NI], const int (&L4)[NI], int k0, int lane,
                                               const double *__restrict__ lds, int zslot)
{
#pragma clang fp contract(off)
    const unsigned zz = (unsigned)zslot | ((unsigned)zslot << 16);
    u2v_t q[NI][KP];
    f4v_t xv[NI][KP];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int u = 0; u < KP; u++) {
            const int kk = max(min(k0 + u, L4[i] - 1), 0);
            q[i][u] = pack_load<NT>(idx, base[i], kk, lane);
            if (HASVAL) xv[i][u] = pack_load_val<NT>(val, base[i], kk, lane);
            if (k0 + u >= L4[i]) { q[i][u].x = zz; q[i][u].y = zz; }
        }
#pragma unroll
    for (int i = 0; i < NI; i++) {
#pragma unroll
        for (int u = 0; u < KP; u++) a[i] = pack_sum<HASVAL>(a[i], q[i][u], xv[i][u], lds);
        // keep the LDS reads of later items behind this item's adds: hoisting all NI*KP*4 of them costs 8 VGPRs per pack
        if ((i & 1) == 1) asm volatile("" ::: "memory");
    }
}

// Staging of up to STAGE_MAX2 * 1024 double2 from global memory into LDS through registers, so that the loads of one
// slice can be in flight while the previous slice is still being read (issue stage_fetch, work, barrier, stage_store).
#define STAGE_MAX2 10
struct StageRegs { double2 r[STAGE_MAX2]; };
__device__ __forceinline__ void stage_fetch(StageRegs &R, const double *__restrict__ src, int cnt, int tid)
{
    typedef double d2v_t __attribute__((ext_vector_type(2)));
    const d2v_t *__restrict__ s2 = reinterpret_cast<const d2v_t *>(src);
    const int np2 = cnt >> 1;
#pragma unroll
    for (int u = 0; u < STAGE_MAX2; u++) {
        const int i = tid + u * 1024;
        const d2v_t t = gld(s2 + min(i, max(np2 - 1, 0)));
        R.r[u].x = t.x; R.r[u].y = t.y;
    }
}
__device__ __forceinline__ void stage_store(const StageRegs &R, double *__restrict__ lds, const double *__restrict__ src, int cnt, int tid)
{
    const int np2 = cnt >> 1;
#pragma unroll
    for (int u = 0; u < STAGE_MAX2; u++) {
        const int i = tid + u * 1024;
        if (i < np2) { lds[2 * i] = R.r[u].x; lds[2 * i + 1] = R.r[u].y; }
    }
    if ((cnt & 1) && tid == 0) lds[cnt - 1] = src[cnt - 1];
}

// The first pack of NI work items whose ids index GLOBAL memory (the cold column slices of the row pass): NI id loads in
// flight, then 4 NI gathers in flight -- two memory latencies for all of them. Padding id 0xFFFF adds 0.0.
template <bool HASVAL, bool NT, int NI>
__device__ __forceinline__ void sell_gather_round(double *a, const uint16_t *__restrict__ idx, const float *__restrict__ val,
                                                  const int *base, const int *L4, int k, int lane,
                                                  const double *__restrict__ src)
{
#pragma clang fp contract(off)
    u2v_t q[NI];
    f4v_t xv[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int kk = max(min(k, L4[i] - 1), 0);
        q[i] = pack_load<NT>(idx, base[i], kk, lane);
        if (HASVAL) xv[i] = pack_load_val<NT>(val, base[i], kk, lane);
        if (k >= L4[i]) { q[i].x = 0xFFFFFFFFu; q[i].y = 0xFFFFFFFFu; }
    }
    double c[NI][4];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const unsigned id[4] = {q[i].x & 0xFFFFu, q[i].x >> 16, q[i].y & 0xFFFFu, q[i].y >> 16};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double g = gld(src + (id[e] == 0xFFFFu ? 0u : id[e]));        // unconditional load, clamped address
            c[i][e] = id[e] == 0xFFFFu ? 0.0 : g;
        }
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
        a[i] = a[i] + (HASVAL ? c[i][0] * (double)xv[i].x : c[i][0]);
        a[i] = a[i] + (HASVAL ? c[i][1] * (double)xv[i].y : c[i][1]);
        a[i] = a[i] + (HASVAL ? c[i][2] * (double)xv[i].z : c[i][2]);
        a[i] = a[i] + (HASVAL ? c[i][3] * (double)xv[i].w : c[i][3]);
    }
}

// Row pass. One 1024-thread workgroup = (problem, chunk of 16 GPW row groups of 64 rows); a wave owns GPW consecutive
// groups, a thread one row of each and keeps their running sums in registers. The HOT slice of the gathered vector (d or
// w_new; the slw most frequent columns) is staged in LDS once, then per slice -- hot first, then the cold ones in
// ascending column order, so a row's entries are added in ascending column id as before -- the block offsets of the
// wave's groups come from ONE load and the packs of all its groups are fetched together: a handful of dependent memory
// latencies per workgroup, where a loop over groups paid three per group and slice (in-kernel phase timing,
// profiles/r2_notes.md). Workgroups are mapped XCD-aware: all chunks of problem p run on XCD p % 8, so the vector they all
// stage and gather stays in that L2.
#ifndef ROW_KP
#define ROW_KP 1       // packs per group and round of the hot slice
#endif
#ifndef RO_LONG_T
#define RO_LONG_T 64   // reference-order column pass: slices of more packs than this run as a relay over the workgroup's waves (16 / 32 / 64: no difference, profiles/r5_notes.md)
#endif
// RO (reference-order numerics, mlx_ro_kernels.h): every slice is hot, so a row's entries join ONE running sum in ascending column
// id -- Xv's order (llf/LogisticRegressionL2.java:115-129); the row map evaluates the portable exp / log1p the oracle's twin uses and
// leaves the row's loss in rowtmp[] for the step's sequential fold; no per-group partial sums.
template <bool HASVAL, bool NT, int GPW, bool RO = false>
__global__ void __launch_bounds__(1024)
k_rowpass_lds(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq, int gx,
              int cold_sep, int *__restrict__ coldone = nullptr /* RO: the column pass's per-problem unit counters, cleared here */)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) double vs[];      // [slw + 1]: the staged hot slice, then the zero slot
    PT_INIT;
    int pi_, bx_;
    if (!xcd_map(nq, gx, pi_, bx_)) return;
    const int q = qlist[pi_];
    ProbDev &pr = probs[q];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    const int c = bx_;
    if (c >= pa.nblk) return;
    // (reference order, column pass in one launch: its counter of this problem's finished work units starts every tick at 0 -- cleared
    //  HERE, by the row pass that precedes every column pass of the problem on the same stream, served at the memory side like the
    //  column pass's own accesses to it)
    if (RO && coldone != nullptr && c == 0 && threadIdx.x == 0) __hip_atomic_store(coldone + q, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool cg = (phase == PH_CG);
    const double *__restrict__ v = cg ? pr.d : pr.w_new;
    const double *__restrict__ wdcur = pr.wd[pr.dsel];
    double *__restrict__ wdnew = pr.wd[pr.dsel ^ 1];
    double *__restrict__ coef = pr.coef;
    const int nf = pa.n_feat, slw = pa.slw, ncs = pa.n_cs, nhs = pa.n_hs, ngr = pa.n_rgroups, l = pa.l;
    const int g0 = c * (16 * GPW);
    const int gcount = min(16 * GPW, ngr - g0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wg0 = wave * GPW;                               // this wave's first row group within the chunk
    const uint16_t *__restrict__ rs_idx = pa.rs_idx;
    const float *__restrict__ rs_val = pa.rs_val;
    double acc[GPW];
#pragma unroll
    for (int i = 0; i < GPW; i++) acc[i] = 0.0;
    // block offsets of this wave's GPW consecutive groups in slice sl: one load, then lane broadcasts (wave-uniform scalars)
    int base[GPW], L4[GPW];
    int kmax;
    auto offsets = [&](int sl) {
        const int32_t *__restrict__ ptr = pa.rs_ptr + (int64_t)sl * ngr + g0;
        const int pv = gld(ptr + min(wg0 + min(lane, GPW), gcount));
        kmax = 0;
#pragma unroll
        for (int i = 0; i < GPW; i++) {
            base[i] = __builtin_amdgcn_readlane(pv, i);
            const int nx = __builtin_amdgcn_readlane(pv, i + 1);
            L4[i] = (wg0 + i < gcount) ? (nx - base[i]) >> 8 : 0;
            kmax = max(kmax, L4[i]);
        }
    };
    auto stage = [&](int sl) {
        // hot slice sl: columns [sl*slw, sl*slw + cnt) of the gathered vector, then the zero slot
        StageRegs SR;
        const int cnt = min(slw, nf - sl * slw);
        stage_fetch(SR, v + (int64_t)sl * slw, cnt, tid);    // all loads of the slice in flight at once
        stage_store(SR, vs, v + (int64_t)sl * slw, cnt, tid);
        if (tid == 0) vs[slw] = 0.0;
    };
    constexpr int KP = HASVAL ? 1 : ROW_KP;
    // ---- hot slice 0 (straight-line: the common case is this slice and nothing else in this kernel)
    stage(0);
    PT_MARK(0);
    __syncthreads();
    PT_MARK(1);
    offsets(0);
    PT_MARK(2);
    for (int k = 0; k < kmax; k += KP)
        sell_lds_first<HASVAL, NT, GPW, KP>(acc, rs_idx, rs_val, base, L4, k, lane, vs, slw);
    PT_MARK(3);
    // ---- further hot slices, staged one after the other
    for (int sl = 1; sl < nhs; sl++) {
        __syncthreads();                                     // every wave is done with the previous slice
        stage(sl);
        __syncthreads();
        offsets(sl);
        for (int k = 0; k < kmax; k += KP)
            sell_lds_first<HASVAL, NT, GPW, KP>(acc, rs_idx, rs_val, base, L4, k, lane, vs, slw);
    }
    // ---- cold slices, unless k_rowcold (launched just before) left their sums in coef[]
    // (inline, the cold sum runs in accumulators of its own and joins the hot sum once, as k_rowcold's does through coef[]: a row's sum
    // is the same bits whichever way the handle's chunking sends it -- MLX_ROW_NG / DESIGN.md section 8)
    const bool add_cold = cold_sep && ncs > nhs;
    if (!cold_sep && ncs > nhs) {
        double accc[GPW];
#pragma unroll
        for (int i = 0; i < GPW; i++) accc[i] = 0.0;
        for (int sl = nhs; sl < ncs; sl++) {
            offsets(sl);
            const double *__restrict__ src = v + (int64_t)nhs * slw + (int64_t)(sl - nhs) * 65535;
            // (GPW gathers x 4 in flight; 8 groups, or 4 valued ones, go in two halves: their 32 results do not fit beside the rest)
            constexpr int NIH = (GPW > 4 || (HASVAL && GPW > 2)) ? GPW / 2 : GPW;
            for (int k = 0; k < kmax; k++) {
                sell_gather_round<HASVAL, NT, NIH>(accc, rs_idx, rs_val, base, L4, k, lane, src);
                if (NIH < GPW) sell_gather_round<HASVAL, NT, NIH>(accc + NIH, rs_idx, rs_val, base + NIH, L4 + NIH, k, lane, src);
            }
            PT_MARK(4);
        }
#pragma unroll
        for (int i = 0; i < GPW; i++) acc[i] = acc[i] + accc[i];
    }
    // row maps: the loads of (up to) four of the wave's rows are issued before the first is used (clamped, unconditional)
    // The loss and the coefficient sum (the intercept's column of X'c) leave the kernel as ONE partial per 64-row GROUP -- a wave
    // all-reduce over the group's rows -- not per workgroup: the step adds the groups' partials in group order, so the sums do not
    // depend on how many groups a workgroup owns (chosen at mlx_finalize from the handle's total work).
    const double vb = v[nf];
    constexpr int EH = GPW < 4 ? GPW : 4;
#pragma unroll
    for (int i0 = 0; i0 < GPW; i0 += EH) {
        int rowi[EH];
        bool ok[EH];
        double wdv0[EH], zc[EH];
        float offv[EH], wtv[EH];
        int yv[EH];
#pragma unroll
        for (int i = 0; i < EH; i++) {
            const int row = (g0 + wg0 + i0 + i) * 64 + lane;
            ok[i] = (wg0 + i0 + i < gcount) && row < l;
            rowi[i] = min(row, l - 1);
            wdv0[i] = cg ? gld_nt(wdcur + rowi[i]) : 0.0;      // (read once per tick)
            zc[i] = add_cold ? gld(coef + rowi[i]) : 0.0;
            offv[i] = cg ? 0.f : gld(pa.off + rowi[i]);
            wtv[i] = cg ? 0.f : gld(pa.wt + rowi[i]);
            yv[i] = cg ? 0 : (int)gld(pa.y + rowi[i]);
        }
#pragma unroll
        for (int i = 0; i < EH; i++) {
            double lossv = 0.0, cfv = 0.0;
            if (ok[i]) {
                const double t = (add_cold ? acc[i0 + i] + zc[i] : acc[i0 + i]) + vb;
                if (cg) {
                    cfv = wdv0[i] * t;
                } else {
                    double wdv;
                    row_eval<RO>(t + (double)offv[i], yv[i], (double)wtv[i], lossv, wdv, cfv);
                    gst(wdnew + rowi[i], wdv);
                    if (RO) gst(pr.rowtmp + rowi[i], lossv);
                }
                gst(coef + rowi[i], cfv);
            }
            if (!RO && wg0 + i0 + i < gcount) {                // (wave-uniform)
                const double cs = wave_allreduce_sum(cfv);
                const double ls = cg ? 0.0 : wave_allreduce_sum(lossv);
                if (lane == 0) { pr.csump[g0 + wg0 + i0 + i] = cs; if (!cg) pr.lossp[g0 + wg0 + i0 + i] = ls; }
            }
        }
    }
    PT_MARK(6);
    PT_MARK(5);
}

// The cold column slices of the row pass as a launch of their own (12 % of the entries of the one-hot configs, but 27 % of
// the fused kernel's time: two dependent L2 latencies per round on a workgroup that owns a whole CU's LDS). Here they run
// at full occupancy -- 256-thread workgroups, no LDS, a wave owns 4 consecutive row groups and a lane one row of each --
// and leave a row's cold sum (entries in ascending column id) in coef[row]; k_rowpass_lds adds it to the row's hot sum.
template <bool HASVAL, bool NT>
__global__ void __launch_bounds__(256)
k_rowcold(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq, int gx)
{
#pragma clang fp contract(off)
    int pi_, bx_;
    if (!xcd_map(nq, gx, pi_, bx_)) return;
    ProbDev &pr = probs[qlist[pi_]];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    const int ncs = pa.n_cs, nhs = pa.n_hs, ngr = pa.n_rgroups, slw = pa.slw, l = pa.l;
    if (!pa.sell || ncs <= nhs) return;
    constexpr int GC = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = bx_ * (4 * GC) + wave * GC;
    if (g0 >= ngr) return;
    const int gcount = min(GC, ngr - g0);
    const double *__restrict__ v = (phase == PH_CG) ? pr.d : pr.w_new;
    const uint16_t *__restrict__ rs_idx = pa.rs_idx;
    const float *__restrict__ rs_val = pa.rs_val;
    double acc[GC];
#pragma unroll
    for (int i = 0; i < GC; i++) acc[i] = 0.0;
    for (int sl = nhs; sl < ncs; sl++) {
        const int32_t *__restrict__ ptr = pa.rs_ptr + (int64_t)sl * ngr + g0;
        const int pv = ptr[min(lane, gcount)];
        int base[GC], L4[GC];
        int kmax = 0;
#pragma unroll
        for (int i = 0; i < GC; i++) {
            base[i] = __builtin_amdgcn_readlane(pv, i);
            const int nx = __builtin_amdgcn_readlane(pv, i + 1);
            L4[i] = (i < gcount) ? (nx - base[i]) >> 8 : 0;
            kmax = max(kmax, L4[i]);
        }
        const double *__restrict__ src = v + (int64_t)nhs * slw + (int64_t)(sl - nhs) * 65535;
        for (int k = 0; k < kmax; k++) sell_gather_round<HASVAL, NT, GC>(acc, rs_idx, rs_val, base, L4, k, lane, src);
    }
    double *__restrict__ coef = pr.coef;
#pragma unroll
    for (int i = 0; i < GC; i++) {
        const int row = (g0 + i) * 64 + lane;
        if (i < gcount && row < l) coef[row] = acc[i];
    }
}

// Column pass with the row coefficients in LDS. One workgroup = (problem, work unit): the unit's row block of `coef`
// (<= 20 160 doubles) is staged once, then every wave walks item slices of that block: one THREAD per item, entry k of
// the 64 items one coalesced 128-B index load, the gather served by LDS instead of the L2 request path (which is what
// bounds the global-gather form: ~250 G random 8-byte requests/s chip-wide). Sums run in row order, contraction off.
// RO (reference-order numerics, mlx_ro_kernels.h): launched once per row block (only_blk), in block order. Items are unsplit inside
// a block and start from the sum their column reached in the earlier blocks (item_init), so a column's sum is ONE chain over its rows
// in ascending order -- XTv's order (llf/LogisticRegressionL2.java:140-145); a column's last item also stores the sum at xtc[column]
// (ProbDev::c0f). (The intercept's column -- the sum of all coefficients in row order -- is folded by the step kernel, k_ro_step.)
// Reference-order numerics: the intercept's column of X'c -- XTv[n-1] += v[i] * 1.0 over ALL rows in row order
// (llf/LogisticRegressionL2.java:143-145; the bias entry closes every row) -- is a LITERAL chain over the l row coefficients: its running
// sum stays as small as its terms, so mlx_seqfold.h's exact parallel fold has nothing to hold on to. It needs the row pass only, so it
// runs BESIDE the column pass: the first `lead` workgroups of the column pass's first launch are chain workgroups -- the dispatcher
// starts workgroups in index order, so they are resident before the pass fills the chip -- one problem per wave; 1/8 of
// the CUs for the ~0.2 ms the chain takes, and the step kernel finds csump[0] ready instead of running the chain on its critical path
// (tried before: the chain on one wave of a pass workgroup -- the pass keeps the CU's LDS pipe and registers full, +160 ... +660 us
// per tick; a kernel of its own on a side stream -- it is placed only when the pass drains, +365 us; profiles/r6_notes.md).
// Eight waves per chain workgroup (two per SIMD: four chains per SIMD would share its one add per four cycles and crawl at 48 cycles
// per term), one problem each. A wave copies 1 024 coefficients at a time into an LDS region of its own (coalesced loads, the next 1 024
// in flight behind the chain) and walks them with 16-byte broadcast reads -- every lane adds the same terms, one v_add_f64 per term;
// the reads of the next eight terms are pinned in front of the eight adds they hide behind (left to itself the scheduler sinks every
// read next to its use and the chain pays one LDS latency per pair).
constexpr int RO_CSUM_WAVES = 8, RO_CSUM_CH = 1024;
__device__ __forceinline__ void ro_csum_chain(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq, double *lds)
{
#pragma clang fp contract(off)
    typedef double d2v_t __attribute__((ext_vector_type(2)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave >= RO_CSUM_WAVES) return;
    const int qi = blockIdx.x * RO_CSUM_WAVES + wave;
    if (qi >= nq) return;
    ProbDev &pr = probs[qlist[qi]];
    if (pr.phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    if (pa.dense || !pa.sell || pa.n_cunits <= 0) return;
    const int l = pa.l;
    const double *__restrict__ coef = pr.coef;
    double *reg = lds + wave * (2 * RO_CSUM_CH);              // two regions of 1 024 doubles: being chained / being filled
    double q[16];
    auto fetch = [&](int c) {                                 // chunk c -> registers (terms behind the end: -0.0, x + (-0.0) == x)
#pragma unroll
        for (int u = 0; u < 16; u++) { const int j = c * RO_CSUM_CH + u * 64 + lane; const double x = gld(coef + min(j, l - 1)); q[u] = j < l ? x : -0.0; }
    };
    auto put = [&](int c) {
        double *r = reg + (c & 1) * RO_CSUM_CH;
#pragma unroll
        for (int u = 0; u < 16; u++) r[u * 64 + lane] = q[u];
    };
    const int nch = (l + RO_CSUM_CH - 1) / RO_CSUM_CH;
    fetch(0);
    put(0);
    double s = 0.0;
    for (int c = 0; c < nch; c++) {
        fetch(c + 1);                                         // (behind the end: clamped re-reads, all terms -0.0)
        const d2v_t *c2 = reinterpret_cast<const d2v_t *>(reg + (c & 1) * RO_CSUM_CH);
        d2v_t A0 = c2[0], A1 = c2[1], A2 = c2[2], A3 = c2[3], B0, B1, B2, B3;
        for (int j = 0; j < RO_CSUM_CH / 2; j += 8) {
            const int jb = j + 4, ja = min(j + 8, RO_CSUM_CH / 2 - 4);
            B0 = c2[jb]; B1 = c2[jb + 1]; B2 = c2[jb + 2]; B3 = c2[jb + 3];
            __builtin_amdgcn_sched_barrier(0);
            s = s + A0.x; s = s + A0.y; s = s + A1.x; s = s + A1.y; s = s + A2.x; s = s + A2.y; s = s + A3.x; s = s + A3.y;
            __builtin_amdgcn_sched_barrier(0);
            A0 = c2[ja]; A1 = c2[ja + 1]; A2 = c2[ja + 2]; A3 = c2[ja + 3];
            __builtin_amdgcn_sched_barrier(0);
            s = s + B0.x; s = s + B0.y; s = s + B1.x; s = s + B1.y; s = s + B2.x; s = s + B2.y; s = s + B3.x; s = s + B3.y;
            __builtin_amdgcn_sched_barrier(0);
        }
        put(c + 1);                                           // (the other region: nobody reads it before the next trip)
    }
    if (lane == 0) gst(pr.csump, s);
}

// Accesses served at the memory side (agent-scope relaxed atomics: global_load / global_store with sc1), for the few values one workgroup
// hands to another INSIDE a launch: no cache maintenance (a device-scope fence writes back / invalidates a whole L2, profiles/r2_notes.md).
__device__ __forceinline__ void st_coh(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_coh(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_coh(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool HASVAL, bool NT, bool RO = false>
__global__ void __launch_bounds__(1024)
k_colpass_lds(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq, int gx, int only_blk, int lead,
              int *__restrict__ coldone = nullptr /* RO, only_blk < 0: [total problems] finished work units of this launch per problem */)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) double cf[];      // [rblk_rows + 1]: the block's coefficients, then the zero slot
    PT_INIT;
    if (RO && (int)blockIdx.x < lead) {
        if ((int)blockIdx.x * RO_CSUM_WAVES < nq) ro_csum_chain(parts, probs, qlist, nq, cf);      // (needs 128 KiB of the launch's dynamic LDS: mlxk_xpass_csr)
        return;
    }
    // (reference order, only_blk < 0: ALL row blocks in one launch -- the grid is [chain workgroups][block 0's units of every problem]
    //  [block 1's units] ...; a unit of a later block waits for its problem's earlier units below)
    const bool merged = RO && only_blk < 0;
    const int XG = ((nq + 7) / 8 * 8) * gx;
    if (merged) only_blk = ((int)blockIdx.x - lead) / XG;
    int pi_, bx_;
    if (!xcd_map(nq, gx, pi_, bx_, RO ? lead + (merged ? only_blk * XG : 0) : 0)) return;
    const int q = qlist[pi_];
    ProbDev &pr = probs[q];
    if (pr.phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    int ro_need = 0;
    if (RO) {
        // (reference order: the launch of row block only_blk; the first unit of every block sits behind cw_blk[], mlx_api.hip)
        if (only_blk >= pa.n_rblk) return;
        const int u0 = gld(pa.cw_blk + pa.n_cunits + only_blk), u1 = gld(pa.cw_blk + pa.n_cunits + only_blk + 1);
        bx_ += u0;
        if (bx_ >= u1) return;
        ro_need = u0;                                          // units of the earlier blocks: all of them hand sums on to this block
    }
    if (bx_ >= pa.n_cunits) return;
    const int blk = pa.cw_blk[bx_];
    if (RO && blk != only_blk) return;
    const int s0 = pa.cw_slice[bx_], s1 = pa.cw_slice[bx_ + 1];
    const int r0 = blk * pa.rblk_rows;
    const int nr = min(pa.rblk_rows, pa.l - r0);
    {
        StageRegs SR;
        const double *__restrict__ src = pr.coef + r0;      // r0 is a multiple of 64: 16-byte aligned pairs
        stage_fetch(SR, src, nr, threadIdx.x);               // <= 20 160 doubles: all loads of the block in flight at once
        stage_store(SR, cf, src, nr, threadIdx.x);
        if (threadIdx.x == 0) cf[pa.rblk_rows] = 0.0;
    }
    __syncthreads();
    int *cnt_mine = nullptr;
    if (merged) {
        // The problem's counter of finished work units: cleared by the row pass in front of this launch (k_rowpass_lds<.., RO>). A unit
        // of a later block has its coefficients
        // staged by now and waits until every unit of the earlier blocks has counted itself (units are dispatched in grid order --
        // earlier blocks first -- so whoever it waits for is running or done; a wait that outlasts 5 s gives up and flags the problem).
        cnt_mine = coldone + q;
        if (threadIdx.x == 0) {
            if (ro_need > 0) {
                const unsigned long long t0 = wall_clock64();
                while (ld_coh(cnt_mine) < ro_need) {
                    __builtin_amdgcn_s_sleep(8);
                    if (wall_clock64() - t0 > 500000000ull) { pr.status = ST_SYNC; break; }      // (100 MHz: 5 s -- time-sliced GPUs included, nobody waits that long for a unit that is running)
                }
            }
        }
        __syncthreads();
    }
    PT_MARK(8);
#if defined(MLX_PHASE_TIMING) && defined(MLX_PT_PASSES_ONLY)
    if (threadIdx.x == 0) atomicAdd(&g_phase[13], 100ull);       // work units run (x 0.01 in mlx_debug_phase_times)
#endif
    const uint16_t *__restrict__ cs_idx = pa.cs_idx;
    const float *__restrict__ cs_val = pa.cs_val;
    const int32_t *__restrict__ cs_ptr = pa.cs_ptr;
    const int32_t *__restrict__ item_dst = RO ? pa.item_chain : pa.item_dst;      // (RO: where the running sum is handed on, PartDev::item_chain)
    double *__restrict__ out = pr.parts;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // COL_B item slices per wave and round: their offsets, then their destinations and first packs, are fetched together
    // (three dependent latencies per round instead of per slice); items longer than the first packs continue in the deep loop
    constexpr int COL_B = 8, KP = HASVAL ? 1 : 2;
    const int zs = pa.rblk_rows;
    double *__restrict__ xtc = pr.c0f;
    int s0n = s0;                                           // first slice of the loop below
    if (RO) {
        // The LONG slices of the unit (unsplit columns of thousands of entries; a block's slices are sorted by length) as a RELAY over
        // the 16 waves. A column's sum is one chain of dependent adds, so a slice cannot be split between waves -- but only the adds
        // are sequential: wave w takes the batches b = w, w + 16, ... of PB packs, fetches its index packs and gathers its coefficients
        // from LDS at once, and when batch b - 1 is through (relay_turn) continues the 64 running sums (relay_run) with its 4 PB
        // adds per lane and hands on. One wave alone on such a slice waits for HBM every 12 packs (measured: ~350 us for the hottest
        // slice of a configs[2] block); here 16 batches are in flight and the chain runs at the speed of the adds.
        __shared__ double relay_run[64];
        __shared__ volatile int relay_turn;
        constexpr int PB = HASVAL ? 2 : 8, LONG_T = RO_LONG_T;
        for (; s0n < s1; s0n++) {
            const int b0 = __builtin_amdgcn_readfirstlane(gld(cs_ptr + s0n));
            const int L = (__builtin_amdgcn_readfirstlane(gld(cs_ptr + s0n + 1)) - b0) >> 8;
            if (L <= LONG_T) break;
            // (one word per item: >= 0 the item the sum is handed to, ~column for the column's last item, INT32_MIN for padding; the start
            //  value is the item's own slot -- 0.0 for ever where no earlier item hands a sum on: mlx_api.hip, prep_csr)
            const int code = gld_nt(item_dst + s0n * 64 + lane);
            const int dl = code >= 0 ? code : -1, dla = (code < 0 && code != (int)0x80000000) ? ~code : -1;
            if (wave == 0) relay_run[lane] = merged ? ld_coh(out + s0n * 64 + lane) : gld(out + s0n * 64 + lane);      // (the column's sum over the earlier blocks)
            if (threadIdx.x == 0) relay_turn = 0;
            __syncthreads();
            const unsigned zz = (unsigned)zs | ((unsigned)zs << 16);
            const int nbatch = (L + PB - 1) / PB;
            for (int b = wave; b < nbatch; b += 16) {
                u2v_t q[PB];
                f4v_t xv[PB];
                double c[PB][4];
#pragma unroll
                for (int u = 0; u < PB; u++) {
                    const int kk = b * PB + u;
                    q[u] = pack_load<NT>(cs_idx, b0, min(kk, L - 1), lane);
                    if (HASVAL) xv[u] = pack_load_val<NT>(cs_val, b0, min(kk, L - 1), lane);
                    if (kk >= L) { q[u].x = zz; q[u].y = zz; }
                }
#pragma unroll
                for (int u = 0; u < PB; u++) {
                    c[u][0] = cf[q[u].x & 0xFFFFu]; c[u][1] = cf[q[u].x >> 16]; c[u][2] = cf[q[u].y & 0xFFFFu]; c[u][3] = cf[q[u].y >> 16];
                    if (HASVAL) { c[u][0] = c[u][0] * (double)xv[u].x; c[u][1] = c[u][1] * (double)xv[u].y; c[u][2] = c[u][2] * (double)xv[u].z; c[u][3] = c[u][3] * (double)xv[u].w; }
                }
                while (relay_turn != b) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                double a = relay_run[lane];
#pragma unroll
                for (int u = 0; u < PB; u++) { a = a + c[u][0]; a = a + c[u][1]; a = a + c[u][2]; a = a + c[u][3]; }
                relay_run[lane] = a;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) relay_turn = b + 1;
            }
            __syncthreads();
            if (wave == 0) {
                const double a = relay_run[lane];
                if (dl >= 0) { if (merged) st_coh(out + dl, a); else gst(out + dl, a); }
                if (dla >= 0) gst(xtc + dla, a);
            }
            __syncthreads();
        }
        PT_MARK(12);
    }
    // (slice s0 + wave + 16 t: slices are sorted by length, so this deals the long ones evenly; batches of consecutive slices
    // per wave -- one offset load instead of 16 -- put a hot unit's 16 long slices on 2 waves: 196 -> 278 us)
    for (int sb = s0n + wave; sb < s1; sb += 16 * COL_B) {
        int base[COL_B], L4[COL_B], dst[COL_B], dlast[COL_B];
        double a[COL_B];
#pragma unroll
        for (int u = 0; u < COL_B; u++) {
            const int sl = sb + 16 * u;
            const int sc = min(sl, s1 - 1);
            base[u] = __builtin_amdgcn_readfirstlane(gld(cs_ptr + sc));
            const int nx = __builtin_amdgcn_readfirstlane(gld(cs_ptr + sc + 1));
            L4[u] = (sl < s1) ? (nx - base[u]) >> 8 : 0;
#if defined(MLX_ABLATE) && (MLX_ABLATE & 8)      /* timing experiments only: no item meta data */
            dst[u] = (sl < s1) ? sc * 64 + lane : -1; a[u] = 0.0; dlast[u] = -1;
            if (false) {
#else
#if defined(MLX_ABLATE) && (MLX_ABLATE & 256)     /* timing experiments only: every slice's meta data from slice (sc & 63): cache-resident */
#define SCM ((sc & 63) + s0)
#else
#define SCM sc
#endif
            const int dl = gld_nt(item_dst + SCM * 64 + lane);   // unconditional, clamped; read once per tick
            dst[u] = (sl < s1) ? dl : -1;
            a[u] = 0.0;
            dlast[u] = -1;
            if (RO) {
#endif
                // Every load of the round unconditional and independent of the others: the first form fetched the start value under
                // `if (di >= 0)` and the last-item flag under `if (sl < s1)` -- two branches per slice, and the compiler waited for ALL
                // outstanding loads in front of each: eight full memory latencies per round, one after the other (28 of the 130 us of
                // a work unit). Now: ONE word per item (dl: >= 0 the item the sum is handed to, ~column for the column's last item,
                // INT32_MIN for padding) and the item's own hand-over slot as the start value (0.0 for ever where nobody hands a sum on).
                const double ov = merged ? ld_coh(out + SCM * 64 + lane) : gld(out + SCM * 64 + lane);   // (written by the previous block's launch / units)
                dst[u] = (sl < s1 && dl >= 0) ? dl : -1;
                dlast[u] = (sl < s1 && dl < 0 && dl != (int)0x80000000) ? ~dl : -1;
                a[u] = (sl < s1) ? ov : 0.0;
            }
#undef SCM
        }
        PT_MARK(9);
#if defined(MLX_ABLATE) && (MLX_ABLATE & 16)     /* timing experiments only: no packs at all */
        if (false)
#endif
        sell_lds_first<HASVAL, NT, COL_B, KP>(a, cs_idx, cs_val, base, L4, 0, lane, cf, zs);
        PT_MARK(10);
#if defined(MLX_ABLATE) && (MLX_ABLATE & 16)
        if (false) {
#else
        if (RO) {
#endif
            // (unsplit items: the deep parts of the round's slices together, sell_lds_deep_multi)
            if (L4[0] > KP) sell_lds_deep_multi<HASVAL, NT, COL_B>(a, cs_idx, cs_val, base, L4, KP, lane, cf, zs);
        }
#pragma unroll
        for (int u = 0; u < COL_B; u++) {
            if (!RO && L4[u] > KP) a[u] = sell_lds_sum<HASVAL, NT>(a[u], cs_idx, cs_val, base[u], KP, L4[u], lane, cf, zs);
#if defined(MLX_ABLATE) && (MLX_ABLATE & 4)      /* timing experiments only: one store per lane and round instead of up to 16 */
            if (u == 0 && dst[u] >= 0) gst(out + sb * 64 + lane, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]);
#else
            if (dst[u] >= 0) { if (merged) st_coh(out + dst[u], a[u]); else gst(out + dst[u], a[u]); }      // (plain store: phase A reads the slots from L2 right after; a streaming store cost the pass 13 %)
            if (RO && dlast[u] >= 0) gst(xtc + dlast[u], a[u]);
#endif
        }
        PT_MARK(11);
    }
    if (merged && only_blk + 1 < pa.n_rblk) {
        // every hand-over store of this unit has been acknowledged at the memory side (the barrier waits for each wave's outstanding
        // stores), then the unit counts itself
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt_mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// per-iteration problem setup: warm start z~, prior mean z~ - u_k on the partition's local index set
// (jobs/RegressionAdmmTrain.java:692-698 ; llf/LibLinear.java:236-245 initSetup)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_setup(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, int nprob, int n_lambda, int n_global,
        const float *__restrict__ z32, const float *__restrict__ u, const double *__restrict__ pinv_l,
        double epsilon, int max_iter)
{
    const int q = blockIdx.y;
    if (q >= nprob) return;
    ProbDev &pr = probs[q];
    const PartDev &pa = parts[pr.part];
    const int n = pa.n_local;
    const float *__restrict__ zl = z32 + (int64_t)pr.lambda_idx * n_global;
    const float *__restrict__ uk = u + ((int64_t)pr.part * n_lambda + pr.lambda_idx) * n_global;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int gj = pa.l2g[j];
        const double zt = (double)zl[gj], uj = (double)uk[gj];
        pr.w[j] = zt;
        pr.w_new[j] = zt;
        pr.m[j] = -1.0 * uj + 1.0 * zt;       // priormean.linearCombine(-1, 1, initvalue)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        pr.pinv = pinv_l[pr.lambda_idx];
        pr.pinv_vec = nullptr;
        const int mn = pa.pos < pa.neg ? pa.pos : pa.neg;
        pr.eps = epsilon * (double)mn / (double)pa.l;     // llf/LibLinear.java:310-311
        pr.max_iter = max_iter;
        pr.phase = PH_EVAL0;
        pr.iter = 1;
        pr.cg_iter = 0;
        pr.newton = pr.accepted = pr.cg_total = pr.ticks = 0;
        pr.status = ST_OK;
        pr.f = pr.delta = pr.gnorm = pr.gnorm1 = pr.rTr = pr.cgtol = pr.prered = pr.gs = 0.0;
        pr.rsel = 0; pr.gsq = pr.snorm = 0.0;
    }
}

// Mean-model warm start (initialize.boost.rate): the per-key solve of jobs/RegressionNaiveTrain.java:380
// `liblinear.train(dataset, null, null, priorVarMap, prior.mean, 1/lambda, option)`: w0 = 0, prior mean constant,
// prior precision per coordinate (default 1/(1/lambda); lambda.map / intercept overrides in pinv_ovr, NaN = none).
__global__ void __launch_bounds__(256)
k_setup_naive(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, int nprob, int max_nlocal,
              const double *__restrict__ pinv_l, const double *__restrict__ pinv_ovr, double *__restrict__ pinv_buf,
              double prior_mean, double epsilon, int max_iter)
{
    const int q = blockIdx.y;
    if (q >= nprob) return;
    ProbDev &pr = probs[q];
    const PartDev &pa = parts[pr.part];
    const int n = pa.n_local;
    double *__restrict__ pv = pinv_buf + (int64_t)q * max_nlocal;
    const double dflt = pinv_l[pr.lambda_idx];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const double o = pinv_ovr[pa.l2g[j]];
        pv[j] = (o != o) ? dflt : o;
        pr.w[j] = 0.0;
        pr.w_new[j] = 0.0;
        pr.m[j] = prior_mean;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        pr.pinv = 0.0;
        pr.pinv_vec = pv;
        const int mn = pa.pos < pa.neg ? pa.pos : pa.neg;
        pr.eps = epsilon * (double)mn / (double)pa.l;     // llf/LibLinear.java:310-311
        pr.max_iter = max_iter;
        pr.phase = PH_EVAL0;
        pr.iter = 1;
        pr.cg_iter = 0;
        pr.newton = pr.accepted = pr.cg_total = pr.ticks = 0;
        pr.status = ST_OK;
        pr.f = pr.delta = pr.gnorm = pr.gnorm1 = pr.rTr = pr.cgtol = pr.prered = pr.gs = 0.0;
        pr.rsel = 0; pr.gsq = pr.snorm = 0.0;
    }
}

// models/part files of the naive job: every feature of the partition's dataset, float32 (models/LinearModel.java:697-720);
// B was zeroed first, so features absent from the partition add nothing to the mean (LinearModel.java:181-201).
__global__ void __launch_bounds__(256)
k_outputs_naive(const PartDev *__restrict__ parts, const ProbDev *__restrict__ probs, int nprob, int n_lambda,
                int n_global, float *__restrict__ B)
{
    const int q = blockIdx.y;
    if (q >= nprob) return;
    const ProbDev &pr = probs[q];
    const PartDev &pa = parts[pr.part];
    const int64_t base = ((int64_t)pr.part * n_lambda + pr.lambda_idx) * n_global;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < pa.n_local; j += gridDim.x * 256) B[base + pa.l2g[j]] = (float)pr.w[j];
}

// ------------------------------------------------------------------------------------------------
// TRON / CG control flow: one workgroup per problem, one call per tick (bw/Tron.java:30-179).
// Elementwise updates mirror the Java statement by statement with contraction OFF (Java never
// fuses a*b+c); only the reductions (dot, norm) use a tree instead of a sequential sum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pinv_at(const ProbDev &pr, int j)
{
    return pr.pinv_vec ? pr.pinv_vec[j] : pr.pinv;
}

// out = X' c assembled from the pass partials in a fixed order (dense: per-block slices; CSR: column
// segments in row order + the intercept's per-block coefficient sums).
// Deterministic sum of a short global array over the block (strided per-thread partials, then the block tree).
__device__ __forceinline__ double block_sum_array(const double *__restrict__ a, int cnt, double *scratch)
{
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) v[0] += a[i];
    block_allreduce_sum<1>(v, scratch);
    return v[0];
}

__device__ __forceinline__ void assemble_out(const PartDev &pa, const ProbDev &pr, double *__restrict__ out,
                                             double *scratch, double *stage /* LDS, blockDim.x doubles */)
{
    const int n = pa.n_local, nf = pa.n_feat;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (pa.dense) {
        // thread = (slice group g, column c): 256 columns x nt/256 groups; each group walks its slices with
        // several loads in flight (a single thread walking all P slices is latency-bound: 85 us -> ~10 us).
        // terms = PAIRS of row units (k_xpass_dense): stored as such by two-unit workgroups, formed here from one-unit partials
        const int P = (pa.n_units + 1) / 2;
        const bool pairs = pa.units_per_wg == 2;
        const int nun = pa.n_units;
        const int CW = nt < 256 ? nt : 256, NG = nt / CW;
        const int c = tid % CW, g = tid / CW;
        const double *__restrict__ parts = pr.parts;
        for (int base = 0; base < n; base += CW) {
            const int j = base + c;
            double a = 0.0;
            if (j < n) {
                // 8 slice loads in flight: few-problem shapes cut the rows into up to ~250 chunks (mlx_finalize)
                if (pairs) {
#pragma unroll 8
                    for (int p = g; p < P; p += NG) a += parts[(int64_t)p * n + j];
                } else {
#pragma unroll 4
                    for (int p = g; p < P; p += NG) {
                        const double t0 = parts[(int64_t)(2 * p) * n + j];
                        a += (2 * p + 1 < nun) ? t0 + parts[(int64_t)(2 * p + 1) * n + j] : t0;
                    }
                }
            }
            stage[g * CW + c] = a;
            __syncthreads();
            if (g == 0 && j < n) {
                double t = stage[c];
                for (int k = 1; k < NG; k++) t += stage[k * CW + c];
                out[j] = t;
            }
            __syncthreads();
        }
    } else {
        const double *__restrict__ parts = pr.parts;
        const int32_t *__restrict__ cptr = pa.col_ptr;
        for (int j = tid; j < nf; j += nt) {
            double a = 0.0;
            const int i0 = cptr[j], i1 = cptr[j + 1];
            for (int it = i0; it < i1; it++) a += parts[it];
            out[j] = a;
        }
        const double cs = block_sum_array(pr.csump, pa.n_rowparts, scratch);
        if (tid == 0) out[nf] = cs;
    }
}

// c0 = X' t0 (data part of grad(0)) from an EVAL pass at w = 0; run once per partition at finalize.
__global__ void __launch_bounds__(256)
k_collect_c0(const PartDev *__restrict__ parts, const ProbDev *__restrict__ probs, const int *__restrict__ qlist,
             double *const *__restrict__ c0_ptrs)
{
    __shared__ double scratch[64];
    __shared__ double stage[256];
    const ProbDev &pr = probs[qlist[blockIdx.x]];
    assemble_out(parts[pr.part], pr, c0_ptrs[blockIdx.x], scratch, stage);
}

// ---- grid-rounded dots: what Tron.dot's SEQUENTIAL loop does to the small terms, without its dependency chain -------------------
// `for (i) p += a[i]*b[i]` (bw/Tron.java:204-213) adds every term to a running sum that is already large after the first few dozen
// (hot) columns: the term's bits below the running sum's ulp are rounded away on the spot. That rounding is a property of (term,
// binade of the running sum), hardly of the order of the terms, so every row / feature order the reference may see reproduces almost
// the same sum -- while a tree (or an exact, compensated sum) keeps those bits and lands 50-100 ulp away: enough to leave the
// reference's own family of TRON trajectories on one-hot data. Measured (tools/sum_order_experiment.py, profiles/r4_notes.md; solves
// of full-size configs[2] partitions whose TRON counters equal the oracle's): oracle on permuted rows 253-269 of 384, this library
// with tree dots 200, with compensated dots fewer still, with the dots below 242.
// The grid comes from the HEAD sum: the terms of the problem's first STEP_HEAD columns (the hottest: library ids are frequency-
// sorted; STEP_HEAD = 64 of them). For r.r every workgroup of phase B adds them itself (two loads per column, L2 hits); for d.Hd / g.g the head needs the
// column sums of the hottest columns -- O(100) slots each -- so one small launch per tick (k_step_head, one workgroup per problem)
// leaves it in the descriptor for phase A. u = the ulp of the head sum's binade. Terms of the later columns are rounded to a multiple of u before they are added: the magic-constant form
// (x + 1.5 * 2^52 u) - 1.5 * 2^52 u, exactly what the FPU does to x when it is added to a sum of that size. Sums of multiples of u
// are exact, so the trees that follow add no rounding of their own. (The grid of the running sum after the first 64 / 256 / 2048 terms,
// or of the true prefix at every 2048-column chunk, all follow the oracle equally well on CPU models of this arithmetic; the head
// sum needs no exchange between workgroups. A first version published chunk sums through agent-scope granules and looked back over
// them: same parity, 17 % of the sparse leg's throughput; a second recomputed the d.Hd head in every workgroup of phase A: 9 %.)
#ifndef STEP_HEAD
#define STEP_HEAD 64           // head columns of phase B's r.r (computed in the kernel) and, by default, of phase A's d.Hd (k_step_head).
#endif                         // (64 against 256: CPU model 132 against 120-130 of 192 solves followed, GPU 256 against 250 of 384; A/B: -DSTEP_HEAD=256)
#ifndef STEP_HEAD_A
#define STEP_HEAD_A STEP_HEAD
#endif
#define STEP_HEAD_LANES (1024 / STEP_HEAD_A)               // lanes of k_step_head per head column
__device__ __forceinline__ double grid_of_sum(double h)
{
    const double ah = fabs(h);
    if (!(ah > 0.0) || !(ah < 1e300)) return 0.0;
    int e;
    (void)frexp(ah, &e);                                       // ah in [2^(e-1), 2^e): ulp = 2^(e-53)
    return ldexp(1.0, e - 53);
}
// x as the FPU leaves it when it is added to a running sum whose ulp is u (round to nearest multiple of u, ties to even); u = 0: x
__device__ __forceinline__ double round_to_grid(double x, double u)
{
#pragma clang fp contract(off)
    const double magic = 6755399441055744.0 * u;              // 1.5 * 2^52 * u
    return (fabs(x) < 1125899906842624.0 * u) ? (x + magic) - magic : x;      // (|x| >= 2^50 u, or u = 0: x as it is)
}

// The same dot for the one-workgroup solver (k_solve_small; its vectors are whole in front of the team): the update loop in front adds
// the terms of the first STEP_HEAD columns only (`head`), this adds every later term rounded to that sum's grid -- k_step_head + phase A /
// phase B of the tick kernels in one place. (Multiples of u add exactly: head + tail is one rounding. Every thread reads the elements
// it wrote itself in the loop before: j = tid mod nt in both.)
template <typename T, typename VP>
__device__ __forceinline__ double team_grid_dot_tail(VP x, VP y, int n, double head, double *scratch)
{
#pragma clang fp contract(off)
    if (n <= STEP_HEAD) return head;
    const double u = grid_of_sum(head);
    double a[1] = {0.0};
    for (int j = STEP_HEAD + T::tid(); j < n; j += T::nt()) a[0] += round_to_grid(x[j] * y[j], u);
    T::template allreduce<1>(a, scratch);
    return head + a[0];
}

// ---- order-faithful verification mode (MLX_FAITHFUL=1): every n- or l-long reduction is done by
// ONE thread in the reference's sequential order, with the reference's formulas (bw/Tron.java:204-252). Slow by design.
// The operands are staged through LDS 1024 at a time by all threads (coalesced loads, products formed in parallel -- an
// elementwise operation, identical to the reference's), and thread 0 folds each chunk in index order.
__device__ __forceinline__ double seq_bcast(double v, double *scratch)
{
    __syncthreads();
    if (threadIdx.x == 0) scratch[0] = v;
    __syncthreads();
    const double out = scratch[0];
    __syncthreads();
    return out;
}
template <typename F>
__device__ __forceinline__ double seq_fold_sum(double init, int n, double *scratch, double *stage, F term)
{
#pragma clang fp contract(off)
    double a = init;
    for (int base = 0; base < n; base += 1024) {
        const int j = base + (int)threadIdx.x;
        __syncthreads();
        if (j < n && threadIdx.x < 1024) stage[threadIdx.x] = term(j);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = min(1024, n - base);
            for (int i = 0; i < cnt; i++) a += stage[i];
        }
    }
    return seq_bcast(a, scratch);
}
__device__ __forceinline__ double seq_dot(const double *x, const double *y, int n, double *scratch, double *stage)   // Tron.dot :204-213
{
    return seq_fold_sum(0.0, n, scratch, stage, [=](int j) { return x[j] * y[j]; });
}
__device__ __forceinline__ double seq_sum(const double *x, int n, double *scratch, double *stage)
{
    return seq_fold_sum(0.0, n, scratch, stage, [=](int j) { return x[j]; });
}
__device__ __forceinline__ double seq_norm(const double *v, int n, double *scratch, double *stage)                   // Tron.euclideanNorm :220-252
{
#pragma clang fp contract(off)
    if (n < 1) return 0.0;
    if (n == 1) return seq_bcast(fabs(v[0]), scratch);
    double scale = 0, sum = 1;
    for (int base = 0; base < n; base += 1024) {
        const int j = base + (int)threadIdx.x;
        __syncthreads();
        if (j < n && threadIdx.x < 1024) stage[threadIdx.x] = v[j];
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = min(1024, n - base);
            for (int i = 0; i < cnt; i++) {
                const double vi = stage[i];
                if (vi != 0) {
                    const double a = fabs(vi);
                    if (scale < a) { const double t = scale / a; sum = 1 + sum * (t * t); scale = a; }
                    else { const double t = a / scale; sum += t * t; }
                }
            }
        }
    }
    return seq_bcast(scale * sqrt(sum), scratch);
}

template <bool SEQ, typename T = BlockTeam, typename VP = double *>
__device__ __forceinline__ void tron_step_body(const PartDev &pa, ProbDev &pr, double *scratch, double *stage,
                                               int *__restrict__ done_counter, bool grid_dots = false)
{
#pragma clang fp contract(off)
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const int n = pa.n_local;
    static_assert(!SEQ || std::is_same<T, BlockTeam>::value, "the order-faithful mode runs on the whole workgroup");
    const int tid = T::tid(), nt = T::nt();
    static_assert(!SEQ || std::is_same<VP, double *>::value, "the order-faithful mode keeps its vectors in global memory");
    const VP w = (VP)pr.w, w_new = (VP)pr.w_new, g = (VP)pr.g, s = (VP)pr.s, r = (VP)pr.r, d = (VP)pr.d, Hd = (VP)pr.Hd, m = (VP)pr.m;

    // X'c of this tick: dense partitions assemble their per-block partial vectors into Hd[] first; CSR partitions read
    // their column-segment sums inline in the first update loop (one write + one read of Hd[] less per tick).
    const bool inl = !pa.dense;
    const int nf = pa.n_feat;
    double csum_icpt = 0.0;
    if (inl) csum_icpt = SEQ ? seq_sum(pr.coef, pa.l, scratch, stage) : team_sum_array<T>((VP)pr.csump, pa.n_rowparts, scratch);   // SEQ: XTv's row order
    else if constexpr (std::is_same<VP, double *>::value) assemble_out(pa, pr, Hd, scratch, stage);
    T::sync();
    const VP segsum = (VP)pr.parts;
    const int32_t *__restrict__ cptr = pa.col_ptr;
    // X'c for XB strided columns at once (j = jb + u*nt): all slot ranges are fetched first, then all first slots, then the
    // (rare) further ones -- two dependent latencies per batch instead of per column.
    constexpr int XB = std::is_same<T, WaveTeam>::value ? 4 : 8;      // (one wave: n <= 256 is one batch of 4 without idle slots)
    auto xtc_batch = [&](int jb, double (&xa)[XB]) {
        if (!inl) {
#pragma unroll
            for (int u = 0; u < XB; u++) { const int j = jb + u * nt; xa[u] = j < n ? Hd[j] : 0.0; }
            return;
        }
        int i0[XB], i1[XB];
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int j = jb + u * nt;
            const int jc = min(j, nf - 1);
            const bool col = j < nf && nf > 0;
            i0[u] = col ? cptr[jc] : 0; i1[u] = col ? cptr[jc + 1] : 0;
        }
        double f0[XB];
#pragma unroll
        for (int u = 0; u < XB; u++) f0[u] = i1[u] > i0[u] ? segsum[i0[u]] : 0.0;
        int more = 0;
#pragma unroll
        for (int u = 0; u < XB; u++) {
            xa[u] = 0.0;                                       // slot order = (block, segment) order
            if (i1[u] > i0[u]) xa[u] += f0[u];
            more = max(more, i1[u] - i0[u]);
        }
        for (int k = 1; k < more; k++) {                       // (columns with several slots: rare; ONE loop for the batch)
#pragma unroll
            for (int u = 0; u < XB; u++) if (i0[u] + k < i1[u]) xa[u] += segsum[i0[u] + k];
        }
#pragma unroll
        for (int u = 0; u < XB; u++) if (jb + u * nt == nf) xa[u] = csum_icpt;
    };
    // The elementwise loops below run in batches of SB strided elements: all loads of a batch first, then the arithmetic and the
    // stores in element order -- the vectors may alias as far as the compiler knows, so a plain loop is one dependent
    // load -> store chain per element (what a single wave of the small solver spends its step on). Same operations, same order.
    constexpr int SB = 4;
    const double *const pvec = pr.pinv_vec;
    const double pscal = pr.pinv;

    const double rTr0 = pr.rTr, delta0 = pr.delta, cgtol0 = pr.cgtol, eps0 = pr.eps, gnorm1_0 = pr.gnorm1;
    double gnorm_cur = pr.gnorm;
#ifdef MLX_SMALL_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) g_small_prof_t = clock64();
#endif
    bool start_trcg = false, finished = false;

    if (phase == PH_CG) {
        // ---- one CG step (bw/Tron.java:145-175)
        double a1[1] = {0.0};
        for (int jb = tid; jb < n; jb += XB * nt) {
            double xa[XB];
            xtc_batch(jb, xa);
            double dv[XB], pv[XB];
#pragma unroll
            for (int u = 0; u < XB; u++) {
                const int j = jb + u * nt;
                dv[u] = j < n ? d[j] : 0.0;
                pv[u] = j < n ? (pvec ? pvec[j] : pscal) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < XB; u++) {
                const int j = jb + u * nt;
                if (j < n) {
                    const double hd = dv[u] * pv[u] + xa[u];             // Hs[i] = (s[i]*priorVar_inv[i] + Hs[i]) * 1
                    Hd[j] = hd;
                    if (!grid_dots || j < STEP_HEAD) a1[0] += dv[u] * hd;
                }
            }
        }
        SPROF2(0);      // CG: Hd loop
        T::template allreduce<1>(a1, scratch);
        SPROF2(1);      // its reduction
        if (SEQ) a1[0] = seq_dot(pr.d, pr.Hd, n, scratch, stage);
        else if (grid_dots) a1[0] = team_grid_dot_tail<T>(d, Hd, n, a1[0], scratch);       // the fast contract's d.Hd (grid_of_sum)
        double alpha = rTr0 / a1[0];
        double ss1[1] = {0.0};
        for (int jb = tid; jb < n; jb += SB * nt) {
            double sv[SB], dv[SB];
#pragma unroll
            for (int u = 0; u < SB; u++) { const int j = jb + u * nt; sv[u] = j < n ? s[j] : 0.0; dv[u] = j < n ? d[j] : 0.0; }
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const int j = jb + u * nt;
                if (j < n) {
                    const double sj = sv[u] + alpha * dv[u];        // daxpy(alpha, d, s)
                    s[j] = sj;
                    ss1[0] += sj * sj;
                }
            }
        }
        T::template allreduce<1>(ss1, scratch);
        const double snorm = SEQ ? seq_norm(pr.s, n, scratch, stage) : team_norm_from_sumsq<T>(ss1[0], s, n, scratch);
        SPROF2(2);      // s update + norm
        bool end_cg = false;
        if (snorm > delta0) {
            // cg reaches trust region boundary (:150-168)
            alpha = -alpha;
            double a3[3] = {0.0, 0.0, 0.0};
            _Pragma("unroll 8") for (int j = tid; j < n; j += nt) {
                const double sj = s[j] + alpha * d[j];
                s[j] = sj;
                a3[0] += sj * d[j];
                a3[1] += sj * sj;
                a3[2] += d[j] * d[j];
            }
            T::template allreduce<3>(a3, scratch);
            if (SEQ) { a3[0] = seq_dot(pr.s, pr.d, n, scratch, stage); a3[1] = seq_dot(pr.s, pr.s, n, scratch, stage); a3[2] = seq_dot(pr.d, pr.d, n, scratch, stage); }
            const double std_ = a3[0], sts = a3[1], dtd = a3[2];
            const double dsq = delta0 * delta0;
            const double rad = sqrt(std_ * std_ + dtd * (dsq - sts));
            if (std_ >= 0) alpha = (dsq - sts) / (std_ + rad);
            else alpha = (rad - std_) / dtd;
            const double nalpha = -alpha;
            _Pragma("unroll 8") for (int j = tid; j < n; j += nt) {
                s[j] += alpha * d[j];
                r[j] += nalpha * Hd[j];
            }
            end_cg = true;
        } else {
            alpha = -alpha;
            double a2[1] = {0.0};
            for (int jb = tid; jb < n; jb += SB * nt) {
                double rv[SB], hv[SB];
#pragma unroll
                for (int u = 0; u < SB; u++) { const int j = jb + u * nt; rv[u] = j < n ? r[j] : 0.0; hv[u] = j < n ? Hd[j] : 0.0; }
#pragma unroll
                for (int u = 0; u < SB; u++) {
                    const int j = jb + u * nt;
                    if (j < n) {
                        const double rj = rv[u] + alpha * hv[u];
                        r[j] = rj;
                        if (!grid_dots || j < STEP_HEAD) a2[0] += rj * rj;
                    }
                }
            }
            T::template allreduce<1>(a2, scratch);
            if (SEQ) a2[0] = seq_dot(pr.r, pr.r, n, scratch, stage);
            else if (grid_dots) a2[0] = team_grid_dot_tail<T>(r, r, n, a2[0], scratch);     // ... and its r'.r'
            const double rnew = a2[0];
            const double beta = rnew / rTr0;
            for (int jb = tid; jb < n; jb += SB * nt) {
                double dv[SB], rv[SB];
#pragma unroll
                for (int u = 0; u < SB; u++) { const int j = jb + u * nt; dv[u] = j < n ? d[j] : 0.0; rv[u] = j < n ? r[j] : 0.0; }
#pragma unroll
                for (int u = 0; u < SB; u++) {
                    const int j = jb + u * nt;
                    if (j < n) {
                        double dj = dv[u];
                        if (beta != 1.0) dj = dj * beta;           // scale(beta, d)
                        d[j] = dj + 1.0 * rv[u];                   // daxpy(one, r, d)
                    }
                }
            }
            const double rnorm = SEQ ? seq_norm(pr.r, n, scratch, stage) : team_norm_from_sumsq<T>(rnew, r, n, scratch);
            if (tid == 0) pr.rTr = rnew;
            if (rnorm <= cgtol0) end_cg = true;                  // loop-top test of the next trip (:144)
        }
        T::sync();
        SPROF2(3);      // r, d updates + norm
        if (tid == 0) { pr.cg_iter += 1; pr.ticks += 1; }
        if (end_cg) {
            // back in tron(): w_new = w + s, gs, prered (:69-73)
            double a2[2] = {0.0, 0.0};
            for (int jb = tid; jb < n; jb += SB * nt) {
                double wv[SB], sv[SB], gv[SB], rv[SB];
#pragma unroll
                for (int u = 0; u < SB; u++) {
                    const int j = jb + u * nt;
                    wv[u] = j < n ? w[j] : 0.0; sv[u] = j < n ? s[j] : 0.0; gv[u] = j < n ? g[j] : 0.0; rv[u] = j < n ? r[j] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < SB; u++) {
                    const int j = jb + u * nt;
                    if (j < n) {
                        w_new[j] = wv[u] + 1.0 * sv[u];
                        a2[0] += gv[u] * sv[u];
                        a2[1] += sv[u] * rv[u];
                    }
                }
            }
            T::template allreduce<2>(a2, scratch);
            if (SEQ) { a2[0] = seq_dot(pr.g, pr.s, n, scratch, stage); a2[1] = seq_dot(pr.s, pr.r, n, scratch, stage); }
            if (tid == 0) {
                pr.gs = a2[0];
                pr.prered = -0.5 * (a2[0] - a2[1]);
                pr.newton += 1;
                pr.cg_total += pr.cg_iter;
                pr.phase = PH_EVAL;
            }
        }
        SPROF2(4);      // end of trcg
#ifdef MLX_SMALL_PROFILE
        if (blockIdx.x == 0 && threadIdx.x == 0) g_small_prof2[14] += 1;
#endif
        return;
    }

    // ---- PH_EVAL0 / PH_EVAL: objective and gradient at w_new
    double a1[1] = {0.0};
    for (int jb = tid; jb < n; jb += XB * nt) {
        double xa[XB];
        xtc_batch(jb, xa);
        double wv[XB], mv[XB], pv[XB];
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int j = jb + u * nt;
            wv[u] = j < n ? w_new[j] : 0.0; mv[u] = j < n ? m[j] : 0.0;
            pv[u] = j < n ? (pvec ? pvec[j] : pscal) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int j = jb + u * nt;
            if (j < n) {
                const double t = wv[u] - mv[u];
                const double pj = pv[u];
                a1[0] += t * t * pj;                                        // fun :187-188
                Hd[j] = t * pj + xa[u];                                     // grad :224 (multiplier 1)
            }
        }
    }
    T::template allreduce<1>(a1, scratch);
    const double loss = SEQ ? seq_sum(pr.rowtmp, pa.l, scratch, stage)
                            : (pa.dense ? team_sum_pairs<T>((VP)pr.lossp, pa.n_units, pa.units_per_wg == 2, scratch)
                                        : team_sum_array<T>((VP)pr.lossp, pa.n_rowparts, scratch));
    double fnew = 2.0 * loss;
    if (SEQ) {
        // fun :184-189 adds the prior terms to the running f one by one
        const ProbDev *prp = &pr;
        fnew = seq_fold_sum(fnew, n, scratch, stage, [=](int j) { const double temp = w_new[j] - m[j]; return temp * temp * pinv_at(*prp, j); });
    } else {
        fnew = fnew + a1[0];
    }
    fnew = fnew / 2.0;
    T::sync();
    SPROF2(5);          // EVAL: objective + gradient loop, reductions
    const double *__restrict__ c0 = SEQ ? pr.c0f : pa.c0;

    if (phase == PH_EVAL0) {
        // Tron prologue (:47-62): gnorm1 = ||grad(0)||, f, g, delta at the warm start
        for (int jb = tid; jb < n; jb += SB * nt) {
            double hv[SB], mv[SB], pv[SB], cv[SB];
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const int j = jb + u * nt;
                hv[u] = j < n ? Hd[j] : 0.0; mv[u] = j < n ? m[j] : 0.0; pv[u] = j < n ? (pvec ? pvec[j] : pscal) : 0.0; cv[u] = j < n ? c0[j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const int j = jb + u * nt;
                if (j < n) {
                    g[j] = hv[u];
                    s[j] = (0.0 - mv[u]) * pv[u] + cv[u];           // grad(0) staged in s[]
                }
            }
        }
        T::sync();
        const double gnorm1 = SEQ ? seq_norm(pr.s, n, scratch, stage) : team_norm<T>(s, n, scratch);
        const double gnorm = SEQ ? seq_norm(pr.g, n, scratch, stage) : team_norm<T>(g, n, scratch);
        if (tid == 0) {
            pr.f = fnew; pr.gnorm1 = gnorm1; pr.gnorm = gnorm; pr.delta = gnorm;
            pr.dsel ^= 1; pr.ticks += 1;
        }
        gnorm_cur = gnorm;
        if (gnorm <= eps0 * gnorm1) finished = true;               // search = 0
        else start_trcg = true;
        if (!(fnew == fnew) || !(gnorm == gnorm)) { finished = true; start_trcg = false; if (tid == 0) pr.status = ST_NAN; }
    } else {
        const double eta0 = 1e-4, eta1 = 0.25, eta2 = 0.75;
        const double sigma1 = 0.25, sigma2 = 0.5, sigma3 = 4;
        double f = pr.f, delta = delta0, gnorm = gnorm_cur;
        const double gs = pr.gs, prered = pr.prered;
        const double actred = f - fnew;
        const double snorm = SEQ ? seq_norm(pr.s, n, scratch, stage) : team_norm<T>(s, n, scratch);
        if (pr.iter == 1) delta = fmin(delta, snorm);
        double alpha;
        if (fnew - f - gs <= 0) alpha = sigma3;
        else alpha = fmax(sigma1, -0.5 * (gs / (fnew - f - gs)));
        if (actred < eta0 * prered) delta = fmin(fmax(alpha, sigma1) * snorm, sigma2 * delta);
        else if (actred < eta1 * prered) delta = fmax(sigma1 * delta, fmin(alpha * snorm, sigma2 * delta));
        else if (actred < eta2 * prered) delta = fmax(sigma1 * delta, fmin(alpha * snorm, sigma3 * delta));
        else delta = fmax(delta, fmin(alpha * snorm, sigma3 * delta));
        int iter = pr.iter;
        bool brk = false;
        const bool accept = actred > eta0 * prered;
        if (accept) {
            iter++;
            for (int jb = tid; jb < n; jb += SB * nt) {
                double wv[SB], hv[SB];
#pragma unroll
                for (int u = 0; u < SB; u++) { const int j = jb + u * nt; wv[u] = j < n ? w_new[j] : 0.0; hv[u] = j < n ? Hd[j] : 0.0; }
#pragma unroll
                for (int u = 0; u < SB; u++) { const int j = jb + u * nt; if (j < n) { w[j] = wv[u]; g[j] = hv[u]; } }
            }
            f = fnew;
            T::sync();
            gnorm = SEQ ? seq_norm(pr.g, n, scratch, stage) : team_norm<T>(g, n, scratch);
            if (gnorm <= eps0 * gnorm1_0) brk = true;
        }
        if (!brk) {
            if (f < -1.0e+32) brk = true;
            else if (fabs(actred) <= 0 && prered <= 0) brk = true;
            else if (fabs(actred) <= 1.0e-12 * fabs(f) && fabs(prered) <= 1.0e-12 * fabs(f)) brk = true;
        }
        T::sync();
        if (tid == 0) {
            pr.f = f; pr.delta = delta; pr.gnorm = gnorm; pr.iter = iter; pr.ticks += 1;
            if (accept) { pr.accepted += 1; pr.dsel ^= 1; }
        }
        if (!(fnew == fnew) || !(gnorm == gnorm)) { brk = true; if (tid == 0) pr.status = ST_NAN; }
        gnorm_cur = gnorm;
        if (brk || iter > pr.max_iter) finished = true;            // while (iter <= max_iter && search)
        else start_trcg = true;
    }

    if (start_trcg) {
        // trcg prologue (:133-141): s = 0, r = -g, d = r, cgtol = 0.1||g||, rTr = r.r
        double a2[1] = {0.0};
        for (int jb = tid; jb < n; jb += SB * nt) {
            double gv[SB];
#pragma unroll
            for (int u = 0; u < SB; u++) { const int j = jb + u * nt; gv[u] = j < n ? g[j] : 0.0; }
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const int j = jb + u * nt;
                if (j < n) {
                    const double rj = -gv[u];
                    s[j] = 0.0; r[j] = rj; d[j] = rj;
                    if (!grid_dots || j < STEP_HEAD) a2[0] += rj * rj;
                }
            }
        }
        T::template allreduce<1>(a2, scratch);
        if (SEQ) a2[0] = seq_dot(pr.r, pr.r, n, scratch, stage);
        else if (grid_dots) a2[0] = team_grid_dot_tail<T>(r, r, n, a2[0], scratch);         // ... and the r.r a trcg call starts from
        const double gn = gnorm_cur;      // ||r|| = ||-g|| = ||g||
        if (tid == 0) {
            pr.rTr = a2[0];
            pr.cgtol = 0.1 * gn;
            pr.cg_iter = 0;
        }
        if (gn <= 0.1 * gn) {
            // CG loop exits at once (:144) with s = 0: evaluate the (null) step like the reference does
            _Pragma("unroll 8") for (int j = tid; j < n; j += nt) w_new[j] = w[j] + 1.0 * 0.0;
            if (tid == 0) { pr.gs = 0.0; pr.prered = -0.5 * (0.0 - 0.0); pr.newton += 1; pr.phase = PH_EVAL; }
        } else if (tid == 0) {
            pr.phase = PH_CG;
        }
    }
    if (finished && tid == 0) {
        pr.phase = PH_DONE;
        atomicAdd(done_counter, 1);
    }
    SPROF2(6);          // EVAL: the rest (norms, trust region update, trcg prologue)
#ifdef MLX_SMALL_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) g_small_prof2[15] += 1;
#endif
}

__global__ void __launch_bounds__(1024)
k_tron_step(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq,
            int *__restrict__ done_counter)
{
    __shared__ double scratch[64];
    __shared__ double stage[1024];
    if ((int)blockIdx.x >= nq) return;
    ProbDev &pr = probs[qlist[blockIdx.x]];
    tron_step_body<false>(parts[pr.part], pr, scratch, stage, done_counter);
}

// ------------------------------------------------------------------------------------------------
// Multi-workgroup TRON/CG step of the CSR tick path (bw/Tron.java:30-179, same statements as tron_step_body).
// A tick's step is four launches -- A, B, C, commit -- because a CG step has two global reductions in sequence
// (alpha = rTr / d.Hd, then beta = r'.r' / rTr) and every coordinate update needs the scalar before it:
//     A  Hd = d*pinv + X'c            partial d.Hd                    | EVAL: gradient candidate, sum t^2 pinv, |grad|^2
//     B  s += alpha d ; r' = r - alpha Hd   partial |s|^2, |r'|^2 (+ the boundary sums s.d, s.s, d.d, speculatively)
//                                                                     | EVAL: accept/reject, w/g copies, trcg prologue
//     C  d = beta d + r'  (or the trust-region boundary step)  ;  at the end of trcg: w_new = w + s, partial g.s, s.r
//     commit  one workgroup per problem writes the problem's scalars (phase, f, delta, rTr, counters ...)
// Each problem is cut into column chunks of `ch` columns, one 256-thread workgroup per chunk, so 128 problems of 35 K
// columns are ~2 300 workgroups instead of 128. Reductions cross the KERNEL BOUNDARY only: every workgroup writes its
// partial sums, and every workgroup of the next launch adds all of them in chunk order itself (a few hundred L2 hits) and
// derives the same scalars from them -- fixed order, hence bit-reproducible, and no device-scope fence inside a kernel
// (on this multi-XCD part a fence writes back / invalidates a whole L2: a last-arrival reduction with one fence per
// workgroup ran the three phases at 1.3 TB/s). No scalar of ProbDev is written before the commit launch, so A, B and C
// all see the tick's initial phase.
// Norms are sqrt(sum v^2) of sums gathered inside the update loops (euclideanNorm's scaled form up to the last bits).
// ------------------------------------------------------------------------------------------------
#define STEP_T 256
#ifdef STEP_MINW
#define STEP_LB __launch_bounds__(STEP_T, STEP_MINW)      // A/B builds: minimum waves per SIMD of the streaming phases (tools/ablate_build.sh)
#else
#define STEP_LB __launch_bounds__(STEP_T)
#endif
// The step's n-vector streams are touched once per launch and not again before ~1 GB of other traffic has gone by: streaming
// (non-temporal) loads and stores keep them from displacing the gathered vectors and index packs in L2 (config #3: step 316 -> 283 us per
// tick, 5.06 k -> 5.29 k solves/s). NOT for what the next launch reads from L2: the column pass's slot stores and phase A's slot /
// pointer loads as streaming accesses cost 13 % of the column pass and 10 % of phase A. -DMLX_STEP_NO_NT: A/B build.
#ifdef MLX_STEP_NO_NT
#define SLD(p) gld(p)
#define SST(p, v) gst((p), (v))
#else
#define SLD(p) gld_nt(p)
#define SST(p, v) gst_nt((p), (v))
#endif
#ifndef STEP_XB
#define STEP_XB 4      // columns per thread and round (independent loads in flight)
#endif

// chunk-ordered totals of the previous launch's partial sums px[nwg][STEP_NP] -> tot[NP] (LDS); the loads run in
// parallel (one partial per thread), the additions sequentially per column
template <int NP>
__device__ __forceinline__ void step_gather(const double *__restrict__ px, int nwg, double *tot /* LDS [NP] */,
                                            double *stage /* LDS [STEP_T] */)
{
    static_assert(NP <= STEP_NP, "partials per workgroup");
    const int tid = threadIdx.x;
    constexpr int WPR = STEP_T / STEP_NP;          // workgroups' partials per round
    double a = 0.0;
    for (int w0 = 0; w0 < nwg; w0 += WPR) {
        const int w = w0 + tid / STEP_NP, k = tid % STEP_NP;
        __syncthreads();
        stage[tid] = (w < nwg && k < NP) ? px[w * STEP_NP + k] : 0.0;
        __syncthreads();
        if (tid < NP) {
            const int cnt = min(WPR, nwg - w0);
            for (int i = 0; i < cnt; i++) a += stage[i * STEP_NP + tid];
        }
    }
    if (tid < NP) tot[tid] = a;
    __syncthreads();
}


// sqrt of a sum of squares gathered in an update loop; a sum that overflowed (or is NaN) is reported as NaN so that the
// caller stops the solve with ST_NAN instead of comparing against inf
__device__ __forceinline__ double norm_of_sumsq(double ss) { return (ss < 1e300) ? sqrt(ss) : (0.0 / 0.0); }

struct StepGeom { int n, nf, j0, j1, wg, nwg; };

__device__ __forceinline__ bool step_geom(const PartDev &pa, int ch, StepGeom &g)
{
    g.n = pa.n_local; g.nf = pa.n_feat;
    g.wg = blockIdx.x;
    g.j0 = g.wg * ch;
    g.nwg = (g.n + ch - 1) / ch;
    if (g.j0 >= g.n) return false;
    g.j1 = min(g.n, g.j0 + ch);
    return true;
}

// ---- head of phase A's grid-rounded dot: one workgroup per problem, STEP_HEAD_LANES threads per head column (grid_of_sum) ---------
// (the hottest columns have O(100) column-sum slots each: one thread per column spent 16 us per launch on the longest of them)
__global__ void __launch_bounds__(1024)
k_step_head(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist)
{
#pragma clang fp contract(off)
    __shared__ double scratch[64];
    ProbDev &pr = probs[qlist[blockIdx.x]];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    constexpr int LN = STEP_HEAD_LANES;
    const int n = pa.n_local, nf = pa.n_feat, j = threadIdx.x / LN, q = threadIdx.x % LN;
    const bool cg = (phase == PH_CG);
    double csum_icpt = 0.0;
    if (nf < STEP_HEAD_A) csum_icpt = block_sum_array(pr.csump, pa.n_rowparts, scratch);     // (uniform: a problem of few columns)
    const double *__restrict__ v = cg ? pr.d : pr.w_new;
    double ht[1] = {0.0};
    double xa = 0.0;
    if (j < min(STEP_HEAD_A, nf)) {
        const int i0 = gld(pa.col_ptr + j), i1 = gld(pa.col_ptr + j + 1);
#pragma unroll 4
        for (int it = i0 + q; it < i1; it += LN) xa += gld(pr.parts + it);
    }
    xa = group_allreduce_sum<LN>(xa);                          // (all lanes take part; lanes without a column hold 0)
    if (q == 0 && j < min(STEP_HEAD_A, n)) {
        if (j == nf) xa = csum_icpt;
        const double pjv = pr.pinv_vec ? gld(pr.pinv_vec + j) : pr.pinv;
        const double vj = gld(v + j);
        if (cg) { const double hd = vj * pjv + xa; ht[0] = vj * hd; }
        else { const double t = vj - gld(pr.m + j); const double hd = t * pjv + xa; ht[0] = hd * hd; }
    }
    block_allreduce_sum<1>(ht, scratch);
    if (threadIdx.x == 0) pr.head = ht[0];
}

// ---- phase A ------------------------------------------------------------------------------------
template <bool EMU>
__global__ void STEP_LB
k_step_a(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int ch)
{
#pragma clang fp contract(off)
    __shared__ double scratch[64];
    ProbDev &pr = probs[qlist[blockIdx.y]];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    StepGeom G;
    if (!step_geom(pa, ch, G)) return;
    const int tid = threadIdx.x, n = G.n, nf = G.nf;
    const bool cg = (phase == PH_CG);
    // the intercept's column sum (the chunk that holds column nf) and the loss (chunk 0) come from the row pass's partials
    double csum_icpt = 0.0, loss = 0.0;
    if (G.j1 == n) csum_icpt = block_sum_array(pr.csump, pa.n_rowparts, scratch);
    if (G.wg == 0 && !cg) loss = block_sum_array(pr.lossp, pa.n_rowparts, scratch);
    const double *__restrict__ segsum = pr.parts;
    const int32_t *__restrict__ cptr = pa.col_ptr;
    const double *__restrict__ v = cg ? pr.d : pr.w_new;
    const double *__restrict__ m = pr.m;
    const double *__restrict__ c0 = pa.c0;
    const double *__restrict__ pvec = pr.pinv_vec;
    const double pscal = pr.pinv;
    double *__restrict__ Hd = pr.Hd;
    double acc[3] = {0.0, 0.0, 0.0};
    // EMU: the dot the reference computes sequentially -- d.Hd on a CG tick, g.g (= the r.r a trcg call starts from) on an EVAL tick --
    // as a grid-rounded sum: u from the head sum k_step_head left in the descriptor just before this launch
    const double ugrid = EMU ? grid_of_sum(pr.head) : 0.0;
    for (int jb = G.j0 + tid; jb < G.j1; jb += STEP_XB * STEP_T) {
        int i0[STEP_XB], i1[STEP_XB];
        double vv[STEP_XB], mm[STEP_XB], pj[STEP_XB], cc[STEP_XB];
#pragma unroll
        for (int u = 0; u < STEP_XB; u++) {
            const int j = jb + u * STEP_T;
            const int jc = min(j, G.j1 - 1);
            const bool col = j < nf;
            const int jf = min(jc, max(nf - 1, 0));
            i0[u] = (col && nf > 0) ? gld(cptr + jf) : 0; i1[u] = (col && nf > 0) ? gld(cptr + jf + 1) : 0;
            vv[u] = SLD(v + jc);
            pj[u] = pvec ? gld(pvec + jc) : pscal;
            mm[u] = cg ? 0.0 : gld(m + jc);
            cc[u] = (phase == PH_EVAL0) ? gld(c0 + jc) : 0.0;
        }
        double f0[STEP_XB];
#pragma unroll
        for (int u = 0; u < STEP_XB; u++) f0[u] = i1[u] > i0[u] ? gld(segsum + i0[u]) : 0.0;
#pragma unroll
        for (int u = 0; u < STEP_XB; u++) {
            const int j = jb + u * STEP_T;
            if (j >= G.j1) continue;
            double xa = 0.0;                                   // slot order = (row block, segment) order
            if (i1[u] > i0[u]) { xa += f0[u]; for (int it = i0[u] + 1; it < i1[u]; it++) xa += gld(segsum + it); }
            if (j == nf) xa = csum_icpt;
            if (cg) {
                const double hd = vv[u] * pj[u] + xa;          // Hs[i] = (s[i]*priorVar_inv[i] + Hs[i]) * 1
                SST(Hd + j, hd);
                const double term = vv[u] * hd;
                acc[0] += (EMU && j >= STEP_HEAD_A) ? round_to_grid(term, ugrid) : term;
            } else {
                const double t = vv[u] - mm[u];
                acc[0] += t * t * pj[u];                       // fun :187-188
                const double hd = t * pj[u] + xa;              // grad :224 (multiplier 1)
                gst(Hd + j, hd);
                const double term = hd * hd;
                acc[1] += (EMU && j >= STEP_HEAD_A) ? round_to_grid(term, ugrid) : term;
                if (phase == PH_EVAL0) {
                    const double g0 = (0.0 - mm[u]) * pj[u] + cc[u];      // grad(0)
                    acc[2] += g0 * g0;
                }
            }
        }
    }
    block_allreduce_sum<3>(acc, scratch);
    if (tid == 0) {
        double *__restrict__ px = pr.pA + G.wg * STEP_NP;
        px[0] = acc[0]; px[1] = acc[1]; px[2] = acc[2];
        px[3] = loss;                                          // chunk 0 carries the loss, the others add 0
    }
}

// Everything Tron.tron decides after fun(w_new) is known (bw/Tron.java:75-122) and its prologue (:47-62), from the
// problem's scalars and phase A's totals. Pure function: every workgroup of phase B and the commit evaluate it identically.
struct EvalDecision {
    double f, delta, gnorm, gnorm1, gsq;
    int iter;
    bool copy_w, copy_g, accept, start, nullstep, finished, nan;
};

__device__ __forceinline__ EvalDecision eval_decide(const ProbDev &pr, int phase, const double *totA)
{
#pragma clang fp contract(off)
    EvalDecision D;
    const double tpp = totA[0], hsq = totA[1], g0sq = totA[2], loss = totA[3];
    double fnew = 2.0 * loss;
    fnew = fnew + tpp;
    fnew = fnew / 2.0;
    D.copy_w = D.copy_g = D.accept = D.start = D.nullstep = D.finished = D.nan = false;
    D.gnorm1 = pr.gnorm1; D.iter = pr.iter; D.gsq = pr.gsq;
    if (phase == PH_EVAL0) {
        // Tron prologue (:47-62): gnorm1 = ||grad(0)||, f, g, delta at the warm start
        D.gnorm1 = norm_of_sumsq(g0sq);
        D.gnorm = norm_of_sumsq(hsq);
        D.gsq = hsq;
        D.f = fnew; D.delta = D.gnorm;
        D.copy_g = true;
        if (D.gnorm <= pr.eps * D.gnorm1) D.finished = true;            // search = 0
        else D.start = true;
    } else {
        const double eta0 = 1e-4, eta1 = 0.25, eta2 = 0.75;
        const double sigma1 = 0.25, sigma2 = 0.5, sigma3 = 4;
        double f = pr.f, delta = pr.delta, gnorm = pr.gnorm;
        const double gs = pr.gs, prered = pr.prered;
        const double actred = f - fnew;
        const double snorm = pr.snorm;
        if (pr.iter == 1) delta = fmin(delta, snorm);
        double alpha;
        if (fnew - f - gs <= 0) alpha = sigma3;
        else alpha = fmax(sigma1, -0.5 * (gs / (fnew - f - gs)));
        if (actred < eta0 * prered) delta = fmin(fmax(alpha, sigma1) * snorm, sigma2 * delta);
        else if (actred < eta1 * prered) delta = fmax(sigma1 * delta, fmin(alpha * snorm, sigma2 * delta));
        else if (actred < eta2 * prered) delta = fmax(sigma1 * delta, fmin(alpha * snorm, sigma3 * delta));
        else delta = fmax(delta, fmin(alpha * snorm, sigma3 * delta));
        bool brk = false;
        D.accept = actred > eta0 * prered;
        if (D.accept) {
            D.iter = pr.iter + 1;
            D.copy_w = D.copy_g = true;
            f = fnew;
            gnorm = norm_of_sumsq(hsq);
            D.gsq = hsq;
            if (gnorm <= pr.eps * pr.gnorm1) brk = true;
        }
        if (!brk) {
            if (f < -1.0e+32) brk = true;
            else if (fabs(actred) <= 0 && prered <= 0) brk = true;
            else if (fabs(actred) <= 1.0e-12 * fabs(f) && fabs(prered) <= 1.0e-12 * fabs(f)) brk = true;
        }
        D.f = f; D.delta = delta; D.gnorm = gnorm;
        if (brk || D.iter > pr.max_iter) D.finished = true;             // while (iter <= max_iter && search)
        else D.start = true;
    }
    if (!(fnew == fnew) || !(D.gnorm == D.gnorm) || !(D.gnorm1 == D.gnorm1)) { D.nan = true; D.finished = true; D.start = false; }
    // trcg exits at once when ||r|| <= 0.1 ||g|| with r = -g (:144), i.e. only for g = 0
    if (D.start && D.gnorm <= 0.1 * D.gnorm) D.nullstep = true;
    return D;
}

// What one trcg trip decides once |s + alpha d| and |r - alpha Hd| are known (bw/Tron.java:150-175), from the problem's
// scalars and the totals of phases A and B. Pure function: phase C and the commit evaluate it identically.
struct CgDecision {
    double alpha, alpha2, beta, rnew;
    bool boundary, end_cg, nan;
};

__device__ __forceinline__ CgDecision cg_decide(const ProbDev &pr, const double *totA, const double *totB)
{
#pragma clang fp contract(off)
    CgDecision D;
    D.alpha = pr.rTr / totA[0];
    const double ss = totB[0], std_ = totB[1], sts = totB[2], dtd = totB[3];
    D.rnew = totB[4];
    const double delta0 = pr.delta;
    const double snorm = norm_of_sumsq(ss);
    D.boundary = snorm > delta0;
    D.nan = !(snorm == snorm);
    D.alpha2 = 0.0; D.beta = 0.0;
    if (D.boundary) {
        // cg reaches trust region boundary (:150-168)
        const double dsq = delta0 * delta0;
        const double rad = sqrt(std_ * std_ + dtd * (dsq - sts));
        if (std_ >= 0) D.alpha2 = (dsq - sts) / (std_ + rad);
        else D.alpha2 = (rad - std_) / dtd;
        D.end_cg = true;
    } else {
        D.beta = D.rnew / pr.rTr;
        D.end_cg = norm_of_sumsq(D.rnew) <= pr.cgtol;                  // loop-top test of the next trip (:144)
    }
    if (D.nan) D.end_cg = true;
    return D;
}

// ---- phase B ------------------------------------------------------------------------------------
template <bool EMU>
__global__ void STEP_LB
k_step_b(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int ch)
{
#pragma clang fp contract(off)
    __shared__ double scratch[96];
    __shared__ double stage[STEP_T];
    __shared__ double totA[4];
    ProbDev &pr = probs[qlist[blockIdx.y]];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    StepGeom G;
    if (!step_geom(pa, ch, G)) return;
    const int tid = threadIdx.x;
    step_gather<4>(pr.pA, G.nwg, totA, stage);
    double *__restrict__ s = pr.s, *__restrict__ d = pr.d;
    const double *__restrict__ Hd = pr.Hd;
    if (phase == PH_CG) {
        // daxpy(alpha, d, s); r' = r - alpha Hd into the other residual buffer; the sums of both continuations
        const double alpha = pr.rTr / totA[0], nalpha = -alpha;
        const double *__restrict__ rc = pr.rb[pr.rsel];
        double *__restrict__ rn = pr.rb[pr.rsel ^ 1];
        double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        // EMU: r'.r' as a grid-rounded sum (see grid_of_sum): u from the head columns, one per thread
        double ugrid = 0.0;
        if (EMU) {
            double ht[1] = {0.0};
            if (tid < min(STEP_HEAD, G.n)) { const double r1 = gld(rc + tid) + nalpha * gld(Hd + tid); ht[0] = r1 * r1; }
            block_allreduce_sum<1>(ht, scratch);
            ugrid = grid_of_sum(ht[0]);
        }
        for (int jb = G.j0 + tid; jb < G.j1; jb += STEP_XB * STEP_T) {
            double dv[STEP_XB], sv[STEP_XB], rv[STEP_XB], hv[STEP_XB];
#pragma unroll
            for (int u = 0; u < STEP_XB; u++) {
                const int jc = min(jb + u * STEP_T, G.j1 - 1);
                dv[u] = SLD(d + jc); sv[u] = SLD(s + jc); rv[u] = SLD(rc + jc); hv[u] = SLD(Hd + jc);
            }
#pragma unroll
            for (int u = 0; u < STEP_XB; u++) {
                const int j = jb + u * STEP_T;
                if (j >= G.j1) continue;
                const double s1 = sv[u] + alpha * dv[u];               // daxpy(alpha, d, s)
                SST(s + j, s1);
                acc[0] += s1 * s1;
                const double sb = s1 + nalpha * dv[u];                 // the boundary case steps back first (:153)
                acc[1] += sb * dv[u];
                acc[2] += sb * sb;
                acc[3] += dv[u] * dv[u];
                const double r1 = rv[u] + nalpha * hv[u];              // daxpy(-alpha, Hd, r)
                SST(rn + j, r1);
                const double term = r1 * r1;
                acc[4] += (EMU && j >= STEP_HEAD) ? round_to_grid(term, ugrid) : term;
            }
        }
        block_allreduce_sum<5>(acc, scratch);
        if (tid == 0) {
            double *__restrict__ px = pr.pB + G.wg * STEP_NP;
#pragma unroll
            for (int k = 0; k < 5; k++) px[k] = acc[k];
        }
        return;
    }
    // ---- PH_EVAL0 / PH_EVAL
    const EvalDecision D = eval_decide(pr, phase, totA);
    double *__restrict__ w = pr.w, *__restrict__ w_new = pr.w_new, *__restrict__ g = pr.g;
    double *__restrict__ r0 = pr.rb[0];
    if (!(D.copy_w || D.copy_g || D.start)) return;
    for (int jb = G.j0 + tid; jb < G.j1; jb += STEP_XB * STEP_T) {
        double hv[STEP_XB], gv[STEP_XB], wn[STEP_XB], wv[STEP_XB];
#pragma unroll
        for (int u = 0; u < STEP_XB; u++) {
            const int jc = min(jb + u * STEP_T, G.j1 - 1);
            hv[u] = gld(Hd + jc);
            gv[u] = D.copy_g ? 0.0 : gld(g + jc);
            wn[u] = (D.copy_w || D.nullstep) ? gld(w_new + jc) : 0.0;
            wv[u] = (D.nullstep && !D.copy_w) ? gld(w + jc) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < STEP_XB; u++) {
            const int j = jb + u * STEP_T;
            if (j >= G.j1) continue;
            if (D.copy_w) gst(w + j, wn[u]);
            if (D.copy_g) gst(g + j, hv[u]);
            if (D.start) {
                // trcg prologue (:133-141): s = 0, r = -g, d = r
                const double gj = D.copy_g ? hv[u] : gv[u];
                const double rj = -gj;
                gst(s + j, 0.0); gst(r0 + j, rj); gst(d + j, rj);
                // the CG loop exits at once with s = 0: the (null) step is evaluated like any other
                if (D.nullstep) gst(w_new + j, (D.copy_w ? wn[u] : wv[u]) + 1.0 * 0.0);
            }
        }
    }
}

// ---- phase C (CG ticks only) ----------------------------------------------------------------------
__global__ void STEP_LB
k_step_c(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int ch)
{
#pragma clang fp contract(off)
    __shared__ double scratch[64];
    __shared__ double stage[STEP_T];
    __shared__ double totA[4], totB[5];
    ProbDev &pr = probs[qlist[blockIdx.y]];
    if (pr.phase != PH_CG) return;
    const PartDev &pa = parts[pr.part];
    StepGeom G;
    if (!step_geom(pa, ch, G)) return;
    const int tid = threadIdx.x;
    step_gather<4>(pr.pA, G.nwg, totA, stage);
    step_gather<5>(pr.pB, G.nwg, totB, stage);
    const CgDecision D = cg_decide(pr, totA, totB);
    const bool boundary = D.boundary, end_cg = D.end_cg;
    const double nalpha = -D.alpha, alpha2 = D.alpha2, nalpha2 = -D.alpha2, beta = D.beta;
    double *__restrict__ s = pr.s, *__restrict__ d = pr.d;
    const double *__restrict__ Hd = pr.Hd;
    const double *__restrict__ rc = pr.rb[pr.rsel];
    double *__restrict__ rn = pr.rb[pr.rsel ^ 1];
    const double *__restrict__ w = pr.w, *__restrict__ g = pr.g;
    double *__restrict__ w_new = pr.w_new;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int jb = G.j0 + tid; jb < G.j1; jb += STEP_XB * STEP_T) {
        double dv[STEP_XB], r1[STEP_XB], sv[STEP_XB], hv[STEP_XB], rv[STEP_XB], wv[STEP_XB], gv[STEP_XB];
#pragma unroll
        for (int u = 0; u < STEP_XB; u++) {
            const int jc = min(jb + u * STEP_T, G.j1 - 1);
            dv[u] = SLD(d + jc);
            r1[u] = boundary ? 0.0 : SLD(rn + jc);
            sv[u] = (boundary || end_cg) ? SLD(s + jc) : 0.0;
            hv[u] = boundary ? SLD(Hd + jc) : 0.0;
            rv[u] = boundary ? SLD(rc + jc) : 0.0;
            wv[u] = end_cg ? SLD(w + jc) : 0.0;
            gv[u] = end_cg ? SLD(g + jc) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < STEP_XB; u++) {
            const int j = jb + u * STEP_T;
            if (j >= G.j1) continue;
            double sf = sv[u], rf = r1[u];
            if (boundary) {
                const double sb = sv[u] + nalpha * dv[u];              // daxpy(-alpha, d, s)
                sf = sb + alpha2 * dv[u];                              // daxpy(alpha', d, s)
                gst(s + j, sf);
                rf = rv[u] + nalpha2 * hv[u];                          // daxpy(-alpha', Hd, r)
                gst(rn + j, rf);
            } else {
                double dj = dv[u];
                if (beta != 1.0) dj = dj * beta;                       // scale(beta, d)
                gst(d + j, dj + 1.0 * r1[u]);                               // daxpy(one, r, d)
            }
            if (end_cg) {
                // back in tron(): w_new = w + s, gs, prered (:69-73)
                gst(w_new + j, wv[u] + 1.0 * sf);
                acc[0] += gv[u] * sf;
                acc[1] += sf * rf;
                acc[2] += sf * sf;
            }
        }
    }
    if (!end_cg) return;
    block_allreduce_sum<3>(acc, scratch);
    if (tid == 0) {
        double *__restrict__ px = pr.pC + G.wg * STEP_NP;
        px[0] = acc[0]; px[1] = acc[1]; px[2] = acc[2];
    }
}

// ---- commit: one workgroup per problem writes the scalars of the tick ---------------------------------------------
__global__ void __launch_bounds__(STEP_T)
k_step_commit(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int ch,
              int *__restrict__ done_counter)
{
#pragma clang fp contract(off)
    __shared__ double stage[STEP_T];
    __shared__ double totA[4], totB[5], totC[3];
    constexpr int CW = 64;                                           // problems of <= CW chunks: all three partial arrays in ONE round trip
    __shared__ double all3[3][CW * STEP_NP];
    ProbDev &pr = probs[qlist[blockIdx.x]];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    const int nwg = (pa.n_local + ch - 1) / ch;
    const bool batched = nwg <= CW;
    if (batched) {
        // this launch is one workgroup per problem on a mostly idle chip: a chain of dependent loads IS its duration. The chunk-ordered
        // sums are those of step_gather (same order, same values).
        const int cnt = nwg * STEP_NP;
        for (int i = threadIdx.x; i < cnt; i += STEP_T) {
            all3[0][i] = pr.pA[i];
            all3[1][i] = (phase == PH_CG) ? pr.pB[i] : 0.0;
            all3[2][i] = (phase == PH_CG) ? pr.pC[i] : 0.0;          // (read whether or not the trcg call ends: stale values are not used)
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            const int arr = threadIdx.x < 4 ? 0 : (threadIdx.x < 9 ? 1 : 2), k = threadIdx.x - (arr == 0 ? 0 : (arr == 1 ? 4 : 9));
            double a = 0.0;
            for (int w = 0; w < nwg; w++) a += all3[arr][w * STEP_NP + k];
            (arr == 0 ? totA : (arr == 1 ? totB : totC))[k] = a;
        }
        __syncthreads();
    } else step_gather<4>(pr.pA, nwg, totA, stage);
    if (phase == PH_CG) {
        if (!batched) step_gather<5>(pr.pB, nwg, totB, stage);
        const CgDecision D = cg_decide(pr, totA, totB);
        if (D.end_cg && !batched) step_gather<3>(pr.pC, nwg, totC, stage);        // (uniform: every thread holds the same D)
        if (threadIdx.x != 0) return;
        if (!D.boundary) pr.rTr = D.rnew;
        pr.rsel ^= 1;
        pr.cg_iter += 1;
        pr.ticks += 1;
        if (D.nan) pr.status = ST_NAN;       // the EVAL tick that follows still runs; the host reports the status
        if (D.end_cg) {
            pr.gs = totC[0];
            pr.prered = -0.5 * (totC[0] - totC[1]);
            pr.snorm = norm_of_sumsq(totC[2]);
            pr.newton += 1;
            pr.cg_total += pr.cg_iter;
            pr.phase = PH_EVAL;
        }
        return;
    }
    const EvalDecision D = eval_decide(pr, phase, totA);
    if (threadIdx.x != 0) return;
    pr.f = D.f; pr.delta = D.delta; pr.gnorm = D.gnorm; pr.gnorm1 = D.gnorm1; pr.gsq = D.gsq; pr.iter = D.iter;
    pr.ticks += 1;
    if (phase == PH_EVAL0 || D.accept) pr.dsel ^= 1;
    if (D.accept) pr.accepted += 1;
    if (D.nan) pr.status = ST_NAN;
    if (D.start) {
        pr.rTr = D.gsq;                    // r = -g: r.r = g.g
        pr.cgtol = 0.1 * D.gnorm;
        pr.cg_iter = 0;
        pr.rsel = 0;
        if (D.nullstep) { pr.gs = 0.0; pr.prered = -0.5 * (0.0 - 0.0); pr.snorm = 0.0; pr.newton += 1; pr.phase = PH_EVAL; }
        else pr.phase = PH_CG;
    }
    if (D.finished) {
        pr.phase = PH_DONE;
        atomicAdd(done_counter, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// Whole solve in one launch for SMALL CSR partitions (config #1: 125 rows x 200 features): at that size a tick of three
// kernels costs ~30 us of launch/dependent-load latency for ~1 us of work, so one workgroup per problem runs the tick
// loop itself -- row pass, column pass and the same TRON step body, separated by workgroup barriers -- until its problem
// is DONE (or max_ticks, then the host relaunches). Row sums run in the same entry order as the sliced kernels; the
// loss / coefficient sums are one block reduction instead of per-chunk partials.
// ------------------------------------------------------------------------------------------------
// LDSV: every fp64 work vector of the problem (8 n-vectors, wd x2, coef, item sums, partial sums) lives in LDS for the whole
// solve and is written back at the end -- for problems of a few hundred features the tick loop then waits on LDS instead of
// ~20 dependent L2 round trips per tick.
// SEQ: the order-faithful verification mode (MLX_FAITHFUL=1): ONE lane per row / per (unsplit) column so that every row and
// column sum runs in the reference's order, one thread for every n- or l-long reduction (tron_step_body<true>), grad(0)
// from its own pass at w = 0 like bw/Tron.java:50-53, and the portable exp/log1p the oracle's verification twin uses too.
// XL (1: uint8 ids, 2: uint16 ids; needs LDSV): the partition's index and value arrays live in LDS too -- CSR and CSC copies with
// narrow ids, row / item pointers, item destinations and column pointers -- when everything fits beside the vectors (config #1:
// 125 rows x 200 features x 12.5 K non-zeros = 125 KiB + 17 KiB of vectors). A tick then waits on LDS instead of ~7 dependent L2
// round trips.
template <bool HASVAL, bool LDSV, bool SEQ, int XL = 0>
__global__ void __launch_bounds__(1024)
k_solve_small(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nprob, int max_ticks,
              int *__restrict__ done_counter, int wave_step, int grid_dots)
{
#pragma clang fp contract(off)
    static_assert(XL == 0 || (LDSV && !SEQ), "X in LDS rides on the LDS-resident vectors");
    constexpr int G = SEQ ? 1 : 8, U = 8;   // lanes per row / per column item, loads in flight per lane
    using IdT = typename std::conditional<XL == 1, uint8_t, typename std::conditional<XL == 2, uint16_t, int32_t>::type>::type;
    __shared__ double scratch[64];
    __shared__ double stage[1024];
    extern __shared__ double dyn[];
    __shared__ ProbDev prl;
    __shared__ PartDev pal;
    // static LDS of this kernel beside the <= 150 KiB of dynamic LDS the launcher may ask for (mlxk_solve_small): a descriptor that
    // grows past the budget must fail the build, not the launch
    static_assert(sizeof(ProbDev) + sizeof(PartDev) + (64 + 1024) * sizeof(double) + 256 <= (160 - 150) * 1024,
                  "k_solve_small: static LDS + 150 KiB dynamic LDS exceed the 160 KiB of a CU");
    if ((int)blockIdx.x >= nprob) return;
    const int q = qlist[blockIdx.x];
    ProbDev &prg = probs[q];                 // the descriptor in global memory
    const PartDev &pag = parts[prg.part];
    // (nt a CONSTANT: with nt = blockDim.x the same source compiled -- ROCm 7.2 -- to a kernel whose results sat 1e-6 off on
    // the ill-conditioned binary case of tests/test_gpu_parity.py::test_sparse_absent_features_weights_offsets, deterministically and
    // with every counter equal; 512 / 256 / 128 threads were slower on config #1 anyway: 14.0 / 18.3 / 26.8 ms against 11.7)
    const int tid = threadIdx.x;
    constexpr int nt = 1024;
    if (XL != 0) { if (tid == 0) pal = pag; __syncthreads(); }
    const PartDev &pa = XL != 0 ? pal : pag;
    if (LDSV) {
        const int n = pa.n_local, l0 = pa.l, ni = pa.n_items, nbk = pa.n_rowparts;
        if (tid == 0) {
            prl = prg;
            double *p = dyn;
            double **vs[8] = {&prl.w, &prl.w_new, &prl.g, &prl.s, &prl.r, &prl.d, &prl.Hd, &prl.m};
            for (int v = 0; v < 8; v++) { *vs[v] = p; p += n; }
            prl.wd[0] = p; p += l0;
            prl.wd[1] = p; p += l0;
            prl.coef = p; p += l0;
            prl.parts = p; p += ni;
            prl.lossp = p; p += nbk;
            prl.csump = p; p += nbk;
        }
        __syncthreads();
        for (int j = tid; j < n; j += nt) {
            prl.w[j] = prg.w[j]; prl.w_new[j] = prg.w_new[j]; prl.g[j] = prg.g[j]; prl.s[j] = prg.s[j];
            prl.r[j] = prg.r[j]; prl.d[j] = prg.d[j]; prl.Hd[j] = prg.Hd[j]; prl.m[j] = prg.m[j];
        }
        for (int i = tid; i < l0; i += nt) { prl.wd[0][i] = prg.wd[0][i]; prl.wd[1][i] = prg.wd[1][i]; }
        __syncthreads();
    }
    ProbDev &pr = LDSV ? prl : prg;          // the working descriptor
    const int gid = tid / G, gl = tid % G, ng = nt / G;
    const int l = pa.l, nitems = pa.n_items;
    const int32_t *__restrict__ rp = pag.rp;
    const IdT *__restrict__ ci = reinterpret_cast<const IdT *>(pag.ci);          // (XL == 0: IdT is int32_t, the global arrays)
    const float *__restrict__ val = pag.val;
    const int32_t *__restrict__ item_ptr = pag.item_ptr;
    const IdT *__restrict__ cri = reinterpret_cast<const IdT *>(pag.cri);
    const float *__restrict__ cval = pag.cval;
    const int32_t *__restrict__ item_dst = pag.item_dst;
    const float *__restrict__ row_wt = pag.wt, *__restrict__ row_off = pag.off;
    const int8_t *__restrict__ row_y = pag.y;
    if (XL != 0) {
        // carve the X region behind the vectors (same sizes the host summed: mlx_finalize) and copy, ids narrowed
        const int n = pag.n_local, nz = (int)pag.nnz, nf1 = pag.n_feat + 1;
        char *xb = reinterpret_cast<char *>(dyn + (8 * n + 3 * l + nitems + 2 * pag.n_rowparts));
        int32_t *s_rp = reinterpret_cast<int32_t *>(xb); xb += 4 * (size_t)(l + 1);
        int32_t *s_ip = reinterpret_cast<int32_t *>(xb); xb += 4 * (size_t)(nitems + 1);
        int32_t *s_id = reinterpret_cast<int32_t *>(xb); xb += 4 * (size_t)nitems;
        int32_t *s_cp = reinterpret_cast<int32_t *>(xb); xb += 4 * (size_t)nf1;
        float *s_v = reinterpret_cast<float *>(xb); xb += (HASVAL && pag.val) ? 4 * (size_t)nz : 0;
        float *s_cv = reinterpret_cast<float *>(xb); xb += (HASVAL && pag.cval) ? 4 * (size_t)nz : 0;
        float *s_wt = reinterpret_cast<float *>(xb); xb += 4 * (size_t)l;
        float *s_off = reinterpret_cast<float *>(xb); xb += 4 * (size_t)l;
        IdT *s_ci = reinterpret_cast<IdT *>(xb); xb += sizeof(IdT) * (size_t)nz;
        IdT *s_cr = reinterpret_cast<IdT *>(xb); xb += sizeof(IdT) * (size_t)nz;
        int8_t *s_y = reinterpret_cast<int8_t *>(xb);
        for (int i = tid; i < l; i += nt) { s_wt[i] = pag.wt[i]; s_off[i] = pag.off[i]; s_y[i] = pag.y[i]; }
        for (int i = tid; i <= l; i += nt) s_rp[i] = pag.rp[i];
        for (int i = tid; i <= nitems; i += nt) s_ip[i] = pag.item_ptr[i];
        for (int i = tid; i < nitems; i += nt) s_id[i] = pag.item_dst[i];
        for (int i = tid; i < nf1; i += nt) s_cp[i] = pag.col_ptr[i];
        for (int k = tid; k < nz; k += nt) {
            s_ci[k] = (IdT)pag.ci[k];
            s_cr[k] = (IdT)pag.cri[k];
            if (HASVAL && pag.val) s_v[k] = pag.val[k];
            if (HASVAL && pag.cval) s_cv[k] = pag.cval[k];
        }
        if (tid == 0) pal.col_ptr = s_cp;
        row_wt = s_wt; row_off = s_off; row_y = s_y;
        rp = s_rp; item_ptr = s_ip; item_dst = s_id; ci = s_ci; cri = s_cr;
        if (HASVAL && pag.val) val = s_v;
        if (HASVAL && pag.cval) cval = s_cv;
        __syncthreads();
    }
    using VP = typename std::conditional<LDSV, lds_dptr, double *>::type;      // see lds_dptr
    const VP coef = (VP)pr.coef, segsum = (VP)pr.parts;
    for (int b = 1 + tid; b < pa.n_rowparts; b += nt) { pr.lossp[b] = 0.0; pr.csump[b] = 0.0; }      // (this kernel leaves ONE sum, in slot 0)
    // lane-group sum of sparse dot products: lane gl takes entries k0+gl, k0+gl+G, ...; fixed xor tree inside the group
    auto group_dot = [&](const IdT *__restrict__ idxs, const float *__restrict__ vals, const VP vec, int k0, int k1) -> double {
        double a = 0.0;
        for (int kb = k0 + gl; kb < k1; kb += G * U) {
            int idx[U];
            float xv[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int kk = min(kb + u * G, k1 - 1);
                idx[u] = (int)idxs[kk];
                if (HASVAL) xv[u] = vals ? vals[kk] : 1.0f;
            }
            double vv[U];
#pragma unroll
            for (int u = 0; u < U; u++) vv[u] = vec[idx[u]];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double term = HASVAL ? vv[u] * (double)xv[u] : vv[u];
                if (kb + u * G < k1) a = a + term;
            }
        }
        return group_allreduce_sum<G>(a);
    };
    // SEQ: tick -1 is the reference's fun(0) + grad(0) (bw/Tron.java:50-53): an EVAL pass at w = 0 (d is free before the
    // first trcg) whose X'c is kept in c0f
    const bool c0_tick = SEQ && pr.phase == PH_EVAL0;
    if (c0_tick) {
        for (int j = tid; j < pa.n_local; j += nt) pr.d[j] = 0.0;
    }
#ifdef MLX_SMALL_PROFILE
    unsigned long long tprev = clock64();
#endif
    for (int tick = c0_tick ? -1 : 0; tick < max_ticks; tick++) {
        __syncthreads();
        SPROF(4);                            // (waiting for the step's wave at the loop barrier)
        const int phase = pr.phase;
        if (phase == PH_DONE) break;
#ifdef MLX_SMALL_PROFILE
        if (blockIdx.x == 0 && tid == 0) g_small_prof[5] += 1;
#endif
        const bool cg = (phase == PH_CG) && tick >= 0;
        const VP v = (VP)(tick < 0 ? pr.d : (cg ? pr.d : pr.w_new));
        const VP wdcur = (VP)pr.wd[pr.dsel], wdnew = (VP)pr.wd[pr.dsel ^ 1];
        const double vb = v[pa.n_feat];
        double red[2] = {0.0, 0.0};
        // EVAL ticks of short partitions evaluate the row maps in a second, compact sweep (one row per lane): in the sweep below only
        // lane 0 of every 8-lane group holds a row, so the ~400 fp64 instructions of exp + log1p would issue on all 16 waves for 8 rows
        // each. The sums are then collected by the same lanes in the same order as before (bit-identical).
        const bool compact = LDSV && !SEQ && !cg && l <= 1024;
        for (int rowb = 0; rowb < l; rowb += ng) {
            const int row = rowb + gid;
            const bool valid = row < l;
            const int rowc = min(row, l - 1);
            const int k0 = rp[rowc], k1 = valid ? rp[rowc + 1] : k0;
            const double a = group_dot(ci, val, v, k0, k1);
            if (valid && gl == 0) {
                const double t = a + vb;
                if (compact) { coef[row] = t; continue; }          // parked for the compact sweep
                double cf;
                if (cg) {
                    cf = wdcur[row] * t;
                } else {
                    double loss, wdv;
                    row_eval<SEQ>(t + (double)row_off[row], (int)row_y[row], (double)row_wt[row], loss, wdv, cf);
                    wdnew[row] = wdv;
                    red[0] += loss;
                    if (SEQ) pr.rowtmp[row] = loss;
                }
                coef[row] = cf;
                red[1] += cf;
            }
        }
        if (compact) {
            __syncthreads();
            if (tid < l) {
                double loss, wdv, cf;
                row_eval<false>(coef[tid] + (double)row_off[tid], (int)row_y[tid], (double)row_wt[tid], loss, wdv, cf);
                wdnew[tid] = wdv;
                coef[tid] = cf;
                stage[tid] = loss;
            }
            __syncthreads();
            for (int rowb = 0; rowb < l; rowb += ng) {
                const int row = rowb + gid;
                if (row < l && gl == 0) { red[0] += stage[row]; red[1] += coef[row]; }
            }
        }
        SPROF(0);                            // row pass
        // loss and coefficient sums over the block: wave tree, then the 16 wave sums in wave order (block_allreduce_sum's order).
        // When the step runs on the first wave alone, that wave also adds the wave sums (after the column pass): one barrier
        // here instead of three, and 15 waves spared the 32 LDS reads.
        const bool wave_red = XL != 0 && wave_step != 0;
        if (wave_red) {
            red[0] = wave_allreduce_sum(red[0]); red[1] = wave_allreduce_sum(red[1]);
            if ((tid & 63) == 0) { scratch[tid >> 6] = red[0]; scratch[16 + (tid >> 6)] = red[1]; }
        } else {
            block_allreduce_sum<2>(red, scratch);
            if (tid == 0) { pr.lossp[0] = red[0]; pr.csump[0] = red[1]; }
        }
        __syncthreads();
        SPROF(1);                            // its block reduction
        for (int itb = 0; itb < nitems; itb += ng) {
            const int it = itb + gid;
            const bool valid = it < nitems;
            const int itc = min(it, nitems - 1);
            const int k0 = item_ptr[itc], k1 = valid ? item_ptr[itc + 1] : k0;
            const double a = group_dot(cri, cval, coef, k0, k1);
            if (valid && gl == 0 && k1 > k0) segsum[item_dst[itc]] = a;
        }
        __syncthreads();
        if (SEQ && tick < 0) {
            // c0f = X' t(0): unsplit columns, so a column's slot IS its sum; the intercept's is the row-ordered sum of coef
            const double ci = seq_sum(pr.coef, l, scratch, stage);
            for (int j = tid; j < pa.n_local; j += nt)
                pr.c0f[j] = (j == pa.n_feat) ? ci : (pa.col_ptr[j + 1] > pa.col_ptr[j] ? segsum[pa.col_ptr[j]] : 0.0);
            continue;
        }
        SPROF(2);                            // column pass + barrier
        if constexpr (XL != 0) {
            // the step on the first wave alone (WaveTeam); the others wait at the loop's barrier
            if (wave_step) {
                if (tid < 64) {
                    double a0 = 0, a1 = 0;
                    for (int w = 0; w < nt / 64; w++) { a0 += scratch[w]; a1 += scratch[16 + w]; }
                    if (tid == 0) { pr.lossp[0] = a0; pr.csump[0] = a1; }
                    __builtin_amdgcn_wave_barrier();
                    tron_step_body<SEQ, WaveTeam, VP>(pa, pr, scratch, stage, done_counter, grid_dots != 0);
                }
            }
            else tron_step_body<SEQ, BlockTeam, VP>(pa, pr, scratch, stage, done_counter, grid_dots != 0);
        } else {
            tron_step_body<SEQ, BlockTeam, VP>(pa, pr, scratch, stage, done_counter, grid_dots != 0);
        }
        SPROF(3);                            // step
    }
    if (LDSV) {
        // write the state back (everything a relaunch, the outputs kernels or the host read)
        __syncthreads();
        const int n = pa.n_local, l0 = pa.l;
        for (int j = tid; j < n; j += nt) {
            prg.w[j] = prl.w[j]; prg.w_new[j] = prl.w_new[j]; prg.g[j] = prl.g[j]; prg.s[j] = prl.s[j];
            prg.r[j] = prl.r[j]; prg.d[j] = prl.d[j]; prg.Hd[j] = prl.Hd[j];
        }
        for (int i = tid; i < l0; i += nt) { prg.wd[0][i] = prl.wd[0][i]; prg.wd[1][i] = prl.wd[1][i]; }
        if (tid == 0) {
            prg.phase = prl.phase; prg.dsel = prl.dsel; prg.iter = prl.iter; prg.cg_iter = prl.cg_iter;
            prg.newton = prl.newton; prg.accepted = prl.accepted; prg.cg_total = prl.cg_total; prg.ticks = prl.ticks;
            prg.status = prl.status; prg.f = prl.f; prg.delta = prl.delta; prg.gnorm = prl.gnorm; prg.gnorm1 = prl.gnorm1;
            prg.rTr = prl.rTr; prg.cgtol = prl.cgtol; prg.prered = prl.prered; prg.gs = prl.gs;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// reducer outputs: beta_k and u_k + beta_k as float32 (jobs/RegressionAdmmTrain.java:706-711;
// absent features keep priorMean = z~ - u_k, llf/LibLinear.java:373-383)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_outputs_absent(const ProbDev *__restrict__ probs, int nprob, int n_lambda, int n_global,
                 const float *__restrict__ z32, const float *__restrict__ u, float *__restrict__ B,
                 float *__restrict__ UPX)
{
#pragma clang fp contract(off)
    const int q = blockIdx.y;
    if (q >= nprob) return;
    const ProbDev &pr = probs[q];
    const int64_t base = ((int64_t)pr.part * n_lambda + pr.lambda_idx) * n_global;
    const float *__restrict__ zl = z32 + (int64_t)pr.lambda_idx * n_global;
    for (int gj = blockIdx.x * 256 + threadIdx.x; gj < n_global; gj += gridDim.x * 256) {
        const double zt = (double)zl[gj], uj = (double)u[base + gj];
        const double beta = -1.0 * uj + 1.0 * zt;
        B[base + gj] = (float)beta;
        UPX[base + gj] = (float)(1.0 * uj + 1.0 * beta);
    }
}
__global__ void __launch_bounds__(256)
k_outputs_present(const PartDev *__restrict__ parts, const ProbDev *__restrict__ probs, int nprob, int n_lambda,
                  int n_global, const float *__restrict__ u, float *__restrict__ B, float *__restrict__ UPX)
{
#pragma clang fp contract(off)
    const int q = blockIdx.y;
    if (q >= nprob) return;
    const ProbDev &pr = probs[q];
    const PartDev &pa = parts[pr.part];
    const int64_t base = ((int64_t)pr.part * n_lambda + pr.lambda_idx) * n_global;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < pa.n_local; j += gridDim.x * 256) {
        const int gj = pa.l2g[j];
        const double uj = (double)u[base + gj], wj = pr.w[j];
        B[base + gj] = (float)wj;
        UPX[base + gj] = (float)(1.0 * uj + 1.0 * wj);
    }
}

// ------------------------------------------------------------------------------------------------
// RegressionTest scoring (jobs/RegressionTest.java:147-175): pred_i = (float)(offset_i + eval(features_i)) with
// eval = base + sum_k coef[key_k] * value_k in record order, unknown names skipped (models/LinearModel.java:241-257);
// base = -log(exp(-intercept)) is evaluated once on the host. One thread per row keeps the reference's summation order.
// ------------------------------------------------------------------------------------------------
template <bool HASVAL>
__global__ void __launch_bounds__(256)
k_score_rows(int l, const int64_t *__restrict__ rp, const int32_t *__restrict__ gi, const double *__restrict__ val,
             const double *__restrict__ off, const double *__restrict__ z, double base, float *__restrict__ pred)
{
#pragma clang fp contract(off)
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= l) return;
    double result = base;
    for (int64_t k = rp[row]; k < rp[row + 1]; k++) {
        const int g = gi[k];
        if (g >= 0) result += z[g] * (HASVAL ? val[k] : 1.0);
    }
    pred[row] = (float)(off[row] + result);
}

// ------------------------------------------------------------------------------------------------
// Posterior variance at the mode (LibLinear.train with computePosteriorVar, llf/LibLinear.java:314-337):
// H = diag(1/priorVar) + X' D X (llf/LogisticRegressionL2.java:258-297), D_ii = weight_i p_i (1 - p_i) = the wd[] an EVAL
// pass leaves behind. The n x n Gram build is the one GEMM-shaped piece of this code base: fp64 MFMA.
// ------------------------------------------------------------------------------------------------
// CSR partition -> temporary dense tile (zero-initialised by the caller), duplicates accumulate like the reference's loops
template <bool HASVAL>
__global__ void __launch_bounds__(256)
k_densify(int l, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, const float *__restrict__ val,
          float *__restrict__ X, int64_t ld)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= l) return;
    for (int k = rp[row]; k < rp[row + 1]; k++) X[(int64_t)row * ld + ci[k]] += HASVAL ? val[k] : 1.0f;
}

// per row chunk: s1[c] = sum_i wd_i x_ic (the intercept's Hessian row), s2[c] = sum_i wd_i x_ic^2 (hessianDiagonal,
// llf/LogisticRegressionL2.java:304-327). part[chunk][2][ld]; thread = column, rows of the chunk in order.
__global__ void __launch_bounds__(256)
k_hess_colsums(const float *__restrict__ X, int64_t ld, int l, int rows_per_chunk, const double *__restrict__ wd,
               double *__restrict__ part)
{
#pragma clang fp contract(off)
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ld) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(l, r0 + rows_per_chunk);
    double a1 = 0.0, a2 = 0.0;
    for (int r = r0; r < r1; r++) {
        const double x = (double)X[(int64_t)r * ld + c], q = wd[r];
        a1 += q * x;
        a2 += q * x * x;
    }
    part[((int64_t)blockIdx.y * 2 + 0) * ld + c] = a1;
    part[((int64_t)blockIdx.y * 2 + 1) * ld + c] = a2;
}

// out[0][c] = sum_chunks part[.][0][c], out[1][c] likewise (fixed chunk order); out[2][0] = sum_i wd_i
__global__ void __launch_bounds__(256)
k_hess_colsums_reduce(const double *__restrict__ part, int nchunk, int64_t ld, const double *__restrict__ wd, int l,
                      double *__restrict__ out)
{
    __shared__ double scratch[16];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < ld) {
        double a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < nchunk; k++) { a1 += part[((int64_t)k * 2) * ld + c]; a2 += part[((int64_t)k * 2 + 1) * ld + c]; }
        out[c] = a1;
        out[ld + c] = a2;
    }
    if (blockIdx.x == 0) {
        double v[1] = {0.0};
        for (int i = threadIdx.x; i < l; i += 256) v[0] += wd[i];
        block_allreduce_sum<1>(v, scratch);
        if (threadIdx.x == 0) out[2 * ld] = v[0];
    }
}

// hessianDiagonal of a CSR partition without densifying it (any n_local): one thread per column item adds wd_i x_ic^2 over
// the item's entries in row order and writes the item's slot; the host adds a column's slots (llf/LogisticRegressionL2.java:304-327)
template <bool HASVAL>
__global__ void __launch_bounds__(256)
k_hess_diag_items(int n_items, const int32_t *__restrict__ item_ptr, const int32_t *__restrict__ item_dst,
                  const int32_t *__restrict__ cri, const float *__restrict__ cval, const double *__restrict__ wd,
                  double *__restrict__ slots)
{
#pragma clang fp contract(off)
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= n_items) return;
    const int dst = item_dst[it];
    if (dst < 0) return;
    double a = 0.0;
    for (int k = item_ptr[it]; k < item_ptr[it + 1]; k++) {
        const double x = HASVAL ? (double)cval[k] : 1.0;
        a += wd[cri[k]] * x * x;
    }
    slots[dst] = a;
}

// X' D X, lower-triangle 128 x 128 blocks, rows split in `ksplit` ranges: P[ks][npad][npad] partial Gram matrices.
// One wave = 64 x 64 outputs = 4 x 4 tiles of v_mfma_f64_16x16x4_f64 (A[i][k] = wd_row x[row][m0+i], B[k][j] = x[row][n0+j],
// lane = (i|j = lane & 15, k = lane >> 4); C/D: col = lane & 15, row = (lane >> 4) + 4 reg). The operands of the next 4 rows
// are fetched while the 16 MFMAs of the current rows run.
typedef double d4_t __attribute__((ext_vector_type(4)));
// Column mapping of the 4 tiles of a wave side: tile t, lane index ii <-> matrix column base + 4 ii + t, so that a lane's
// four A (and four B) operands of one row are ONE 16-byte load and a 16-lane group reads 256 contiguous bytes of the row.
// KU k-steps (4 rows each) are fetched ahead while the 16 KU MFMAs of the current rows run.
template <int KU>
__global__ void __launch_bounds__(512)
k_gram_f64(const float *__restrict__ X, int64_t ld, int l, const double *__restrict__ wd, const int2 *__restrict__ blocks,
           int rows_per_split, double *__restrict__ P, int npad, int nf)
{
    // Column nf of the Gram matrix is an IMPLICIT column of ones (the intercept: H[nf][c] = sum_i wd_i x_ic, H[nf][nf] = sum_i
    // wd_i, llf/LogisticRegressionL2.java:259-297), so the intercept's row costs no pass of its own.
    // 8 waves = two groups of 4: each group owns one row split of the same 128 x 128 block, so that two waves share every
    // SIMD (one wave per SIMD reaches only ~45 % of the f64 MFMA rate, two reach 98 %: tools/mfma_f64_probe.hip). The two
    // groups' tiles are added through LDS at the end (group 1 parks its 4 x 64 x 64 accumulators there, group 0 adds its own
    // and stores): one partial Gram block per WORKGROUP instead of one per group -- half the store traffic of the tail and half
    // the partial matrices k_gram_finish adds.
    extern __shared__ __attribute__((aligned(16))) double red[];     // [4 waves][16 tiles][4 regs][64 lanes] = 128 KiB
    const int2 bb = blocks[blockIdx.x];
    const int wave = (threadIdx.x >> 6) & 3, lane = threadIdx.x & 63;
    const int grp = threadIdx.x >> 8;
    const int split = blockIdx.y * 2 + grp;
    const int ii = lane & 15, kk = lane >> 4;
    const int m0 = bb.x * 128 + (wave >> 1) * 64, n0 = bb.y * 128 + (wave & 1) * 64;
    // (the row range is the same for a whole wave: as scalars, the row bases of the operand loads are scalar arithmetic too)
    const int r0 = __builtin_amdgcn_readfirstlane(min(l, split * rows_per_split)), r1 = __builtin_amdgcn_readfirstlane(min(l, r0 + rows_per_split));
    d4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4_t){0.0, 0.0, 0.0, 0.0};
    // ld is a multiple of 4: a 4-column group is either entirely inside the tile or entirely padding
    const int cma = m0 + 4 * ii, cna = n0 + 4 * ii;
    const bool vm = cma < ld, vn = cna < ld;
    const int cm = vm ? cma : 0, cn = vn ? cna : 0;
    const int oa = nf - cma, ob = nf - cna;                  // which of the lane's 4 columns (if any) is the ones column
    const bool aq = cma <= nf;                                // the lane's A columns reach into [0, nf]
    // EDGE = the block touches column nf (the ones column) or the padding behind it: only those blocks (8 of 36 at n = 1001) pay
    // for the per-operand selects -- 96 v_cndmask per 64 MFMAs in the main loop of the one-size-fits-all form
    // Two operand buffers of 4 KU rows used in turn (no register copies between them): while the 16 KU MFMAs of one run, the loads of
    // the other -- issued a whole batch earlier -- land. (The first form rotated `next -> current` registers at the loop top; the
    // compiler moved part of those copies into the body behind the fresh loads, so every trip waited for loads it had just issued.)
    // Full batches run without any row test or clamp (uniform row base + one 32-bit lane offset); the last, partial batch of a split
    // takes the clamped form once. (A ring of 8 single k-step slots refilled one by one -- loads issued 112 instead of 64 MFMAs ahead,
    // pinned in place with sched_barrier -- was no faster: 0.650 against 0.665 of peak; profiles/r5_notes.md.)
    struct GramOps { float4 a[KU], b[KU]; double q[KU]; };
    const unsigned offa = (unsigned)kk * (unsigned)ld + (unsigned)cm, offb = (unsigned)kk * (unsigned)ld + (unsigned)cn;    // (< 2^31: 4 rows of the tile)
    auto run = [&](auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        auto fetch_full = [&](GramOps &o, int r) {          // rows r .. r + 4 KU - 1, all below r1 (r uniform)
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const float *__restrict__ xr = X + (int64_t)(r + 4 * u) * ld;
                o.a[u] = *reinterpret_cast<const float4 *>(xr + offa);
                o.b[u] = *reinterpret_cast<const float4 *>(xr + offb);
                o.q[u] = wd[r + 4 * u + kk];
            }
        };
        auto fetch_tail = [&](GramOps &o, int r) {          // the split's last rows: beyond r1 the weight is 0 and the row clamped
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int row = r + 4 * u + kk;
                const int rc = min(row, l - 1);
                o.q[u] = (row < r1) ? wd[rc] : 0.0;
                const float *__restrict__ xr = X + (int64_t)rc * ld;
                o.a[u] = *reinterpret_cast<const float4 *>(xr + cm);
                o.b[u] = *reinterpret_cast<const float4 *>(xr + cn);
            }
        };
        auto compute = [&](const GramOps &o) {
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const double qq = (!EDGE || aq) ? o.q[u] : 0.0;
                const double xa[4] = {(double)o.a[u].x, (double)o.a[u].y, (double)o.a[u].z, (double)o.a[u].w};
                const double xb[4] = {(double)o.b[u].x, (double)o.b[u].y, (double)o.b[u].z, (double)o.b[u].w};
                double a[4], b[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    a[t] = EDGE ? qq * (t == oa ? 1.0 : (vm ? xa[t] : 0.0)) : qq * xa[t];
                    b[t] = EDGE ? (t == ob ? 1.0 : (vn ? xb[t] : 0.0)) : xb[t];
                }
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int nt = 0; nt < 4; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
        };
        constexpr int BR = 4 * KU;
        const int nfull = (r1 - r0) / BR;
        GramOps A, B;
        if (nfull > 0) {
            fetch_full(A, r0);
            int i = 0;
            for (; i + 2 <= nfull; i += 2) {                 // A holds batch i
                fetch_full(B, r0 + (i + 1) * BR);
                compute(A);
                fetch_full(A, r0 + min(i + 2, nfull - 1) * BR);       // (unconditional -- the last trip's may be a repeat: a branch here
                compute(B);                                          //  makes compute(B) wait for ALL loads, the fresh ones included)
            }
            if (i < nfull) compute(A);
        }
        if (r0 + nfull * BR < r1) { fetch_tail(A, r0 + nfull * BR); compute(A); }
    };
    if ((bb.x + 1) * 128 > nf) run(std::true_type{});        // (bb.y <= bb.x: the column range is never further out than the row range)
    else run(std::false_type{});
    // group 1 -> LDS -> group 0 adds and stores
    double *__restrict__ mine = red + (size_t)wave * 4096 + lane;
    if (grp == 1) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) mine[((mt * 4 + nt) * 4 + reg) * 64] = acc[mt][nt][reg];
    }
    __syncthreads();
    if (grp == 1) return;
    double *__restrict__ out = P + (int64_t)blockIdx.y * npad * npad;
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int row = m0 + 4 * (kk + 4 * reg) + mt, col = n0 + 4 * ii + nt;
                out[(int64_t)row * npad + col] = acc[mt][nt][reg] + mine[((mt * 4 + nt) * 4 + reg) * 64];
            }
}

// H[n][n] (n = nf + 1, row-major): feature block from the partial Gram matrices (lower triangle, mirrored), the intercept's
// row/column from the column sums, 1/priorVar on the diagonal (llf/LogisticRegressionL2.java:259,293-296)
__global__ void __launch_bounds__(256)
k_gram_finish(const double *__restrict__ P, int ksplit, int npad, int nf, const double *__restrict__ pinv, double *__restrict__ H)
{
    const int n = nf + 1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    const int m = (int)(idx / n), c = (int)(idx % n);
    const int hi = max(m, c), lo = min(m, c);               // (row / column nf = the implicit ones column of the Gram build)
    double v = 0.0;
    for (int k = 0; k < ksplit; k++) v += P[((int64_t)k * npad + hi) * npad + lo];
    if (m == c) v = pinv[m] + v;
    H[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// consensus
// ------------------------------------------------------------------------------------------------
// partial means of this shard, sequential over local partitions in add order:
// acc = 1.0*acc + (1/N)*f32 (consumers/MeanLinearModelConsumer.java:61, models/LinearModel.java:181-201)
__global__ void __launch_bounds__(256)
k_partial_means(int nlocal, int n_lambda, int n_global, double invN, const float *__restrict__ B,
                const float *__restrict__ u, double *__restrict__ xbar, double *__restrict__ ubar)
{
#pragma clang fp contract(off)
    const int64_t tot = (int64_t)n_lambda * n_global;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
        double xb = 0.0, ub = 0.0;
        for (int k = 0; k < nlocal; k++) {
            const int64_t o = (int64_t)k * tot + i;
            xb = 1.0 * xb + invN * (double)B[o];
            ub = 1.0 * ub + invN * (double)u[o];
        }
        xbar[i] = xb;
        ubar[i] = ub;
    }
}

// z-update + per-lambda ||z - z_prev||_inf (jobs/RegressionAdmmTrain.java:365-405 L2, :406-451 L1, :455-472)
__global__ void __launch_bounds__(256)
k_z_update(int n_lambda, int n_global, int regularizer, int penalize_intercept, const double *__restrict__ weight_l,
           const double *__restrict__ cmap /* [n_lambda][n_global] per-feature weight or nullptr */,
           const double *__restrict__ xbar, const double *__restrict__ ubar, double *__restrict__ Z,
           float *__restrict__ z32, unsigned long long *__restrict__ diffbits)
{
#pragma clang fp contract(off)
    __shared__ double scratch[16];
    const int li = blockIdx.y;
    const double weight = weight_l[li];
    double mx = 0.0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n_global; j += gridDim.x * 256) {
        const int64_t i = (int64_t)li * n_global + j;
        const double xb = xbar[i], ub = ubar[i];
        double zn;
        const bool icpt = (j == n_global - 1);
        if (icpt && !penalize_intercept) {
            zn = xb + ub;
        } else if (regularizer == 2) {
            const double c = (cmap && !icpt) ? cmap[i] : weight;
            zn = 0.0 + c * xb;
            zn = 1.0 * zn + c * ub;
        } else {
            zn = 0.0 + 1.0 * xb;
            zn = 1.0 * zn + 1.0 * ub;
            if (!icpt) {                                   // iterative thresholding :424-436 (coefficients only)
                if (zn > weight) zn = zn - weight;
                else if (zn < -weight) zn = zn + weight;
            }
        }
        const double dv = fabs(1.0 * Z[i] + -1.0 * zn);
        mx = fmax(mx, dv);
        Z[i] = zn;
        z32[i] = (float)zn;
    }
    mx = block_allreduce_max(mx, scratch);
    if (threadIdx.x == 0) atomicMax(&diffbits[li], (unsigned long long)__double_as_longlong(mx));
}

// u_k = f32( f32(u_k + beta_k) - Z ), Z in double (computeU, jobs/RegressionAdmmTrain.java:752-757)
__global__ void __launch_bounds__(256)
k_u_update(int nlocal, int n_lambda, int n_global, const float *__restrict__ UPX, const double *__restrict__ Z,
           float *__restrict__ u)
{
#pragma clang fp contract(off)
    const int64_t tot = (int64_t)n_lambda * n_global;
    const int64_t all = tot * nlocal;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < all; i += (int64_t)gridDim.x * 256) {
        const int64_t zi = i % tot;
        u[i] = (float)(1.0 * (double)UPX[i] + -1.0 * Z[zi]);
    }
}

// ------------------------------------------------------------------------------------------------
// test log-likelihood of the consensus model z (jobs/RegressionAdmmTrain.java:766-811 ->
// models/LinearModel.java:491-554, eval :241-257 with num_click_replicates = 1): per row
//   xbeta = offset + ( -log(0 + 1*exp(-intercept)) + sum_j z[j] * x_j ),
//   ll = -log1p(exp(-xbeta)) * weight  if y == 1 else  -log1p(exp(xbeta)) * weight.
// 8 lanes per row (global feature ids, -1 = feature not in the model), every lambda in one pass.
// ------------------------------------------------------------------------------------------------
template <bool HASVAL>
__global__ void __launch_bounds__(256)
k_test_loglik(int l, int n_lambda, int n_global, const int64_t *__restrict__ rp, const int32_t *__restrict__ gi,
              const double *__restrict__ val, const int8_t *__restrict__ y, const double *__restrict__ wt,
              const double *__restrict__ off, const double *__restrict__ Z, double *__restrict__ part)
{
    __shared__ double scratch[48];
    constexpr int G = 8, GPB = 256 / G;
    const int gid = threadIdx.x / G, gl = threadIdx.x % G;
    const int row = blockIdx.x * GPB + gid;
    const bool valid = row < l;
    const int rowc = min(row, l - 1);
    const int64_t k0 = rp[rowc], k1 = valid ? rp[rowc + 1] : k0;
    for (int li = 0; li < n_lambda; li++) {
        const double *__restrict__ z = Z + (int64_t)li * n_global;
        double a = 0.0;
        for (int64_t k = k0 + gl; k < k1; k += G) {
            const int g = gi[k];
            if (g >= 0) a += z[g] * (HASVAL ? val[k] : 1.0);
        }
        a = group_allreduce_sum<G>(a);
        double ll[1] = {0.0};
        if (valid && gl == 0) {
            const double base = -log(0.0 + 1.0 * exp(-z[n_global - 1]));
            const double xbeta = off[row] + (base + a);
            ll[0] = (y[row] == 1) ? -log1p(exp(-xbeta)) * wt[row] : -log1p(exp(xbeta)) * wt[row];
        }
        block_allreduce_sum<1>(ll, scratch);
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * n_lambda + li] = ll[0];
    }
}

__global__ void k_round_z(int64_t n, const double *__restrict__ Z, float *__restrict__ z32)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) z32[i] = (float)Z[i];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define LAUNCH_DENSE(NV, U)                                                                                               \
    do {                                                                                                                  \
        if (stream_once) hipLaunchKernelGGL((k_xpass_dense<NV, U, true>), dim3(maxblk, nq), dim3(256),                    \
                                            (4 * NV * 256 + 16) * sizeof(double), st, parts, probs, qlist);               \
        else hipLaunchKernelGGL((k_xpass_dense<NV, U, false>), dim3(maxblk, nq), dim3(256),                               \
                                (4 * NV * 256 + 16) * sizeof(double), st, parts, probs, qlist);                           \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT DEVICE: with one handle + one thread per device in a
// process (the CLI's gpus=0,1,..; a JVM host) every device needs its own call, and two threads must not race on the flag.
// fn runs once per (launcher slot, device), under a lock; a failing call is reported (the launch that follows then fails loudly).
template <typename F>
static void per_device_once(int slot, F &&fn)
{
    static std::mutex mu;
    static bool done[10][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { fn(); return; }
    std::lock_guard<std::mutex> lk(mu);
    if (done[slot][dev]) return;
    fn();
    done[slot][dev] = true;
}
#include "mlx_ro_kernels.h"

static void set_max_lds(const void *func, int bytes)
{
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) fprintf(stderr, "[mlease_hip] hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed: %s\n", bytes, hipGetErrorString(e));
}
#include "mlx_ro_dense.h"

int mlxk_xpass_dense(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int maxblk,
                     int max_nfeat, bool stream_once)
{
    if (nq <= 0) return 0;
    if (max_nfeat <= 256) LAUNCH_DENSE(1, 8);
    else if (max_nfeat <= 512) LAUNCH_DENSE(2, 4);
    else if (max_nfeat <= 1024) LAUNCH_DENSE(4, 4);
    else if (max_nfeat <= 2048) {
        per_device_once(0, [&] {
            set_max_lds(reinterpret_cast<const void *>(&k_xpass_dense<8, 2, true>), (4 * 8 * 256 + 16) * (int)sizeof(double));
            set_max_lds(reinterpret_cast<const void *>(&k_xpass_dense<8, 2, false>), (4 * 8 * 256 + 16) * (int)sizeof(double));
        });
        LAUNCH_DENSE(8, 2);
    }
    else return -1;
    return 0;
}

#define XGRID(nq, gx) ((unsigned)(((nq) + 7) / 8 * 8) * (unsigned)(gx))

template <int G>
static void launch_rowpass(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq,
                           int maxblk, bool hasval)
{
    if (hasval) hipLaunchKernelGGL((k_rowpass_csr<G, true>), dim3(XGRID(nq, maxblk)), dim3(256), 0, st, parts, probs, qlist, nq, maxblk);
    else hipLaunchKernelGGL((k_rowpass_csr<G, false>), dim3(XGRID(nq, maxblk)), dim3(256), 0, st, parts, probs, qlist, nq, maxblk);
}

int mlxk_xpass_csr(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int maxblk,
                   int max_short, int max_long, int rowgroup, bool hasval, bool sell, int max_cunits, int max_rblk_rows, int row_slw, int row_ngc, bool stream_once, int which,
                   int cold_groups, int ro_blocks, int ro_units_blk, int *coldone)
{
    const bool do_row = which & 1, do_col = which & 2;
    if (nq <= 0) return 0;
    if (sell && ro_blocks > 0) {
        // reference-order numerics (mlx_ro_kernels.h): every slice of the row pass hot, the column pass once per row block
        const size_t lds_col = std::max(((size_t)max_rblk_rows + 1) * sizeof(double), (size_t)RO_CSUM_WAVES * 2 * RO_CSUM_CH * sizeof(double));
        const size_t lds_row = ((size_t)row_slw + 1) * sizeof(double);
        per_device_once(6, [&] {
#define SETLDS_RO(HV)                                                                                                                          \
            set_max_lds(reinterpret_cast<const void *>(&k_colpass_lds<HV, false, true>), 160 * 1024 - 1024);   /* (+ 528 bytes of static LDS: the relay) */ \
            set_max_lds(reinterpret_cast<const void *>(&k_rowpass_lds<HV, false, 1, true>), 160 * 1024 - 512); \
            set_max_lds(reinterpret_cast<const void *>(&k_rowpass_lds<HV, false, 2, true>), 160 * 1024 - 512); \
            set_max_lds(reinterpret_cast<const void *>(&k_rowpass_lds<HV, false, 4, true>), 160 * 1024 - 512); \
            set_max_lds(reinterpret_cast<const void *>(&k_rowpass_lds<HV, false, 8, true>), 160 * 1024 - 512)
            SETLDS_RO(true); SETLDS_RO(false);
#undef SETLDS_RO
        });
        // (one block's launch holds that block's units only: with all of a partition's units in the grid half of the 1 024-thread,
        //  157 KB workgroups of a launch were placed just to find out that their block was not this launch's)
        const int gxb = ro_units_blk > 0 ? ro_units_blk : max_cunits;
#define LAUNCH_ROW_RO(HV, GP) hipLaunchKernelGGL((k_rowpass_lds<HV, false, GP, true>), dim3(XGRID(nq, maxblk)), dim3(1024), lds_row, st, parts, probs, qlist, nq, maxblk, 0, coldone)
#define LAUNCH_RO(HV)                                                                                                                          \
        do {                                                                                                                                   \
            if (do_row) switch (row_ngc) {                                                                                                     \
                case 16: LAUNCH_ROW_RO(HV, 1); break;                                                                                          \
                case 32: LAUNCH_ROW_RO(HV, 2); break;                                                                                          \
                case 64: LAUNCH_ROW_RO(HV, 4); break;                                                                                          \
                default: LAUNCH_ROW_RO(HV, 8); break;                                                                                          \
            }                                                                                                                                  \
            if (do_col && max_cunits > 0 && coldone != nullptr && ro_blocks > 1) {                                                              \
                /* all row blocks in ONE launch (the units of a later block wait for their problem's earlier units inside it) */                \
                const int lead = ((nq + RO_CSUM_WAVES - 1) / RO_CSUM_WAVES + 7) / 8 * 8;                                                        \
                hipLaunchKernelGGL((k_colpass_lds<HV, false, true>), dim3(XGRID(nq, gxb) * ro_blocks + lead), dim3(1024), lds_col, st, parts, probs, qlist, nq, gxb, -1, lead, coldone); \
            } else if (do_col && max_cunits > 0)                                                                                               \
                for (int b = 0; b < ro_blocks; b++) {                                                                                          \
                    /* (the chain workgroups of the intercept's column lead the first launch: a multiple of 8, so the XCD mapping holds) */   \
                    const int lead = b == 0 ? ((nq + RO_CSUM_WAVES - 1) / RO_CSUM_WAVES + 7) / 8 * 8 : 0;                                      \
                    hipLaunchKernelGGL((k_colpass_lds<HV, false, true>), dim3(XGRID(nq, gxb) + lead), dim3(1024), lds_col, st, parts, probs, qlist, nq, gxb, b, lead, (int *)nullptr); \
                }                                                                                                                              \
        } while (0)
        if (hasval) LAUNCH_RO(true); else LAUNCH_RO(false);
#undef LAUNCH_RO
#undef LAUNCH_ROW_RO
        return 0;
    }
    if (sell) {
        const size_t lds_col = ((size_t)max_rblk_rows + 1) * sizeof(double), lds_row = ((size_t)row_slw + 1) * sizeof(double);
        per_device_once(1, [&] {
#define SETLDS(HV, NTF)                                                                                                                        \
            set_max_lds(reinterpret_cast<const void *>(&k_colpass_lds<HV, NTF>), 160 * 1024 - 64); \
            set_max_lds(reinterpret_cast<const void *>(&k_rowpass_lds<HV, NTF, 1>), 160 * 1024 - 512); \
            set_max_lds(reinterpret_cast<const void *>(&k_rowpass_lds<HV, NTF, 2>), 160 * 1024 - 512); \
            set_max_lds(reinterpret_cast<const void *>(&k_rowpass_lds<HV, NTF, 4>), 160 * 1024 - 512); \
            set_max_lds(reinterpret_cast<const void *>(&k_rowpass_lds<HV, NTF, 8>), 160 * 1024 - 512)
            SETLDS(true, true); SETLDS(true, false); SETLDS(false, true); SETLDS(false, false);
#undef SETLDS
        });
#define LAUNCH_ROW(HV, NTF, GP) hipLaunchKernelGGL((k_rowpass_lds<HV, NTF, GP>), dim3(XGRID(nq, maxblk)), dim3(1024), lds_row, st, parts, probs, qlist, nq, maxblk, cold_groups > 0 ? 1 : 0)
#define LAUNCH_SELL(HV, NTF)                                                                                                                   \
        do {                                                                                                                                   \
            if (do_row && cold_groups > 0)                                                                                                     \
                hipLaunchKernelGGL((k_rowcold<HV, NTF>), dim3(XGRID(nq, (cold_groups + 15) / 16)), dim3(256), 0, st, parts, probs, qlist, nq, (cold_groups + 15) / 16); \
            if (do_row) switch (row_ngc) {                                                                                                     \
                case 16: LAUNCH_ROW(HV, NTF, 1); break;                                                                                        \
                case 32: LAUNCH_ROW(HV, NTF, 2); break;                                                                                        \
                case 64: LAUNCH_ROW(HV, NTF, 4); break;                                                                                        \
                default: LAUNCH_ROW(HV, NTF, 8); break;                                                                                        \
            }                                                                                                                                  \
            if (do_col && max_cunits > 0)                                                                                                      \
                hipLaunchKernelGGL((k_colpass_lds<HV, NTF>), dim3(XGRID(nq, max_cunits)), dim3(1024), lds_col, st, parts, probs, qlist, nq, max_cunits, -1, 0); \
        } while (0)
        if (hasval) { if (stream_once) LAUNCH_SELL(true, true); else LAUNCH_SELL(true, false); }
        else { if (stream_once) LAUNCH_SELL(false, true); else LAUNCH_SELL(false, false); }
#undef LAUNCH_SELL
#undef LAUNCH_ROW
        return 0;
    }
    if (do_row) switch (rowgroup) {
    case 8: launch_rowpass<8>(st, parts, probs, qlist, nq, maxblk, hasval); break;
    case 16: launch_rowpass<16>(st, parts, probs, qlist, nq, maxblk, hasval); break;
    case 32: launch_rowpass<32>(st, parts, probs, qlist, nq, maxblk, hasval); break;
    default: launch_rowpass<64>(st, parts, probs, qlist, nq, maxblk, hasval); break;
    }
    const int gs = (max_short + 31) / 32, gl = (max_long + 3) / 4;
    if (do_col && gl > 0) {
        if (hasval) hipLaunchKernelGGL((k_colpass_items<64, true>), dim3(XGRID(nq, gl)), dim3(256), 0, st, parts, probs, qlist, nq, gl);
        else hipLaunchKernelGGL((k_colpass_items<64, false>), dim3(XGRID(nq, gl)), dim3(256), 0, st, parts, probs, qlist, nq, gl);
    }
    if (do_col && gs > 0) {
        if (hasval) hipLaunchKernelGGL((k_colpass_items<8, true>), dim3(XGRID(nq, gs)), dim3(256), 0, st, parts, probs, qlist, nq, gs);
        else hipLaunchKernelGGL((k_colpass_items<8, false>), dim3(XGRID(nq, gs)), dim3(256), 0, st, parts, probs, qlist, nq, gs);
    }
    return 0;
}

void mlxk_tron_step(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int threads,
                    int *done_counter)
{
    if (nq > 0) hipLaunchKernelGGL(k_tron_step, dim3(nq), dim3(threads), 0, st, parts, probs, qlist, nq, done_counter);
}

void mlxk_step_phase(hipStream_t st, int which, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int ch,
                     int max_nwg, int *done_counter, bool emu)
{
    if (nq <= 0) return;
    const dim3 grid((unsigned)max_nwg, (unsigned)nq);
    if (which == 0) {
        if (emu) hipLaunchKernelGGL(k_step_head, dim3((unsigned)nq), dim3(1024), 0, st, parts, probs, qlist);
        if (emu) hipLaunchKernelGGL(k_step_a<true>, grid, dim3(STEP_T), 0, st, parts, probs, qlist, ch);
        else hipLaunchKernelGGL(k_step_a<false>, grid, dim3(STEP_T), 0, st, parts, probs, qlist, ch);
    } else if (which == 1) {
        if (emu) hipLaunchKernelGGL(k_step_b<true>, grid, dim3(STEP_T), 0, st, parts, probs, qlist, ch);
        else hipLaunchKernelGGL(k_step_b<false>, grid, dim3(STEP_T), 0, st, parts, probs, qlist, ch);
    }
    else if (which == 2) hipLaunchKernelGGL(k_step_c, grid, dim3(STEP_T), 0, st, parts, probs, qlist, ch);
    else hipLaunchKernelGGL(k_step_commit, dim3((unsigned)nq), dim3(STEP_T), 0, st, parts, probs, qlist, ch, done_counter);
}

void mlxk_solve_small(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nprob, bool hasval,
                      int max_ticks, int *done_counter, int lds_doubles, bool faithful, int xl, int lds_bytes_xl, bool grid_dots)
{
    if (nprob <= 0) return;
    static const int wave_step = getenv("MLX_SMALL_WAVE_STEP") ? atoi(getenv("MLX_SMALL_WAVE_STEP")) : 1;     // A/B switch
    if (xl != 0 && lds_doubles > 0 && !faithful) {
        // vectors AND the partition's arrays in LDS (lds_bytes_xl = what the largest problem needs)
        per_device_once(5, [&] {
            set_max_lds(reinterpret_cast<const void *>(&k_solve_small<true, true, false, 1>), 150 * 1024);
            set_max_lds(reinterpret_cast<const void *>(&k_solve_small<false, true, false, 1>), 150 * 1024);
            set_max_lds(reinterpret_cast<const void *>(&k_solve_small<true, true, false, 2>), 150 * 1024);
            set_max_lds(reinterpret_cast<const void *>(&k_solve_small<false, true, false, 2>), 150 * 1024);
        });
#define LSMALL(HV, X) hipLaunchKernelGGL((k_solve_small<HV, true, false, X>), dim3(nprob), dim3(1024), (size_t)lds_bytes_xl, st, parts, probs, qlist, nprob, max_ticks, done_counter, wave_step, grid_dots ? 1 : 0)
        if (hasval) { if (xl == 1) LSMALL(true, 1); else LSMALL(true, 2); }
        else { if (xl == 1) LSMALL(false, 1); else LSMALL(false, 2); }
#undef LSMALL
        return;
    }
    if (faithful) {
        if (hasval) hipLaunchKernelGGL((k_solve_small<true, false, true>), dim3(nprob), dim3(1024), 0, st, parts, probs, qlist, nprob, max_ticks, done_counter, wave_step, grid_dots ? 1 : 0);
        else hipLaunchKernelGGL((k_solve_small<false, false, true>), dim3(nprob), dim3(1024), 0, st, parts, probs, qlist, nprob, max_ticks, done_counter, wave_step, grid_dots ? 1 : 0);
        return;
    }
    // lds_doubles > 0: the work vectors of every problem fit in LDS (that many doubles for the largest) -> LDS-resident solve
    if (lds_doubles > 0) {
        per_device_once(3, [&] {
            set_max_lds(reinterpret_cast<const void *>(&k_solve_small<true, true, false>), 150 * 1024);
            set_max_lds(reinterpret_cast<const void *>(&k_solve_small<false, true, false>), 150 * 1024);
        });
        const size_t bytes = (size_t)lds_doubles * sizeof(double);
        if (hasval) hipLaunchKernelGGL((k_solve_small<true, true, false>), dim3(nprob), dim3(1024), bytes, st, parts, probs, qlist, nprob, max_ticks, done_counter, wave_step, grid_dots ? 1 : 0);
        else hipLaunchKernelGGL((k_solve_small<false, true, false>), dim3(nprob), dim3(1024), bytes, st, parts, probs, qlist, nprob, max_ticks, done_counter, wave_step, grid_dots ? 1 : 0);
        return;
    }
    if (hasval) hipLaunchKernelGGL((k_solve_small<true, false, false>), dim3(nprob), dim3(1024), 0, st, parts, probs, qlist, nprob, max_ticks, done_counter, wave_step, grid_dots ? 1 : 0);
    else hipLaunchKernelGGL((k_solve_small<false, false, false>), dim3(nprob), dim3(1024), 0, st, parts, probs, qlist, nprob, max_ticks, done_counter, wave_step, grid_dots ? 1 : 0);
}

void mlxk_collect_c0(hipStream_t st, const PartDev *parts, const ProbDev *probs, const int *qlist, int nq,
                     double *const *c0_ptrs, bool ro)
{
    if (nq <= 0) return;
    if (ro) hipLaunchKernelGGL(k_ro_collect_c0, dim3(nq), dim3(256), 0, st, parts, probs, qlist, c0_ptrs);
    else hipLaunchKernelGGL(k_collect_c0, dim3(nq), dim3(256), 0, st, parts, probs, qlist, c0_ptrs);
}

void mlxk_ro_step(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int *done_counter, bool exact_norms)
{
    if (nq <= 0) return;
    per_device_once(7, [&] { set_max_lds(reinterpret_cast<const void *>(&k_ro_step), (int)sizeof(RoLds)); });
    hipLaunchKernelGGL(k_ro_step, dim3((unsigned)nq), dim3(RO_T), sizeof(RoLds), st, parts, probs, qlist, nq, done_counter, exact_norms ? 1 : 0);
}

void mlxk_setup(hipStream_t st, const PartDev *parts, ProbDev *probs, int nprob, int n_lambda, int n_global,
                int max_nlocal, const float *z32, const float *u, const double *pinv_l, double epsilon, int max_iter)
{
    const int gx = max(1, min(64, (max_nlocal + 255) / 256));
    hipLaunchKernelGGL(k_setup, dim3(gx, nprob), dim3(256), 0, st, parts, probs, nprob, n_lambda, n_global, z32, u,
                       pinv_l, epsilon, max_iter);
}

void mlxk_setup_naive(hipStream_t st, const PartDev *parts, ProbDev *probs, int nprob, int max_nlocal,
                      const double *pinv_l, const double *pinv_ovr, double *pinv_buf, double prior_mean,
                      double epsilon, int max_iter)
{
    const int gx = max(1, min(64, (max_nlocal + 255) / 256));
    hipLaunchKernelGGL(k_setup_naive, dim3(gx, nprob), dim3(256), 0, st, parts, probs, nprob, max_nlocal, pinv_l,
                       pinv_ovr, pinv_buf, prior_mean, epsilon, max_iter);
}

void mlxk_outputs_naive(hipStream_t st, const PartDev *parts, const ProbDev *probs, int nprob, int n_lambda,
                        int n_global, int max_nlocal, float *B)
{
    const int gx = max(1, min(64, (max_nlocal + 255) / 256));
    hipLaunchKernelGGL(k_outputs_naive, dim3(gx, nprob), dim3(256), 0, st, parts, probs, nprob, n_lambda, n_global, B);
}

void mlxk_outputs(hipStream_t st, const PartDev *parts, const ProbDev *probs, int nprob, int n_lambda, int n_global,
                  int max_nlocal, bool any_absent, const float *z32, const float *u, float *B, float *UPX)
{
    if (any_absent) {
        const int gx = max(1, min(64, (n_global + 255) / 256));
        hipLaunchKernelGGL(k_outputs_absent, dim3(gx, nprob), dim3(256), 0, st, probs, nprob, n_lambda, n_global, z32,
                           u, B, UPX);
    }
    const int gx = max(1, min(64, (max_nlocal + 255) / 256));
    hipLaunchKernelGGL(k_outputs_present, dim3(gx, nprob), dim3(256), 0, st, parts, probs, nprob, n_lambda, n_global,
                       u, B, UPX);
}

void mlxk_partial_means(hipStream_t st, int nlocal, int n_lambda, int n_global, double invN, const float *B,
                        const float *u, double *xbar, double *ubar)
{
    const int64_t tot = (int64_t)n_lambda * n_global;
    const int gx = (int)max((int64_t)1, min((int64_t)2048, (tot + 255) / 256));
    hipLaunchKernelGGL(k_partial_means, dim3(gx), dim3(256), 0, st, nlocal, n_lambda, n_global, invN, B, u, xbar, ubar);
}

void mlxk_z_update(hipStream_t st, int n_lambda, int n_global, int regularizer, int penalize_intercept,
                   const double *weight_l, const double *cmap, const double *xbar, const double *ubar, double *Z,
                   float *z32, unsigned long long *diffbits)
{
    const int gx = max(1, min(256, (n_global + 255) / 256));
    hipLaunchKernelGGL(k_z_update, dim3(gx, n_lambda), dim3(256), 0, st, n_lambda, n_global, regularizer,
                       penalize_intercept, weight_l, cmap, xbar, ubar, Z, z32, diffbits);
}

void mlxk_u_update(hipStream_t st, int nlocal, int n_lambda, int n_global, const float *UPX, const double *Z, float *u)
{
    const int64_t all = (int64_t)nlocal * n_lambda * n_global;
    const int gx = (int)max((int64_t)1, min((int64_t)4096, (all + 255) / 256));
    hipLaunchKernelGGL(k_u_update, dim3(gx), dim3(256), 0, st, nlocal, n_lambda, n_global, UPX, Z, u);
}

void mlxk_test_loglik(hipStream_t st, int l, int n_lambda, int n_global, const int64_t *rp, const int32_t *gi,
                      const double *val, const int8_t *y, const double *wt, const double *off, const double *Z, double *part)
{
    const int gx = (l + 31) / 32;
    if (val) hipLaunchKernelGGL((k_test_loglik<true>), dim3(gx), dim3(256), 0, st, l, n_lambda, n_global, rp, gi, val, y, wt, off, Z, part);
    else hipLaunchKernelGGL((k_test_loglik<false>), dim3(gx), dim3(256), 0, st, l, n_lambda, n_global, rp, gi, val, y, wt, off, Z, part);
}

// ---- one wave that does nothing for `ticks` of the constant-rate wall clock. mlx_api.hip launches one on each of two HIP streams to
// find out whether the runtime put them on ONE hardware queue (then the second starts when the first ends): pick_tick_streams().
__global__ void __launch_bounds__(64)
k_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void mlxk_spin(hipStream_t st, long long ticks) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks); }

void mlxk_round_z(hipStream_t st, int64_t n, const double *Z, float *z32)
{
    const int gx = (int)max((int64_t)1, min((int64_t)1024, (n + 255) / 256));
    hipLaunchKernelGGL(k_round_z, dim3(gx), dim3(256), 0, st, n, Z, z32);
}

void mlxk_densify(hipStream_t st, int l, const int32_t *rp, const int32_t *ci, const float *val, float *X, int64_t ld)
{
    const int gx = (l + 255) / 256;
    if (val) hipLaunchKernelGGL((k_densify<true>), dim3(gx), dim3(256), 0, st, l, rp, ci, val, X, ld);
    else hipLaunchKernelGGL((k_densify<false>), dim3(gx), dim3(256), 0, st, l, rp, ci, val, X, ld);
}

void mlxk_hess_colsums(hipStream_t st, const float *X, int64_t ld, int l, const double *wd, double *part, int nchunk,
                       int rows_per_chunk, double *out)
{
    const int gx = (int)((ld + 255) / 256);
    hipLaunchKernelGGL(k_hess_colsums, dim3(gx, nchunk), dim3(256), 0, st, X, ld, l, rows_per_chunk, wd, part);
    hipLaunchKernelGGL(k_hess_colsums_reduce, dim3(gx), dim3(256), 0, st, part, nchunk, ld, wd, l, out);
}

void mlxk_hess_diag_items(hipStream_t st, int n_items, const int32_t *item_ptr, const int32_t *item_dst, const int32_t *cri,
                          const float *cval, const double *wd, double *slots)
{
    const int gx = (n_items + 255) / 256;
    if (gx <= 0) return;
    if (cval) hipLaunchKernelGGL((k_hess_diag_items<true>), dim3(gx), dim3(256), 0, st, n_items, item_ptr, item_dst, cri, cval, wd, slots);
    else hipLaunchKernelGGL((k_hess_diag_items<false>), dim3(gx), dim3(256), 0, st, n_items, item_ptr, item_dst, cri, cval, wd, slots);
}

void mlxk_gram_f64(hipStream_t st, const float *X, int64_t ld, int l, const double *wd, const int *blocks_xy, int nblocks,
                   int ksplit, int rows_per_split, double *P, int npad, int nf)
{
    per_device_once(8, [&] { set_max_lds(reinterpret_cast<const void *>(&k_gram_f64<4>), 4 * 4096 * (int)sizeof(double)); });
    hipLaunchKernelGGL((k_gram_f64<4>), dim3(nblocks, ksplit / 2), dim3(512), 4 * 4096 * sizeof(double), st, X, ld, l, wd,
                       reinterpret_cast<const int2 *>(blocks_xy), rows_per_split, P, npad, nf);
}

void mlxk_gram_finish(hipStream_t st, const double *P, int ksplit, int npad, int nf, const double *pinv, double *H)
{
    const int64_t tot = (int64_t)(nf + 1) * (nf + 1);
    hipLaunchKernelGGL(k_gram_finish, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, P, ksplit, npad, nf, pinv, H);
}

void mlxk_score_rows(hipStream_t st, int l, const int64_t *rp, const int32_t *gi, const double *val, const double *off,
                     const double *z, double base, float *pred)
{
    const int gx = (l + 255) / 256;
    if (val) hipLaunchKernelGGL((k_score_rows<true>), dim3(gx), dim3(256), 0, st, l, rp, gi, val, off, z, base, pred);
    else hipLaunchKernelGGL((k_score_rows<false>), dim3(gx), dim3(256), 0, st, l, rp, gi, val, off, z, base, pred);
}
