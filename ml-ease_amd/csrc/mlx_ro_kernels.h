// mlx_ro_kernels.h -- REFERENCE-ORDER numerics (MLX_NUMERICS_REFERENCE_ORDER) on the tick kernels; included by mlx_kernels.hip.
//
// The product path differs from the reference in ONE respect: reductions are trees (or grid-rounded sums) where the Java code runs
// sequential loops. This mode removes that difference at tick-kernel speed instead of the one-thread-per-reduction verification kernel
// (k_solve_small<.., SEQ>, which stays as the independent cross-check and for partitions that cannot be sliced):
//   * library column ids = the caller's (first-seen) ids, library rows = the caller's rows (mlx_api.hip: prep_csr);
//   * row sums (Xv, llf/LogisticRegressionL2.java:115-129): k_rowpass_lds<.., RO> with EVERY column slice staged in LDS one after the
//     other, one thread per row, one running sum over the row's entries in ascending column id, the bias entry last;
//   * column sums (XTv, :131-150): k_colpass_lds<.., RO>, one launch per row block in block order; an item = ALL entries of a column
//     inside the block, one thread per item, rows ascending, started from the sum the column reached in the earlier blocks
//     (handed on through PartDev::item_chain / item_init: the writer scatters, the reader loads its own slot) -- one chain per column
//     over all its rows; the longest slices run as a relay over the workgroup's 16 waves; the intercept's column (the sum of the row
//     coefficients in row order) is one more folding lane of the step's first pass;
//   * every n- or l-long dot / norm / loss sum of Tron.tron / trcg / fun (bw/Tron.java:30-252, llf/LogisticRegressionL2.java:156-193):
//     k_ro_step below -- one 640-thread workgroup per problem walks the vectors in chunks of 1024 elements; four staging waves do the
//     elementwise work of a chunk (the daxpy / scale statements, Hs[i] = s[i]*priorVar_inv[i] + Hs[i], the products a[i]*b[i] of
//     Tron.dot) and leave the chunk's TERMS in one of two LDS buffers; up to six folding WAVES meanwhile fold one term array each out
//     of the other buffer -- the loop of Tron.dot :204-213, `p += term` in index order, evaluated exactly WITHOUT its dependency chain
//     (mlx_seqfold.h, round 6: inside a binade of the running sum the chain equals the exact sum of grid-rounded terms, a scan; every
//     sub-block's prefix range is checked and a failed check re-runs that sub-block as the literal chain; round 5 folded literally,
//     one lane per array, 12.4 cycles per term). The six reductions a CG step needs at once run side by side.
//     euclideanNorm (:220-252) keeps a running scale: its update is `sum = 1 + sum*(scale/a)^2` when |v_i| exceeds the scale and
//     `sum += (a/scale)^2` otherwise. The scale before element i is the running maximum of |v| -- an exclusive prefix maximum, exact
//     in any association -- so the chunk's threads compute it by a scan and form every element's (m, c) with sum' = c + sum*m in
//     parallel (the divisions are off the fold); the ~ln n sub-blocks of a vector that hold a new maximum take that two-operation
//     form as a literal chain, all others are plain sums of c.
//   * exp / log1p: the portable forms of portable_math.h, as the oracle's verification twin evaluates them (device and host libm
//     differ in the last bit).
// Same statements in the same order as tron_step_body<SEQ> (which is bit-identical to the oracle): the tests run both.
#pragma once

#include "mlx_seqfold.h"

constexpr int RO_CH = 64 * SGF_K;                   // elements per chunk: one sub-block of SGF_K terms per lane of a folding wave
#ifndef RO_EPT
#define RO_EPT 2                                     // consecutive elements per staging thread and chunk (A/B: tools/ablate_build.sh -DRO_EPT=4)
#endif
constexpr int RO_E = RO_EPT;
#ifndef RO_STREAM_G
#define RO_STREAM_G 4                              // groups per thread and half-trip of the streamed d = beta d + r (A/B: tools/ablate_build.sh -DRO_STREAM_G=2)
#endif
// RO_NF folding waves + RO_NSW staging waves (64 RO_E elements each). The staging waves are the critical path since the folds stopped
// being literal chains (a chunk: wait for the operands, the elementwise statements, a max-scan and a hand-shake between the staging
// waves for the norms' running scale, two divisions per element, the LDS writes): eight waves with two elements per thread halve it.
constexpr int RO_NSW = RO_CH / (64 * RO_E), RO_NF = 4, RO_NN = 2, RO_T = 64 * (RO_NF + RO_NSW);
static_assert((RO_E == 2 || RO_E == 4) && SGF_K % RO_E == 0 && RO_NSW >= 1 && RO_NSW <= 12, "a staging thread's elements lie inside one sub-block");
constexpr int RO_PITCH = 65, RO_CA = SGF_K * RO_PITCH;
struct RoLds {
    // terms of a chunk, one array per folding wave (norm arrays first), TRANSPOSED: element e = L * SGF_K + i of the chunk -- term i of
    // lane L's sub-block -- sits at [i * 65 + L]: a folding wave's load of term i is 64 consecutive doubles at a constant offset from
    // one base address, and the pitch of 65 keeps the four staging threads of a sub-block (rows i0 .. i0 + 3 of one column) and
    // their neighbours on different LDS banks when they write. Two buffers: folded / being staged.
    double C[2][RO_NF][RO_CA];
    double M[2][RO_NN][RO_CA];     // multipliers of the norm arrays (1.0 except where the running scale changes), same layout
    unsigned char hm[2][RO_NN][64];// sub-block L of a norm array holds a change of the running scale
    double wtot[RO_NN][RO_NSW];    // scan: the staging waves' maxima of the chunk being staged
    double res[RO_NF], mcfin[RO_NN];
    double red[RO_NF + RO_NSW];    // one partial per wave: side sums of the staging threads (ro_block_sum)
    volatile int seq[RO_NSW];      // chunk number (+ 1) each staging wave has published its maximum for
};
struct RoV4 { double v[RO_E]; };                    // RO_E consecutive elements of a vector

// RO_E consecutive elements j0 .. of a 32-byte aligned vector (the work vectors: 256-byte aligned pieces of the slab, padded so
// that the last quad stays inside the piece -- mlx_finalize's carve_size)
__device__ __forceinline__ RoV4 ro_ld4(const double *__restrict__ p, int j0)
{
    typedef double d2v_t __attribute__((ext_vector_type(2)));
    RoV4 r;
#pragma unroll
    for (int e = 0; e < RO_E; e += 2) {
        const d2v_t a = gld(reinterpret_cast<const d2v_t *>(p + j0 + e));
        r.v[e] = a.x; r.v[e + 1] = a.y;
    }
    return r;
}
__device__ __forceinline__ RoV4 ro_ld4c(const double *__restrict__ p, int j0, int n)       // ... of a vector of n elements, clamped to its last group
{
    return ro_ld4(p, min(j0, ((n - 1) / RO_E) * RO_E));
}
__device__ __forceinline__ RoV4 ro_ld4s(const double *__restrict__ p, int j0, int n)      // any alignment, clamped
{
    RoV4 r;
#pragma unroll
    for (int e = 0; e < RO_E; e++) r.v[e] = gld(p + min(j0 + e, n - 1));
    return r;
}
__device__ __forceinline__ void ro_st4(double *__restrict__ p, int j0, int n, const double (&x)[RO_E])
{
#pragma unroll
    for (int e = 0; e < RO_E; e++) if (j0 + e < n) gst(p + j0 + e, x[e]);
}

// inclusive / exclusive prefix MAXIMUM of non-negative values over the 64 lanes, on the VALU (the DPP row shifts and row broadcasts of
// mlx_seqfold.h's scan; lanes without a source read 0.0, the identity here) -- the __shfl_up form was 24 ds_bpermute per chunk
__device__ __forceinline__ double ro_wave_inclusive_max(double v)
{
    v = fmax(v, sgf_dpp_or_zero<0x111, 0xF>(v));
    v = fmax(v, sgf_dpp_or_zero<0x112, 0xF>(v));
    v = fmax(v, sgf_dpp_or_zero<0x114, 0xF>(v));
    v = fmax(v, sgf_dpp_or_zero<0x118, 0xF>(v));
    v = fmax(v, sgf_dpp_or_zero<0x142, 0xA>(v));      // row_bcast:15 into rows 1 and 3
    v = fmax(v, sgf_dpp_or_zero<0x143, 0xC>(v));      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double ro_wave_shift_up_or_zero(double v) { return sgf_dpp_or_zero<0x138, 0xF>(v); }     // wave_shr:1 (lane 0: 0.0)

// Workgroup barrier that orders LDS accesses ONLY. __syncthreads() is a workgroup-scope fence over every address space: it waits for
// the wave's outstanding GLOBAL loads and stores too (s_waitcnt vmcnt(0)), so the operands a staging wave prefetches for the chunk after
// next would have to land before every chunk's barrier -- one memory latency per chunk, which round 5's literal folds (5 us per chunk)
// hid and the wave folds no longer do (step 1 635 -> see profiles/r6_notes.md).
__device__ __forceinline__ void ro_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One folding wave, one chunk: the wave's array out of LDS (term i of lane L's sub-block at cp[i * RO_PITCH + L]; nvalid = how many of
// this lane's SGF_K terms lie inside the vector, the others count as -0.0) into the running sum. NOT inlined: k_ro_step folds in seven
// passes, up to four arrays each -- as inlined copies they made the kernel 135 KB and left the register allocator spilling loop
// invariants of every copy into scratch, inside the chunk loop (profiles/r6_notes.md); two functions now, each with registers of its own.
typedef const __attribute__((address_space(3))) double *ro_lds_cptr;
template <bool MUL>
__device__ __attribute__((noinline)) double ro_fold_chunk(double acc, ro_lds_cptr cp, ro_lds_cptr mp, int nvalid, int has, int &hostile)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    double t[SGF_K], m[SGF_K];
#pragma unroll
    for (int i = 0; i < SGF_K; i++) {
        const double x = cp[RO_PITCH * i + lane];
        t[i] = i < nvalid ? x : -0.0;                          // behind the end: x + (-0.0) == x
        m[i] = 1.0;
    }
    if (MUL && __ballot(has != 0) != 0ull) {                   // (~ln n sub-blocks per vector hold a scale change)
#pragma unroll
        for (int i = 0; i < SGF_K; i++) { const double y = mp[RO_PITCH * i + lane]; m[i] = (has != 0 && i < nvalid) ? y : 1.0; }
    }
    return sgf_wave_fold<SGF_K, MUL>(acc, t, m, has != 0, hostile);
}

// The LITERAL chain over one chunk of an array staged LINEARLY (element e of the chunk at cp[e]): for sums the grid cannot help. One LDS
// read per lane fetches 64 consecutive terms (the next read is in flight behind the chain), and the chain walks the register lane by
// lane as scalars -- v_readlane with constant lane numbers (a run-time lane select costs ~70 cycles per term); every lane adds the same
// terms, nothing diverges: ~12 cycles per term.
__device__ __attribute__((noinline)) double ro_chain_chunk(double acc, ro_lds_cptr cp, int cnt)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int ntrip = (cnt + 63) >> 6;
    double nxt = lane < cnt ? cp[lane] : -0.0;                    // x + (-0.0) == x
    for (int tr = 0; tr < ntrip; tr++) {
        const double cur = nxt;
        const int j = (tr + 1) * 64 + lane;
        nxt = j < cnt ? cp[min(j, RO_CH - 1)] : -0.0;
#pragma unroll
        for (int L = 0; L < 64; L++) acc = acc + mlx_wave_bcast(cur, L);
    }
    return acc;
}

// Elementwise statements over elements 0 .. len-1 by all threads of the workgroup, no reductions: G groups of RO_E elements per thread
// and half-trip, two register sets used in turn -- the loads of the next half-trip are issued before the statements of the current one,
// so the vectors stream without a stop between trips (round 6's first form loaded four groups, waited, computed, stored, and only
// then issued the next loads: one memory latency per trip, 100 us per CG tick for the 3 vectors of d = beta d + r at configs[2]).
// load() clamps to the last group of each vector, emit() masks its stores: trips behind the end cost clamped re-reads only.
template <int G, typename R, typename LD, typename EM>
__device__ __forceinline__ void ro_stream(int len, LD load, EM emit)
{
    const int tid = threadIdx.x;
    constexpr int GS = RO_E * RO_T, HS = G * GS;               // elements per group sweep / per half-trip
    R a[G], b[G];
#pragma unroll
    for (int u = 0; u < G; u++) load(u * GS + RO_E * tid, a[u]);
    for (int base = 0; base < len; base += 2 * HS) {
#pragma unroll
        for (int u = 0; u < G; u++) load(base + HS + u * GS + RO_E * tid, b[u]);
#pragma unroll
        for (int u = 0; u < G; u++) emit(base + u * GS + RO_E * tid, a[u]);
#pragma unroll
        for (int u = 0; u < G; u++) load(base + 2 * HS + u * GS + RO_E * tid, a[u]);
#pragma unroll
        for (int u = 0; u < G; u++) emit(base + HS + u * GS + RO_E * tid, b[u]);
    }
}

// Sum of one value per thread over the workgroup (a fixed tree: wave butterflies, then the waves' sums in wave order). NOT one of the
// reference's sums: only for quantities that are compared against a threshold with a margin (the CG step's |s| and |r| tests below).
__device__ __forceinline__ double ro_block_sum(RoLds &sh, double v)
{
    v = mlx_wave_allreduce_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh.red[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < RO_NF + RO_NSW; w++) t += sh.red[w];
    __syncthreads();
    return t;
}

// One pass over elements 0 .. len-1. load(j0, R) fetches the operands of elements j0 .. j0+3 (it clamps to the last quad of each
// vector itself), emit(j0, R, ct, nv) does their elementwise work (stores included, elements beyond a vector's end masked) and
// returns the terms of the dot arrays ct[k][e] (k >= NN) and the raw values of the norm arrays nv[q][e] (q < NN). Folding wave k
// folds array k over its first lens[k] elements (lens == nullptr: len for all): norms from sum = 1, dots from init[k].
// result[k] (every thread): the norm scale*sqrt(sum) / the dot.
// Round 5 folded with one LANE per array -- a literal chain of dependent adds, 12.4 cycles per term; since round 6 a WAVE folds an array
// with sgf_wave_fold (mlx_seqfold.h): the same bits -- the sequential loop's -- from grid-rounded terms and a scan wherever the
// running sum stays inside a binade, the literal chain for the sub-blocks where it does not.
// LIN: bit k set = array k (a dot array) is staged linearly and folded as a literal chain (ro_chain_chunk).
template <int NF, int NN, int LIN, typename R, typename LD, typename EM>
__device__ __forceinline__ void ro_pass(RoLds &sh, int len, const int *lens, const double *init, double *result, LD load, EM emit)
{
#pragma clang fp contract(off)
    static_assert(NF <= RO_NF && NN <= RO_NN && NN <= NF, "fold arrays");
    constexpr int NFX = NF > 0 ? NF : 1, NNX = NN > 0 ? NN : 1;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (NF == 0) {
        // elementwise only: no terms, no barriers
        ro_stream<2, R>(len, load, [&](int j0, R &r) { double ct[NFX][RO_E], nv[NNX][RO_E]; emit(j0, r, ct, nv); });
        __syncthreads();
        return;
    }
    // Waves 0 .. RO_NF-1 FOLD (wave k array k; waves k >= NF only keep the barriers); waves RO_NF .. STAGE: while the folding waves work
    // on chunk c out of one LDS buffer, the staging waves compute chunk c + 1 into the other from operands they loaded one chunk earlier
    // (two register sets: the loads of chunk c + 2 are in flight meanwhile) -- memory latency, the scan, the divisions of the norms' terms
    // are off the folds' path: one barrier per chunk. The staging waves agree on the running maximum of a norm's operand among
    // themselves: each publishes its part's maximum in LDS and its chunk number behind it (sh.seq), and reads the others' when all are there.
    const bool folder = wave < RO_NF;
    const int sw = wave - RO_NF, stid = tid - 64 * RO_NF;     // staging wave 0..RO_NSW-1, staging thread 0..RO_CH/RO_E-1
    const int nch = (len + RO_CH - 1) / RO_CH;
    const int mylen = (folder && lens != nullptr) ? lens[wave < NF ? wave : 0] : len;      // (folding wave: the array it folds)
    double acc = 0.0;
    int hostile = 0;                                          // (a folding wave's array: did the last chunk run out of budget? mlx_seqfold.h)
    if (folder) acc = (wave < NN) ? 1.0 : init[wave < NF ? wave : 0];
    double mc[NNX];
#pragma unroll
    for (int q = 0; q < NNX; q++) mc[q] = 0.0;
#if defined(MLX_PHASE_TIMING) && !defined(MLX_PT_PASSES_ONLY)
    // (timing-experiment builds: shader-clock cycles per phase, kept in registers by lane 0 of every folding wave and of the first
    //  staging wave, added to g_phase[] once at the end of the pass -- no atomics inside the loop)
    unsigned long long pt_t = clock64(), pt_a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool pt_on = lane == 0 && (wave < NF || wave == RO_NF);
#define RO_PT(slot) do { if (pt_on) { const unsigned long long t_ = clock64(); pt_a[slot] += t_ - pt_t; pt_t = t_; } } while (0)
#else
#define RO_PT(slot) do { } while (0)
#endif
    // element e of the chunk (e = RO_E stid ..) -> term i = e % SGF_K of sub-block L = e / SGF_K, at [i * 65 + L]
    const int ti0 = (RO_E * stid) % SGF_K, tL = (RO_E * stid) / SGF_K;
    auto tpos = [&](int e) { return (ti0 + e) * RO_PITCH + tL; };
    auto stage = [&](int c, R &regs) {
#if defined(MLX_ABLATE) && (MLX_ABLATE & 64)
        if (NF == 2 && LIN == 2) { sh.C[c & 1][0][tpos(0)] = 1.0; sh.C[c & 1][0][tpos(1)] = 1.0; return; }
#endif
        const int b = c & 1;
        const int j0 = c * RO_CH + RO_E * stid;
        double ct[NFX][RO_E], nv[NNX][RO_E];
        RO_PT(2);
        emit(j0, regs, ct, nv);
        RO_PT(3);
        if (NN > 0) {
            // euclideanNorm's running scale in front of every element = exclusive prefix maximum of |v| (zeros never raise it)
            double a[NNX][RO_E], x[NNX];
            int flag[NNX];
#pragma unroll
            for (int q = 0; q < NN; q++) {
#pragma unroll
                for (int e = 0; e < RO_E; e++) a[q][e] = (j0 + e < (lens != nullptr ? lens[q] : len)) ? fabs(nv[q][e]) : 0.0;
                x[q] = a[q][0];
#pragma unroll
                for (int e = 1; e < RO_E; e++) x[q] = fmax(x[q], a[q][e]);
                x[q] = ro_wave_inclusive_max(x[q]);
                if (lane == 63) sh.wtot[q][sw] = x[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // (LDS only: the prefetched operands stay in flight)
            if (lane == 0) sh.seq[sw] = c + 1;
            // (lane w watches staging wave w: one LDS read per poll instead of RO_NSW dependent ones)
            while (__ballot(lane < RO_NSW && sh.seq[lane < RO_NSW ? lane : 0] < c + 1) != 0ull) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            RO_PT(4);
#pragma unroll
            for (int q = 0; q < NN; q++) {
                const double ex = ro_wave_shift_up_or_zero(x[q]);
                double P = fmax(mc[q], ex);
                const double tw = sh.wtot[q][lane < RO_NSW ? lane : 0];          // (lane w: staging wave w's maximum -- one LDS read, then scalars)
#pragma unroll
                for (int w = 0; w < RO_NSW; w++) {
                    const double t = mlx_wave_bcast(tw, w);
                    if (w < sw) P = fmax(P, t);
                    mc[q] = fmax(mc[q], t);
                }
                flag[q] = 0;
#pragma unroll
                for (int e = 0; e < RO_E; e++) {
                    const double ae = a[q][e];
                    double mm, cc;
                    if (!(ae != 0.0)) { mm = 1.0; cc = 0.0; }                       // v[i] == 0: skipped
                    else if (P < ae) { const double t = P / ae; mm = t * t; cc = 1.0; flag[q] = 1; }      // sum = 1 + sum * (t * t); scale = a
                    else { const double t = ae / P; mm = 1.0; cc = t * t; }           // sum += t * t
                    P = fmax(P, ae);
                    sh.C[b][q][tpos(e)] = cc;
                    sh.M[b][q][tpos(e)] = mm;
                }
                // the threads of a sub-block (SGF_K = 16 elements): does it hold a change of the running scale?
                const unsigned long long bal = __ballot(flag[q] != 0);
                constexpr int TPS = SGF_K / RO_E;              // threads per sub-block
                if ((lane % TPS) == 0) sh.hm[b][q][(RO_E * stid) / SGF_K] = ((bal >> (lane - lane % TPS)) & ((1ull << TPS) - 1ull)) ? 1 : 0;
            }
        }
        RO_PT(5);
#pragma unroll
        for (int k = NN; k < NF; k++) {
#pragma unroll
            for (int e = 0; e < RO_E; e++) sh.C[b][k][((LIN >> k) & 1) ? RO_E * stid + e : tpos(e)] = ct[k][e];
        }
    };
    auto fold = [&](int c) {
#if defined(MLX_ABLATE) && (MLX_ABLATE & 32)     /* timing experiments only (results are wrong): the CG step's first pass without its fold */
        if (NF == 2 && LIN == 2) return;
#endif
#if defined(MLX_ABLATE) && (MLX_ABLATE & 64)     /* timing experiments only: ... without its staging (terms = 1.0, no loads, no stores) */
#endif
        if (wave >= NF) return;
        const int b = c & 1;
        const int cnt = max(0, min(RO_CH, mylen - c * RO_CH));
        if (cnt == 0) return;                                   // (this array ends before the chunk: wave-uniform)
        const bool isn = wave < NN;
        const int has = (isn && sh.hm[b][isn ? wave : 0][lane] != 0) ? 1 : 0;
        const int nvalid = max(0, min(SGF_K, cnt - lane * SGF_K));
        ro_lds_cptr cp = (ro_lds_cptr)&sh.C[b][wave][0];
        ro_lds_cptr mp = (ro_lds_cptr)&sh.M[b][isn ? wave : 0][0];
        RO_PT(7);
        if ((LIN >> wave) & 1) acc = ro_chain_chunk(acc, cp, cnt);
        else if (isn) acc = ro_fold_chunk<true>(acc, cp, mp, nvalid, has, hostile);
        else acc = ro_fold_chunk<false>(acc, cp, mp, nvalid, 0, hostile);
    };
    if (NN > 0 && tid < RO_NSW) sh.seq[tid] = 0;
    __syncthreads();
    R ra, rb;
    if (!folder) {
        load(RO_E * stid, ra);
        load(RO_CH + RO_E * stid, rb);                            // (clamped inside load: a pass of one chunk reads its last quads again)
        stage(0, ra);
    }
    __syncthreads();
    for (int c = 0; c < nch; c += 2) {
        // chunk c is folded while chunk c + 1 is staged from rb and the operands of chunk c + 2 arrive in ra; then the roles swap
        if (!folder) { if (c + 1 < nch) { load((c + 2) * RO_CH + RO_E * stid, ra); stage(c + 1, rb); } }
        else fold(c);
        RO_PT(folder ? 0 : 2);
        ro_lds_barrier();
        RO_PT(folder ? 1 : 6);
#if defined(MLX_PHASE_TIMING) && !defined(MLX_PT_PASSES_ONLY)
        pt_a[7] += 100;
#endif
        if (c + 1 >= nch) break;
        if (!folder) { if (c + 2 < nch) { load((c + 3) * RO_CH + RO_E * stid, rb); stage(c + 2, ra); } }
        else fold(c + 1);
        RO_PT(folder ? 0 : 2);
        ro_lds_barrier();
        RO_PT(folder ? 1 : 6);
    }
#if defined(MLX_PHASE_TIMING) && !defined(MLX_SGF_STATS) && !defined(MLX_PT_PASSES_ONLY)
    if (pt_on) {
        if (folder) {
            atomicAdd(&g_phase[wave], pt_a[0]); atomicAdd(&g_phase[4 + wave], pt_a[1]);
            if (wave == 1) atomicAdd(&g_phase[14], pt_a[7]);
        }
        else { for (int i = 2; i <= 7; i++) atomicAdd(&g_phase[6 + i], pt_a[i]); }
    }
#endif
#undef RO_PT
    if (folder && wave < NF && lane == 0) sh.res[wave] = acc;
    if (NN > 0 && stid == 0) {
#pragma unroll
        for (int q = 0; q < NN; q++) sh.mcfin[q] = mc[q];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NF; k++) result[k] = (k < NN) ? sh.mcfin[k] * sqrt(sh.res[k]) : sh.res[k];
    __syncthreads();
}

// euclideanNorm(v) (bw/Tron.java:220-252) as a pass of its own: the CG step below decides `|s| > delta` and `|r| <= cgtol` from sums it
// has anyway and comes here only when such a sum is too close to its threshold to decide. Not inlined: rarely run, registers of its own.
__device__ __attribute__((noinline)) double ro_exact_norm(const double *__restrict__ v, int n)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char ro_smem[];
    RoLds &sh = *reinterpret_cast<RoLds *>(ro_smem);
    const double z[RO_NF] = {0.0, 0.0, 0.0, 0.0};
    double r[RO_NF];
    struct RN { RoV4 x; };
    ro_pass<1, 1, 0, RN>(sh, n, nullptr, z, r,
        [&](int j0, RN &R) { R.x = ro_ld4c(v, j0, n); },
        [&](int j0, RN &R, double (&ct)[1][RO_E], double (&nv)[1][RO_E]) {
#pragma unroll
            for (int e = 0; e < RO_E; e++) nv[0][e] = R.x.v[e];
        });
    return r[0];
}

// Can `norm > thr` (equivalently `norm <= thr`) be decided from ssq, a sum of the vector's squares in ANY order or the sequential dot
// v.v? euclideanNorm's n-step recurrence and either sum of squares lie within (n/2 + 25) 2^-53 of the true norm (relative; every
// term is non-negative), so they agree on the comparison whenever sqrt(ssq) is further than 8 (n + 64) 2^-53 from the threshold -- a
// margin of 8x over the two errors together. Sums outside the range where squares are exact to relative precision, infinite or NaN sums,
// and thresholds that are not finite are never decided here.
__device__ __forceinline__ bool ro_norm_decided(double ssq, double thr, int n)
{
    const double tol = (double)(n + 64) * 8.881784197001252e-16;       // 2^-50
    const double a = sqrt(ssq);
    return (ssq > 1e-280) && (ssq < 1e300) && (thr == thr) && (fabs(thr) < 1e300) && (fabs(a - thr) > tol * fabs(thr));
}

// (timing-experiment builds, tools/ablate_build.sh -DMLX_RO_PASS_TIMING: 100 MHz wall-clock ticks thread 0 of every workgroup spends in each
//  pass of k_ro_step -- [0] CG pass A, [1] CG pass B, [2] the boundary pass, [3] CG pass C, [4] EVAL row folds, [5] EVAL pass over n,
//  [6] EVAL copy pass, [8] CG ticks, [9] EVAL ticks; read back with mlx_debug_ropass_times())
#ifdef MLX_RO_PASS_TIMING
__device__ unsigned long long g_ropass[16];
extern "C" int mlx_debug_ropass_times(double *out16)
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ropass), sizeof h) != hipSuccess) return -1;
    for (int i = 0; i < 16; i++) out16[i] = (double)h[i];
    return 0;
}
#define ROP_INIT unsigned long long rop_last = wall_clock64()
#define ROP_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_ropass[k], t_ - rop_last); rop_last = t_; } } while (0)
#define ROP_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_ropass[k], 1ull); } while (0)
#else
#define ROP_INIT
#define ROP_MARK(k)
#define ROP_COUNT(k)
#endif

// The TRON/CG step of one tick, reference-order numerics: one workgroup per problem (bw/Tron.java:30-179; statement order of
// tron_step_body<SEQ>).
__global__ void __launch_bounds__(RO_T)
k_ro_step(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq,
          int *__restrict__ done_counter, int exact_norms /* 1: every norm test of a CG step runs euclideanNorm's recurrence (option "ro_exact_norms") */)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char ro_smem[];
    RoLds &sh = *reinterpret_cast<RoLds *>(ro_smem);
    if ((int)blockIdx.x >= nq) return;
    ProbDev &pr = probs[qlist[blockIdx.x]];
    const int phase = pr.phase;
    if (phase == PH_DONE) return;
    const PartDev &pa = parts[pr.part];
    const int n = pa.n_local, nf = pa.n_feat, l = pa.l, tid = threadIdx.x;
    // The intercept's column of X'c (the sum of the row coefficients in row order) is folded before this kernel runs: dense tiles by
    // k_ro_dense_cols (on a spare lane, with the loss sum), sliced CSR partitions by the chain workgroups of the column pass's first
    // launch (k_colpass_lds<.., RO>: ro_csum_chain). It is the one chain of a tick the exact parallel fold cannot help: its running sum
    // stays as small as its terms (93 % of the rows push it up a little, 7 % pull it down a lot), the binade changes every few terms --
    // a LITERAL chain of l dependent adds, 0.25 ms at configs[2], which sat on this kernel's critical path (between the d.Hd pass and
    // alpha) while it ran here. (Only a partition without any column work unit still folds it here: ro_chain_chunk.)
    const bool dn = pa.dense != 0;
    const bool csum_done = dn || (pa.sell != 0 && pa.n_cunits > 0);
    const double *__restrict__ xtc = pr.c0f;                      // X'c of this tick, columns 0 .. nf-1 (k_colpass_lds<.., RO>)
    const double *__restrict__ coef = pr.coef;                    // the row coefficients: their sum in row order is the intercept's column
    const double *__restrict__ pvec = pr.pinv_vec;
    const double pscal = pr.pinv;
    double *__restrict__ w = pr.w, *__restrict__ w_new = pr.w_new, *__restrict__ g = pr.g, *__restrict__ s = pr.s,
           *__restrict__ d = pr.d, *__restrict__ Hd = pr.Hd;
    const double *__restrict__ m = pr.m;
    const double zero6[RO_NF] = {0.0, 0.0, 0.0, 0.0};
    double res[RO_NF];
    ROP_INIT;

    if (phase == PH_CG) {
        ROP_COUNT(8);
        SGF_PASS(0);
        // ---- one trip of trcg's loop (bw/Tron.java:145-175)
        // Hd for the feature columns and the first nf terms of Tron.dot(d, Hd) on one lane; beside it, on a second lane, the intercept's
        // column of XTv: the sum of the row coefficients in row order (the bias entry closes every row). The dot's last term needs
        // that sum and is added after the pass: the same chain.
        // (dense tiles: k_ro_dense_cols ran that chain on a spare lane and left it in csump[0])
        struct RA { RoV4 d, x, p, c; };
        const int lensA[2] = {nf, csum_done ? 0 : l};
        ro_pass<2, 0, 2, RA>(sh, csum_done ? nf : max(nf, l), lensA, zero6, res,
            [&](int j0, RA &R) { R.d = ro_ld4c(d, j0, n); R.x = ro_ld4c(xtc, j0, n); if (pvec) R.p = ro_ld4s(pvec, j0, n); if (!csum_done) R.c = ro_ld4c(coef, j0, l); },
            [&](int j0, RA &R, double (&ct)[2][RO_E], double (&nv)[1][RO_E]) {
                double hd[RO_E];
#pragma unroll
                for (int e = 0; e < RO_E; e++) {
                    hd[e] = R.d.v[e] * (pvec ? R.p.v[e] : pscal) + R.x.v[e];      // Hs[i] = (s[i]*priorVar_inv[i] + Hs[i]) * 1
                    ct[0][e] = R.d.v[e] * hd[e];                                   // Tron.dot(d, Hd)
                    ct[1][e] = csum_done ? 0.0 : R.c.v[e];                         // XTv[n-1] += v[i]
                }
                ro_st4(Hd, j0, nf, hd);
            });
        ROP_MARK(0);
        if (csum_done) res[1] = pr.csump[0];
        const double rTr0 = pr.rTr, delta0 = pr.delta, cgtol0 = pr.cgtol;
        const double d_icpt = d[nf];
        const double hd_icpt = d_icpt * (pvec ? pvec[nf] : pscal) + res[1];
        if (tid == 0) Hd[nf] = hd_icpt;
        __syncthreads();
        double alpha = rTr0 / (res[0] + d_icpt * hd_icpt);
        const double nalpha = -alpha;
        const double *__restrict__ rc = pr.rb[pr.rsel];
        double *__restrict__ rn = pr.rb[pr.rsel ^ 1];
        // daxpy(alpha, d, s) and the continuation r' = r - alpha Hd with r'.r' (:169-171). The two norms of this step are only COMPARED:
        // `euclideanNorm(s) > delta` (:150) and `euclideanNorm(r) <= cgtol` (:144 of the next trip). Round 6's first form evaluated
        // both recurrences in every CG step (two of the three folds, and the staging waves' scan, hand-shake and two divisions per
        // element were the pass's critical path: 245 us of a 490 us step). Now the staging threads add s'^2 on the side (any order),
        // r'.r' is there anyway, and the recurrence runs only when such a sum is too close to its threshold (ro_norm_decided).
        struct RB { RoV4 d, s, r, h; };
        double ssq_part = 0.0;
        SGF_PASS(1);
        ro_pass<1, 0, 0, RB>(sh, n, nullptr, zero6, res,
            [&](int j0, RB &R) { R.d = ro_ld4c(d, j0, n); R.s = ro_ld4c(s, j0, n); R.r = ro_ld4c(rc, j0, n); R.h = ro_ld4c(Hd, j0, n); },
            [&](int j0, RB &R, double (&ct)[1][RO_E], double (&nv)[1][RO_E]) {
                double s1[RO_E], r1[RO_E];
#pragma unroll
                for (int e = 0; e < RO_E; e++) {
                    s1[e] = R.s.v[e] + alpha * R.d.v[e];                     // daxpy(alpha, d, s)
                    r1[e] = R.r.v[e] + nalpha * R.h.v[e];                    // daxpy(-alpha, Hd, r)
                    ct[0][e] = r1[e] * r1[e];
                    if (j0 + e < n) ssq_part = ssq_part + s1[e] * s1[e];
                }
                ro_st4(s, j0, n, s1);
                ro_st4(rn, j0, n, r1);
            });
        ROP_MARK(1);
        SGF_PASS(2);
        const double rnew = res[0];
        const double ssq = ro_block_sum(sh, ssq_part);
        bool boundary = false, end_cg = false, nan = false;
        if (!exact_norms && ro_norm_decided(ssq, delta0, n)) boundary = sqrt(ssq) > delta0;
        else {
            const double snorm = ro_exact_norm(s, n);
            nan = !(snorm == snorm);
            boundary = snorm > delta0;
        }
        double alpha2 = 0.0, beta = 0.0;
        if (boundary) {
            // cg reaches trust region boundary (:150-168): the three dots on the stepped-back s (:152-155). (Round 5 computed them
            // speculatively in the pass above -- six folds side by side; with the folds no longer the bottleneck the common CG step
            // carries three, and the boundary step -- at most one per trcg call -- pays a pass of its own.)
            struct RB2 { RoV4 d, s; };
            double res2[RO_NF];
            ro_pass<3, 0, 0, RB2>(sh, n, nullptr, zero6, res2,
                [&](int j0, RB2 &R) { R.d = ro_ld4c(d, j0, n); R.s = ro_ld4c(s, j0, n); },
                [&](int j0, RB2 &R, double (&ct)[3][RO_E], double (&nv)[1][RO_E]) {
#pragma unroll
                    for (int e = 0; e < RO_E; e++) {
                        const double sb = R.s.v[e] + nalpha * R.d.v[e];      // daxpy(-alpha, d, s) (:153)
                        ct[0][e] = sb * R.d.v[e]; ct[1][e] = sb * sb; ct[2][e] = R.d.v[e] * R.d.v[e];
                    }
                });
            ROP_MARK(2);
            const double std_ = res2[0], sts = res2[1], dtd = res2[2];
            const double dsq = delta0 * delta0;
            const double rad = sqrt(std_ * std_ + dtd * (dsq - sts));
            if (std_ >= 0) alpha2 = (dsq - sts) / (std_ + rad);
            else alpha2 = (rad - std_) / dtd;
            end_cg = true;
        } else {
            beta = rnew / rTr0;
            // loop-top test of the next trip (:144): euclideanNorm(r') <= cgtol
            if (!exact_norms && ro_norm_decided(rnew, cgtol0, n)) end_cg = sqrt(rnew) <= cgtol0;
            else end_cg = ro_exact_norm(rn, n) <= cgtol0;
        }
        if (nan) end_cg = true;
        const double nalpha2 = -alpha2;
        struct RC { RoV4 d, s, r1, r, h, w, g; };
        auto ldc = [&](int j0, RC &R) {
            R.d = ro_ld4c(d, j0, n);
            if (boundary) { R.s = ro_ld4c(s, j0, n); R.r = ro_ld4c(rc, j0, n); R.h = ro_ld4c(Hd, j0, n); }
            else { R.r1 = ro_ld4c(rn, j0, n); if (end_cg) R.s = ro_ld4c(s, j0, n); }
            if (end_cg) { R.w = ro_ld4c(w, j0, n); R.g = ro_ld4c(g, j0, n); }
        };
        auto emc = [&](int j0, RC &R, double (&ct)[3][RO_E], double (&nv)[1][RO_E]) {
            double sf[RO_E], rf[RO_E], dn[RO_E], wn[RO_E];
#pragma unroll
            for (int e = 0; e < RO_E; e++) {
                sf[e] = R.s.v[e]; rf[e] = R.r1.v[e];
                if (boundary) {
                    const double sb = R.s.v[e] + nalpha * R.d.v[e];          // daxpy(-alpha, d, s)
                    sf[e] = sb + alpha2 * R.d.v[e];                           // daxpy(alpha', d, s)
                    rf[e] = R.r.v[e] + nalpha2 * R.h.v[e];                    // daxpy(-alpha', Hd, r)
                } else {
                    double dj = R.d.v[e];
                    if (beta != 1.0) dj = dj * beta;                          // scale(beta, d)
                    dn[e] = dj + 1.0 * R.r1.v[e];                             // daxpy(one, r, d)
                }
                if (end_cg) {
                    wn[e] = R.w.v[e] + 1.0 * sf[e];                           // w_new = w + s (:69-70)
                    nv[0][e] = sf[e];
                    ct[1][e] = R.g.v[e] * sf[e];                              // gs = dot(g, s)
                    ct[2][e] = sf[e] * rf[e];                                 // dot(s, r)
                } else { nv[0][e] = 0.0; ct[1][e] = 0.0; ct[2][e] = 0.0; }
            }
            if (boundary) { ro_st4(s, j0, n, sf); ro_st4(rn, j0, n, rf); }
            else ro_st4(d, j0, n, dn);
            if (end_cg) ro_st4(w_new, j0, n, wn);
        };
        if (end_cg) ro_pass<3, 1, 0, RC>(sh, n, nullptr, zero6, res, ldc, emc);
        else {
            // the common step (no boundary, trcg goes on): scale(beta, d); daxpy(one, r, d) (:172-174) -- two vectors in, one out, streamed
            struct RD { RoV4 d, r1; };
            ro_stream<RO_STREAM_G, RD>(n,
                [&](int j0, RD &R) { R.d = ro_ld4c(d, j0, n); R.r1 = ro_ld4c(rn, j0, n); },
                [&](int j0, RD &R) {
                    double dn[RO_E];
#pragma unroll
                    for (int e = 0; e < RO_E; e++) {
                        double dj = R.d.v[e];
                        if (beta != 1.0) dj = dj * beta;                      // scale(beta, d)
                        dn[e] = dj + 1.0 * R.r1.v[e];                         // daxpy(one, r, d)
                    }
                    ro_st4(d, j0, n, dn);
                });
            __syncthreads();
        }
        ROP_MARK(3);
        if (tid == 0) {
            if (!boundary) pr.rTr = rnew;
            pr.rsel ^= 1;
            pr.cg_iter += 1;
            pr.ticks += 1;
            if (nan) pr.status = ST_NAN;          // the EVAL tick that follows still runs; the host reports the status
            if (end_cg) {
                pr.gs = res[1];
                pr.prered = -0.5 * (res[1] - res[2]);
                pr.snorm = res[0];                // euclideanNorm(s) of tron() (:80): the same loop over the same vector
                pr.newton += 1;
                pr.cg_total += pr.cg_iter;
                pr.phase = PH_EVAL;
            }
        }
        return;
    }

    // ---- PH_EVAL0 / PH_EVAL: fun(w_new) and the gradient candidate (llf/LogisticRegressionL2.java:156-225)
    const double *__restrict__ rowtmp = pr.rowtmp;
    struct RR { RoV4 x, c; };
    const int lensR[2] = {l, csum_done ? 0 : l};
    SGF_PASS(3);
    if (dn) { res[0] = pr.lossp[0]; res[1] = pr.csump[0]; }
    else ro_pass<2, 0, 2, RR>(sh, l, lensR, zero6, res,
        [&](int j0, RR &R) { R.x = ro_ld4c(rowtmp, j0, l); R.c = ro_ld4c(coef, j0, l); },
        [&](int j0, RR &R, double (&ct)[2][RO_E], double (&nv)[1][RO_E]) {
#pragma unroll
            for (int e = 0; e < RO_E; e++) {
                ct[0][e] = R.x.v[e];                                         // f += weight * log(1 + exp(..)) in row order (:172-183)
                ct[1][e] = R.c.v[e];                                         // the intercept's column of XTv: the coefficients in row order
            }
        });
    ROP_COUNT(9);
    ROP_MARK(4);
    SGF_PASS(4);
    if (csum_done) res[1] = pr.csump[0];
    double fnew = 2.0 * res[0];
    const double csum = res[1];
    const double *__restrict__ c0 = pa.c0;
    const bool e0 = (phase == PH_EVAL0);
    struct RE { RoV4 w, m, x, p, c; };
    auto lde = [&](int j0, RE &R) {
        R.w = ro_ld4c(w_new, j0, n); R.m = ro_ld4c(m, j0, n); R.x = ro_ld4c(xtc, j0, n);
        if (pvec) R.p = ro_ld4s(pvec, j0, n);
        if (e0) R.c = ro_ld4s(c0, j0, n);
    };
    // one element: t t p (fun :187-188, added to the running f), the gradient candidate hd (grad :224, multiplier 1), grad(0) (bw/Tron.java:50-53)
    auto el = [&](int j, const RE &R, int e, double &ttp, double &hd, double &g0) {
        const double xa = (j == nf) ? csum : R.x.v[e];
        const double pj = pvec ? R.p.v[e] : pscal;
        const double t = R.w.v[e] - R.m.v[e];
        ttp = t * t * pj;
        hd = t * pj + xa;
        g0 = e0 ? (0.0 - R.m.v[e]) * pj + R.c.v[e] : 0.0;
    };
    double gnorm1_c = 0.0;
    if (e0) {
        // the first evaluation of a solve: euclideanNorm(g), euclideanNorm(grad(0)), f, and r.r of the trcg call that starts from this g (r = -g)
        const double init4[RO_NF] = {0.0, 0.0, fnew, 0.0};
        ro_pass<4, 2, 0, RE>(sh, n, nullptr, init4, res, lde,
            [&](int j0, RE &R, double (&ct)[4][RO_E], double (&nv)[2][RO_E]) {
                double hd[RO_E];
#pragma unroll
                for (int e = 0; e < RO_E; e++) {
                    el(j0 + e, R, e, ct[2][e], hd[e], nv[1][e]);
                    nv[0][e] = hd[e];
                    ct[3][e] = hd[e] * hd[e];
                }
                ro_st4(Hd, j0, n, hd);
            });
        gnorm1_c = res[1];
    } else {
        // every later one: no grad(0) -- one norm recurrence instead of two in the staging waves, three folds instead of four
        const double init3[RO_NF] = {0.0, fnew, 0.0, 0.0};
        double r3[RO_NF];
        ro_pass<3, 1, 0, RE>(sh, n, nullptr, init3, r3, lde,
            [&](int j0, RE &R, double (&ct)[3][RO_E], double (&nv)[1][RO_E]) {
                double hd[RO_E], g0;
#pragma unroll
                for (int e = 0; e < RO_E; e++) {
                    el(j0 + e, R, e, ct[1][e], hd[e], g0);
                    nv[0][e] = hd[e];
                    ct[2][e] = hd[e] * hd[e];
                }
                ro_st4(Hd, j0, n, hd);
            });
        res[0] = r3[0]; res[2] = r3[1]; res[3] = r3[2];
    }
    ROP_MARK(5);
    fnew = res[2] / 2.0;
    const double gnorm_c = res[0], gsq_c = res[3];
    bool start_trcg = false, finished = false, copy_w = false, copy_g = false;
    double gnorm_cur = pr.gnorm, gsq = pr.gsq;
    if (e0) {
        // Tron prologue (:47-62)
        const double gnorm1 = gnorm1_c;
        if (tid == 0) { pr.f = fnew; pr.gnorm1 = gnorm1; pr.gnorm = gnorm_c; pr.delta = gnorm_c; pr.dsel ^= 1; pr.ticks += 1; }
        gnorm_cur = gnorm_c; gsq = gsq_c;
        copy_g = true;
        if (gnorm_c <= pr.eps * gnorm1) finished = true;           // search = 0
        else start_trcg = true;
        if (!(fnew == fnew) || !(gnorm_c == gnorm_c) || !(gnorm1 == gnorm1)) { finished = true; start_trcg = false; if (tid == 0) pr.status = ST_NAN; }
    } else {
        const double eta0 = 1e-4, eta1 = 0.25, eta2 = 0.75;
        const double sigma1 = 0.25, sigma2 = 0.5, sigma3 = 4;
        double f = pr.f, delta = pr.delta, gnorm = gnorm_cur;
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
        int iter = pr.iter;
        bool brk = false;
        const bool accept = actred > eta0 * prered;
        if (accept) {
            iter++;
            copy_w = copy_g = true;
            f = fnew;
            gnorm = gnorm_c; gsq = gsq_c;
            if (gnorm <= pr.eps * pr.gnorm1) brk = true;
        }
        if (!brk) {
            if (f < -1.0e+32) brk = true;
            else if (fabs(actred) <= 0 && prered <= 0) brk = true;
            else if (fabs(actred) <= 1.0e-12 * fabs(f) && fabs(prered) <= 1.0e-12 * fabs(f)) brk = true;
        }
        const int max_iter = pr.max_iter;
        __syncthreads();
        if (tid == 0) {
            pr.f = f; pr.delta = delta; pr.gnorm = gnorm; pr.iter = iter; pr.ticks += 1;
            if (accept) { pr.accepted += 1; pr.dsel ^= 1; }
        }
        if (!(fnew == fnew) || !(gnorm == gnorm)) { brk = true; if (tid == 0) pr.status = ST_NAN; }
        gnorm_cur = gnorm;
        if (brk || iter > max_iter) finished = true;               // while (iter <= max_iter && search)
        else start_trcg = true;
    }
    const bool nullstep = start_trcg && gnorm_cur <= 0.1 * gnorm_cur;      // trcg leaves its loop at once (:144): only for g = 0
    if (copy_w || copy_g || start_trcg) {
        // w = w_new, g = grad(w_new); trcg prologue (:133-141): s = 0, r = -g, d = r
        double *__restrict__ r0 = pr.rb[0];
        struct RT { RoV4 h, g, wn, w; };
        ro_pass<0, 0, 0, RT>(sh, n, nullptr, zero6, res,
            [&](int j0, RT &R) {
                R.h = ro_ld4c(Hd, j0, n);
                if (!copy_g) R.g = ro_ld4c(g, j0, n);
                if (copy_w || nullstep) R.wn = ro_ld4c(w_new, j0, n);
                if (nullstep && !copy_w) R.w = ro_ld4c(w, j0, n);
            },
            [&](int j0, RT &R, double (&ct)[1][RO_E], double (&nv)[1][RO_E]) {
                double z4[RO_E], rj[RO_E], wz[RO_E];
#pragma unroll
                for (int e = 0; e < RO_E; e++) {
                    const double gj = copy_g ? R.h.v[e] : R.g.v[e];
                    z4[e] = 0.0; rj[e] = -gj;
                    wz[e] = (copy_w ? R.wn.v[e] : R.w.v[e]) + 1.0 * 0.0;
                }
                if (copy_w) ro_st4(w, j0, n, R.wn.v);
                if (copy_g) ro_st4(g, j0, n, R.h.v);
                if (start_trcg) {
                    ro_st4(s, j0, n, z4); ro_st4(r0, j0, n, rj); ro_st4(d, j0, n, rj);
                    if (nullstep) ro_st4(w_new, j0, n, wz);       // the (null) step is evaluated like any other
                }
            });
    }
    ROP_MARK(6);
    if (tid == 0) {
        pr.gsq = gsq;
        if (start_trcg) {
            pr.rTr = gsq;                      // r = -g: Tron.dot(r, r) adds the same products in the same order as g.g
            pr.cgtol = 0.1 * gnorm_cur;
            pr.cg_iter = 0;
            pr.rsel = 0;
            if (nullstep) { pr.gs = 0.0; pr.prered = -0.5 * (0.0 - 0.0); pr.snorm = 0.0; pr.newton += 1; pr.phase = PH_EVAL; }
            else pr.phase = PH_CG;
        }
        if (finished) {
            pr.phase = PH_DONE;
            atomicAdd(done_counter, 1);
        }
    }
}

// c0 = X' t0 (the data part of grad(0)) from an EVAL pass at w = 0 through the reference-order passes: the chained column sums
__global__ void __launch_bounds__(256)
k_ro_collect_c0(const PartDev *__restrict__ parts, const ProbDev *__restrict__ probs, const int *__restrict__ qlist,
                double *const *__restrict__ c0_ptrs)
{
    const ProbDev &pr = probs[qlist[blockIdx.x]];
    const PartDev &pa = parts[pr.part];
    double *__restrict__ out = c0_ptrs[blockIdx.x];
    for (int j = threadIdx.x; j < pa.n_feat; j += blockDim.x) out[j] = pr.c0f[j];
    if (threadIdx.x == 0) {
        double a = 0.0;                                   // the intercept's column: the coefficients in row order (once per partition)
        for (int i = 0; i < pa.l; i++) a = a + pr.coef[i];
        out[pa.n_feat] = a;
    }
}
