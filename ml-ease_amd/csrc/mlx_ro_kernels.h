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
//     k_ro_step below -- one 320-thread workgroup per problem walks the vectors in chunks of 1024 elements; waves 1-4 do the
//     elementwise work of a chunk (the daxpy / scale statements, Hs[i] = s[i]*priorVar_inv[i] + Hs[i], the products a[i]*b[i] of
//     Tron.dot) and leave the chunk's TERMS in one of two LDS buffers; up to six lanes of wave 0 meanwhile fold one term array each IN
//     INDEX ORDER out of the other buffer (p += term: the loop of Tron.dot :204-213), so the six reductions a CG step needs at once
//     cost one chain, not six, and the staging costs the chain nothing.
//     euclideanNorm (:220-252) keeps a running scale: its update is `sum = 1 + sum*(scale/a)^2` when |v_i| exceeds the scale and
//     `sum += (a/scale)^2` otherwise. The scale before element i is the running maximum of |v| -- an exclusive prefix maximum, exact
//     in any association -- so the chunk's threads compute it by a scan, form every element's (m, c) with sum' = c + sum*m in
//     parallel (the divisions are off the chain), and the folding lane runs the chain; 32-term sub-blocks without a new maximum
//     (all but ~ln n of them) take the plain `sum += c` loop.
//   * exp / log1p: the portable forms of portable_math.h, as the oracle's verification twin evaluates them (device and host libm
//     differ in the last bit).
// Same statements in the same order as tron_step_body<SEQ> (which is bit-identical to the oracle): the tests run both.
#pragma once

#ifndef RO_CHUNK
#define RO_CHUNK 1024        // elements per chunk (a multiple of 256: one staging wave per 256; A/B: tools/ablate_build.sh -DRO_CHUNK=512)
#endif
constexpr int RO_CH = RO_CHUNK, RO_NSW = RO_CH / 256, RO_T = 64 * (1 + RO_NSW), RO_CHP = RO_CH + 2, RO_NF = 6, RO_NN = 2;      // one folding wave + RO_NSW staging waves
static_assert(RO_CH % 256 == 0 && RO_NSW >= 1 && RO_NSW <= 4, "chunk = 256 elements per staging wave, at most 32 sub-blocks of 32 terms");
struct RoLds {
    double C[2][RO_NF][RO_CHP];    // terms of a chunk, one array per folding lane (norm arrays first); two buffers: folded / being staged
    double M[2][RO_NN][RO_CHP];    // multipliers of the norm arrays (1.0 except where the running scale changes)
    double wtot[RO_NN][RO_NSW];    // scan: the staging waves' maxima of the chunk being staged
    double res[RO_NF], mcfin[RO_NN];
    volatile int seq[RO_NSW];      // chunk number (+ 1) each staging wave has published its maximum for
    unsigned mask8[2][RO_NSW];     // per buffer and staging wave, bit k: terms 32 k .. 32 k + 31 of its 256 hold a change of a running scale
    double pad[64];                // ro_fold32 reads up to 48 doubles ahead of the last term it adds
};
struct RoV4 { double v[4]; };

// four consecutive elements j0 .. j0+3 of a 32-byte aligned vector (the work vectors: 256-byte aligned pieces of the slab, padded so
// that the last quad stays inside the piece -- mlx_finalize's carve_size)
__device__ __forceinline__ RoV4 ro_ld4(const double *__restrict__ p, int j0)
{
    typedef double d2v_t __attribute__((ext_vector_type(2)));
    const d2v_t a = gld(reinterpret_cast<const d2v_t *>(p + j0)), b = gld(reinterpret_cast<const d2v_t *>(p + j0 + 2));
    RoV4 r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    return r;
}
__device__ __forceinline__ RoV4 ro_ld4c(const double *__restrict__ p, int j0, int n)       // ... of a vector of n elements, clamped to its last quad
{
    return ro_ld4(p, min(j0, ((n - 1) >> 2) << 2));
}
__device__ __forceinline__ RoV4 ro_ld4s(const double *__restrict__ p, int j0, int n)      // any alignment, clamped
{
    RoV4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) r.v[e] = gld(p + min(j0 + e, n - 1));
    return r;
}
__device__ __forceinline__ void ro_st4(double *__restrict__ p, int j0, int n, const double (&x)[4])
{
#pragma unroll
    for (int e = 0; e < 4; e++) if (j0 + e < n) gst(p + j0 + e, x[e]);
}

// s += t[0]; s += t[1]; ... over 32 * nit consecutive doubles at the LDS address t, one dependent v_add_f64 per term: 16 terms are
// added while the ds_reads of the next 16 are in flight (written as one asm block: the compiler sinks a read-ahead written in C back
// into the iteration that uses it, which costs one LDS latency per batch -- 16 cycles per term instead of 8). nit is wave-uniform,
// >= 1; the last trip reads 16 doubles ahead of its terms (inside the LDS allocation: RoLds is padded), unused.
__device__ __forceinline__ double ro_fold32(double s, const double *t, int nit)
{
    unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)t;
    // registers v[192:223] / v[224:255] hold 16 terms each; the eight 16-byte reads of one half are issued between the first adds of
    // the other half (an add waits for its predecessor anyway: the reads ride in those stalls)
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_read_b128 v[192:195], %[a] offset:0\n\t"
        "ds_read_b128 v[196:199], %[a] offset:16\n\t"
        "ds_read_b128 v[200:203], %[a] offset:32\n\t"
        "ds_read_b128 v[204:207], %[a] offset:48\n\t"
        "ds_read_b128 v[208:211], %[a] offset:64\n\t"
        "ds_read_b128 v[212:215], %[a] offset:80\n\t"
        "ds_read_b128 v[216:219], %[a] offset:96\n\t"
        "ds_read_b128 v[220:223], %[a] offset:112\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_f64 %[s], %[s], v[192:193]\n\t"
        "ds_read_b128 v[224:227], %[a] offset:128\n\t"
        "v_add_f64 %[s], %[s], v[194:195]\n\t"
        "ds_read_b128 v[228:231], %[a] offset:144\n\t"
        "v_add_f64 %[s], %[s], v[196:197]\n\t"
        "ds_read_b128 v[232:235], %[a] offset:160\n\t"
        "v_add_f64 %[s], %[s], v[198:199]\n\t"
        "ds_read_b128 v[236:239], %[a] offset:176\n\t"
        "v_add_f64 %[s], %[s], v[200:201]\n\t"
        "ds_read_b128 v[240:243], %[a] offset:192\n\t"
        "v_add_f64 %[s], %[s], v[202:203]\n\t"
        "ds_read_b128 v[244:247], %[a] offset:208\n\t"
        "v_add_f64 %[s], %[s], v[204:205]\n\t"
        "ds_read_b128 v[248:251], %[a] offset:224\n\t"
        "v_add_f64 %[s], %[s], v[206:207]\n\t"
        "ds_read_b128 v[252:255], %[a] offset:240\n\t"
        "v_add_f64 %[s], %[s], v[208:209]\n\t"
        "v_add_f64 %[s], %[s], v[210:211]\n\t"
        "v_add_f64 %[s], %[s], v[212:213]\n\t"
        "v_add_f64 %[s], %[s], v[214:215]\n\t"
        "v_add_f64 %[s], %[s], v[216:217]\n\t"
        "v_add_f64 %[s], %[s], v[218:219]\n\t"
        "v_add_f64 %[s], %[s], v[220:221]\n\t"
        "v_add_f64 %[s], %[s], v[222:223]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_u32 %[a], 0x100, %[a]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "v_add_f64 %[s], %[s], v[224:225]\n\t"
        "ds_read_b128 v[192:195], %[a] offset:0\n\t"
        "v_add_f64 %[s], %[s], v[226:227]\n\t"
        "ds_read_b128 v[196:199], %[a] offset:16\n\t"
        "v_add_f64 %[s], %[s], v[228:229]\n\t"
        "ds_read_b128 v[200:203], %[a] offset:32\n\t"
        "v_add_f64 %[s], %[s], v[230:231]\n\t"
        "ds_read_b128 v[204:207], %[a] offset:48\n\t"
        "v_add_f64 %[s], %[s], v[232:233]\n\t"
        "ds_read_b128 v[208:211], %[a] offset:64\n\t"
        "v_add_f64 %[s], %[s], v[234:235]\n\t"
        "ds_read_b128 v[212:215], %[a] offset:80\n\t"
        "v_add_f64 %[s], %[s], v[236:237]\n\t"
        "ds_read_b128 v[216:219], %[a] offset:96\n\t"
        "v_add_f64 %[s], %[s], v[238:239]\n\t"
        "ds_read_b128 v[220:223], %[a] offset:112\n\t"
        "v_add_f64 %[s], %[s], v[240:241]\n\t"
        "v_add_f64 %[s], %[s], v[242:243]\n\t"
        "v_add_f64 %[s], %[s], v[244:245]\n\t"
        "v_add_f64 %[s], %[s], v[246:247]\n\t"
        "v_add_f64 %[s], %[s], v[248:249]\n\t"
        "v_add_f64 %[s], %[s], v[250:251]\n\t"
        "v_add_f64 %[s], %[s], v[252:253]\n\t"
        "v_add_f64 %[s], %[s], v[254:255]\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [s] "+v"(s), [a] "+v"(a), [n] "+s"(nit)
        :
        : "memory", "scc",
          "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
    return s;
}

// One pass over elements 0 .. len-1. load(j0, R) fetches the operands of elements j0 .. j0+3 (it clamps to the last quad of each
// vector itself), emit(j0, R, ct, nv) does their elementwise work (stores included, elements beyond a vector's end masked) and
// returns the terms of the dot arrays ct[k][e] (k >= NN) and the raw values of the norm arrays nv[q][e] (q < NN). Lane k of the
// first wave folds array k over its first lens[k] elements (lens == nullptr: len for all): norms from sum = 1, dots from init[k].
// result[k] (every thread): the norm scale*sqrt(sum) / the dot.
template <int NF, int NN, typename R, typename LD, typename EM>
__device__ __forceinline__ void ro_pass(RoLds &sh, int len, const int *lens, const double *init, double *result, LD load, EM emit)
{
#pragma clang fp contract(off)
    static_assert(NF <= RO_NF && NN <= RO_NN && NN <= NF, "fold arrays");
    constexpr int NFX = NF > 0 ? NF : 1, NNX = NN > 0 ? NN : 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (NF == 0) {
        // elementwise only: no terms, no barriers; four quads per thread in flight (a single one leaves every trip waiting for HBM)
        for (int base = 0; base < len; base += 16 * RO_T) {
            R r4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) load(base + u * 4 * RO_T + 4 * tid, r4[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                double ct[NFX][4], nv[NNX][4];
                emit(base + u * 4 * RO_T + 4 * tid, r4[u], ct, nv);
            }
        }
        __syncthreads();
        return;
    }
    // Wave 0 FOLDS; waves 1-4 (256 threads) STAGE: while the folding wave runs the chains of chunk c out of one LDS buffer, the staging
    // waves load, compute and store chunk c + 1 into the other (their work -- memory latency, the scan, the divisions of the norms'
    // terms -- is off the chains' critical path: one barrier per chunk, the folding wave never stages). The staging waves agree on the
    // running maximum of a norm's operand among themselves: each publishes its quarter's maximum in LDS and its chunk number behind it
    // (sh.seq), and reads the others' when all four numbers are there.
    const bool folder = wave == 0;
    const int sw = wave - 1, stid = tid - 64;                 // staging wave 0..RO_NSW-1, staging thread 0..RO_CH/4-1
    const int nch = (len + RO_CH - 1) / RO_CH;
    const int mylen = (folder && lens != nullptr) ? lens[lane < NF ? lane : 0] : len;      // (wave 0: the array this lane folds)
    double acc = 0.0;
    if (folder) acc = (lane < NN) ? 1.0 : init[lane < NF ? lane : 0];
    double mc[NNX];
#pragma unroll
    for (int q = 0; q < NNX; q++) mc[q] = 0.0;
    auto stage = [&](int c) {
        const int b = c & 1;
        const int j0 = c * RO_CH + 4 * stid;
        R regs;
        load(j0, regs);
        double ct[NFX][4], nv[NNX][4];
        emit(j0, regs, ct, nv);
        int flag = 0;
        if (NN > 0) {
            // euclideanNorm's running scale in front of every element = exclusive prefix maximum of |v| (zeros never raise it)
            double a[NNX][4], x[NNX];
#pragma unroll
            for (int q = 0; q < NN; q++) {
#pragma unroll
                for (int e = 0; e < 4; e++) a[q][e] = (j0 + e < (lens != nullptr ? lens[q] : len)) ? fabs(nv[q][e]) : 0.0;
                x[q] = fmax(fmax(a[q][0], a[q][1]), fmax(a[q][2], a[q][3]));
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const double y = __shfl_up(x[q], d);
                    if (lane >= d) x[q] = fmax(x[q], y);
                }
                if (lane == 63) sh.wtot[q][sw] = x[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) sh.seq[sw] = c + 1;
            for (int w = 0; w < RO_NSW; w++) while (sh.seq[w] < c + 1) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int q = 0; q < NN; q++) {
                double ex = __shfl_up(x[q], 1);
                if (lane == 0) ex = 0.0;
                double P = fmax(mc[q], ex);
                for (int w = 0; w < RO_NSW; w++) {
                    const double t = sh.wtot[q][w];
                    if (w < sw) P = fmax(P, t);
                    mc[q] = fmax(mc[q], t);
                }
                double mm[4], cc[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double ae = a[q][e];
                    if (!(ae != 0.0)) { mm[e] = 1.0; cc[e] = 0.0; }               // v[i] == 0: skipped
                    else if (P < ae) { const double t = P / ae; mm[e] = t * t; cc[e] = 1.0; flag = 1; }      // sum = 1 + sum * (t * t); scale = a
                    else { const double t = ae / P; mm[e] = 1.0; cc[e] = t * t; }                                // sum += t * t
                    P = fmax(P, ae);
                }
                typedef double d2v_t __attribute__((ext_vector_type(2)));
                d2v_t *cp = reinterpret_cast<d2v_t *>(&sh.C[b][q][4 * stid]), *mp = reinterpret_cast<d2v_t *>(&sh.M[b][q][4 * stid]);
                cp[0] = (d2v_t){cc[0], cc[1]}; cp[1] = (d2v_t){cc[2], cc[3]};
                mp[0] = (d2v_t){mm[0], mm[1]}; mp[1] = (d2v_t){mm[2], mm[3]};
            }
            // which of this wave's eight 32-term sub-blocks hold a change of a running scale (lanes 8 k .. 8 k + 7 own sub-block k)
            const unsigned long long bal = __ballot(flag != 0);
            unsigned m8 = 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) if ((bal >> (8 * k)) & 0xFFull) m8 |= 1u << k;
            if (lane == 0) sh.mask8[b][sw] = m8;
        }
#pragma unroll
        for (int k = NN; k < NF; k++) {
            typedef double d2v_t __attribute__((ext_vector_type(2)));
            d2v_t *cp = reinterpret_cast<d2v_t *>(&sh.C[b][k][4 * stid]);
            cp[0] = (d2v_t){ct[k][0], ct[k][1]}; cp[1] = (d2v_t){ct[k][2], ct[k][3]};
        }
    };
    if (NN > 0 && tid < RO_NSW) sh.seq[tid] = 0;
    __syncthreads();
    if (!folder) stage(0);
    __syncthreads();
    for (int c = 0; c < nch; c++) {
        if (!folder) {
            if (c + 1 < nch) stage(c + 1);
        } else if (lane < NF) {
            const int b = c & 1, base = c * RO_CH;
            const int cnt = max(0, min(RO_CH, mylen - base));
            const double *cp = &sh.C[b][lane][0];
            const double *mp = &sh.M[b][lane < NN ? lane : 0][0];
            const bool isn = lane < NN;
            unsigned mask = 0u;
            if (NN > 0) {
#pragma unroll
                for (int w = 0; w < RO_NSW; w++) mask |= sh.mask8[b][w] << (8 * w);
            }
            double s = acc;
            // p += term: ONE dependent add per element. Sub-blocks of 32 terms common to the folding lanes run in ro_fold32 (the LDS
            // reads of the next 16 terms in flight while 16 are added) -- except the few sub-blocks in which a norm's running scale
            // changes (euclideanNorm's `sum = 1 + sum * (scale/a)^2`: about ln n of them per vector): those, and what is left of a
            // lane's chunk behind the common part, take the per-term form sum = c + sum * m (m = 1.0 wherever nothing changes:
            // c + sum * 1.0 is sum + c bit for bit).
            int T = RO_CH / 32;
#pragma unroll
            for (int k = 0; k < NF; k++) {
                const int ck = __shfl(cnt, k);
                if (ck >= 32) T = min(T, ck >> 5);
            }
            T = __builtin_amdgcn_readfirstlane(T);
            int i = 0;
            if (cnt >= 32) {
                int bq = 0;
                while (bq < T) {
                    const unsigned rest = mask >> bq;
                    if (rest & 1u) {
                        for (int e = 32 * bq; e < 32 * bq + 32; e += 8) {
                            double c8[8], m8[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) { c8[u] = cp[e + u]; m8[u] = isn ? mp[e + u] : 1.0; }
#pragma unroll
                            for (int u = 0; u < 8; u++) s = c8[u] + s * m8[u];
                        }
                        bq += 1;
                    } else {
                        int run = rest == 0u ? T - bq : min(T - bq, (int)__builtin_ctz(rest));
                        run = __builtin_amdgcn_readfirstlane(run);
                        s = ro_fold32(s, cp + 32 * bq, run);
                        bq += run;
                    }
                }
                i = T << 5;
            }
            for (; i < cnt; i++) {
                const double m = isn ? mp[i] : 1.0;
                s = cp[i] + s * m;
            }
            acc = s;
        }
        __syncthreads();
    }
    if (folder && lane < NF) sh.res[lane] = acc;
    if (NN > 0 && stid == 0) {
#pragma unroll
        for (int q = 0; q < NN; q++) sh.mcfin[q] = mc[q];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NF; k++) result[k] = (k < NN) ? sh.mcfin[k] * sqrt(sh.res[k]) : sh.res[k];
    __syncthreads();
}

// The TRON/CG step of one tick, reference-order numerics: one workgroup per problem (bw/Tron.java:30-179; statement order of
// tron_step_body<SEQ>).
__global__ void __launch_bounds__(RO_T)
k_ro_step(const PartDev *__restrict__ parts, ProbDev *__restrict__ probs, const int *__restrict__ qlist, int nq,
          int *__restrict__ done_counter)
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
    const bool dn = pa.dense != 0;                                // dense tile (mlx_ro_dense.h): the two l-long chains are done already
    const double *__restrict__ xtc = pr.c0f;                      // X'c of this tick, columns 0 .. nf-1 (k_colpass_lds<.., RO>)
    const double *__restrict__ coef = pr.coef;                    // the row coefficients: their sum in row order is the intercept's column
    const double *__restrict__ pvec = pr.pinv_vec;
    const double pscal = pr.pinv;
    double *__restrict__ w = pr.w, *__restrict__ w_new = pr.w_new, *__restrict__ g = pr.g, *__restrict__ s = pr.s,
           *__restrict__ d = pr.d, *__restrict__ Hd = pr.Hd;
    const double *__restrict__ m = pr.m;
    const double zero6[RO_NF] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double res[RO_NF];

    if (phase == PH_CG) {
        // ---- one trip of trcg's loop (bw/Tron.java:145-175)
        // Hd for the feature columns and the first nf terms of Tron.dot(d, Hd) on one lane; beside it, on a second lane, the intercept's
        // column of XTv: the sum of the row coefficients in row order (the bias entry closes every row). The dot's last term needs
        // that sum and is added after the pass: the same chain.
        // (dense tiles: k_ro_dense_cols ran that chain on a spare lane and left it in csump[0])
        struct RA { RoV4 d, x, p, c; };
        const int lensA[2] = {nf, dn ? 0 : l};
        ro_pass<2, 0, RA>(sh, dn ? nf : max(nf, l), lensA, zero6, res,
            [&](int j0, RA &R) { R.d = ro_ld4c(d, j0, n); R.x = ro_ld4c(xtc, j0, n); if (pvec) R.p = ro_ld4s(pvec, j0, n); R.c = ro_ld4c(coef, j0, l); },
            [&](int j0, RA &R, double (&ct)[2][4], double (&nv)[1][4]) {
                double hd[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    hd[e] = R.d.v[e] * (pvec ? R.p.v[e] : pscal) + R.x.v[e];      // Hs[i] = (s[i]*priorVar_inv[i] + Hs[i]) * 1
                    ct[0][e] = R.d.v[e] * hd[e];                                   // Tron.dot(d, Hd)
                    ct[1][e] = R.c.v[e];                                           // XTv[n-1] += v[i]
                }
                ro_st4(Hd, j0, nf, hd);
            });
        if (dn) res[1] = pr.csump[0];
        const double rTr0 = pr.rTr, delta0 = pr.delta, cgtol0 = pr.cgtol;
        const double d_icpt = d[nf];
        const double hd_icpt = d_icpt * (pvec ? pvec[nf] : pscal) + res[1];
        if (tid == 0) Hd[nf] = hd_icpt;
        __syncthreads();
        double alpha = rTr0 / (res[0] + d_icpt * hd_icpt);
        const double nalpha = -alpha;
        const double *__restrict__ rc = pr.rb[pr.rsel];
        double *__restrict__ rn = pr.rb[pr.rsel ^ 1];
        // daxpy(alpha, d, s); the norm of s; and, from the same operands, both continuations: r' = r - alpha Hd with r'.r' and |r'|
        // (:169-171, :144 of the next trip) and the three dots of the boundary case on the stepped-back s (:152-155)
        struct RB { RoV4 d, s, r, h; };
        ro_pass<6, 2, RB>(sh, n, nullptr, zero6, res,
            [&](int j0, RB &R) { R.d = ro_ld4c(d, j0, n); R.s = ro_ld4c(s, j0, n); R.r = ro_ld4c(rc, j0, n); R.h = ro_ld4c(Hd, j0, n); },
            [&](int j0, RB &R, double (&ct)[6][4], double (&nv)[2][4]) {
                double s1[4], r1[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    s1[e] = R.s.v[e] + alpha * R.d.v[e];                     // daxpy(alpha, d, s)
                    r1[e] = R.r.v[e] + nalpha * R.h.v[e];                    // daxpy(-alpha, Hd, r)
                    const double sb = s1[e] + nalpha * R.d.v[e];             // the boundary case steps back first (:153)
                    nv[0][e] = s1[e]; nv[1][e] = r1[e];
                    ct[2][e] = r1[e] * r1[e];
                    ct[3][e] = sb * R.d.v[e]; ct[4][e] = sb * sb; ct[5][e] = R.d.v[e] * R.d.v[e];
                }
                ro_st4(s, j0, n, s1);
                ro_st4(rn, j0, n, r1);
            });
        const double snorm = res[0];
        bool boundary = false, end_cg = false, nan = !(snorm == snorm);
        double alpha2 = 0.0, beta = 0.0;
        const double rnew = res[2];
        if (snorm > delta0) {
            // cg reaches trust region boundary (:150-168)
            const double std_ = res[3], sts = res[4], dtd = res[5];
            const double dsq = delta0 * delta0;
            const double rad = sqrt(std_ * std_ + dtd * (dsq - sts));
            if (std_ >= 0) alpha2 = (dsq - sts) / (std_ + rad);
            else alpha2 = (rad - std_) / dtd;
            boundary = true; end_cg = true;
        } else {
            beta = rnew / rTr0;
            if (res[1] <= cgtol0) end_cg = true;                 // loop-top test of the next trip (:144)
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
        auto emc = [&](int j0, RC &R, double (&ct)[3][4], double (&nv)[1][4]) {
            double sf[4], rf[4], dn[4], wn[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
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
        if (end_cg) ro_pass<3, 1, RC>(sh, n, nullptr, zero6, res, ldc, emc);
        else ro_pass<0, 0, RC>(sh, n, nullptr, zero6, res, ldc,
                               [&](int j0, RC &R, double (&ct)[1][4], double (&nv)[1][4]) { double c3[3][4], n1[1][4]; emc(j0, R, c3, n1); });
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
    if (dn) { res[0] = pr.lossp[0]; res[1] = pr.csump[0]; }
    else ro_pass<2, 0, RR>(sh, l, nullptr, zero6, res,
        [&](int j0, RR &R) { R.x = ro_ld4c(rowtmp, j0, l); R.c = ro_ld4c(coef, j0, l); },
        [&](int j0, RR &R, double (&ct)[2][4], double (&nv)[1][4]) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                ct[0][e] = R.x.v[e];                                         // f += weight * log(1 + exp(..)) in row order (:172-183)
                ct[1][e] = R.c.v[e];                                         // the intercept's column of XTv: the coefficients in row order
            }
        });
    double fnew = 2.0 * res[0];
    const double csum = res[1];
    const double init4[RO_NF] = {0.0, 0.0, fnew, 0.0, 0.0, 0.0};
    const double *__restrict__ c0 = pa.c0;
    const bool e0 = (phase == PH_EVAL0);
    struct RE { RoV4 w, m, x, p, c; };
    ro_pass<4, 2, RE>(sh, n, nullptr, init4, res,
        [&](int j0, RE &R) {
            R.w = ro_ld4c(w_new, j0, n); R.m = ro_ld4c(m, j0, n); R.x = ro_ld4c(xtc, j0, n);
            if (pvec) R.p = ro_ld4s(pvec, j0, n);
            if (e0) R.c = ro_ld4s(c0, j0, n);
        },
        [&](int j0, RE &R, double (&ct)[4][4], double (&nv)[2][4]) {
            double hd[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int j = j0 + e;
                const double xa = (j == nf) ? csum : R.x.v[e];
                const double pj = pvec ? R.p.v[e] : pscal;
                const double t = R.w.v[e] - R.m.v[e];
                ct[2][e] = t * t * pj;                                       // fun :187-188, added to the running f
                hd[e] = t * pj + xa;                                         // grad :224 (multiplier 1)
                nv[0][e] = hd[e];                                            // euclideanNorm(g)
                ct[3][e] = hd[e] * hd[e];                                    // r.r of the trcg call that starts from this g (r = -g)
                nv[1][e] = e0 ? (0.0 - R.m.v[e]) * pj + R.c.v[e] : 0.0;      // grad(0) (bw/Tron.java:50-53)
            }
            ro_st4(Hd, j0, n, hd);
        });
    fnew = res[2] / 2.0;
    const double gnorm_c = res[0], gsq_c = res[3];
    bool start_trcg = false, finished = false, copy_w = false, copy_g = false;
    double gnorm_cur = pr.gnorm, gsq = pr.gsq;
    if (e0) {
        // Tron prologue (:47-62)
        const double gnorm1 = res[1];
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
        ro_pass<0, 0, RT>(sh, n, nullptr, zero6, res,
            [&](int j0, RT &R) {
                R.h = ro_ld4c(Hd, j0, n);
                if (!copy_g) R.g = ro_ld4c(g, j0, n);
                if (copy_w || nullstep) R.wn = ro_ld4c(w_new, j0, n);
                if (nullstep && !copy_w) R.w = ro_ld4c(w, j0, n);
            },
            [&](int j0, RT &R, double (&ct)[1][4], double (&nv)[1][4]) {
                double z4[4], rj[4], wz[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
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
