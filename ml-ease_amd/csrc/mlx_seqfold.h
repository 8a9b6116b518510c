// mlx_seqfold.h -- a SEQUENTIAL floating-point sum, evaluated exactly, in parallel ("segmented grid fold").
//
// The reference's reductions are loops `for (i) s += t[i]` (Tron.dot, bw/Tron.java:204-213; euclideanNorm :220-252; the loss sum of fun,
// llf/LogisticRegressionL2.java:172-189): every partial sum is rounded before the next term arrives, so the result depends on the
// order and a tree gives other bits. Round 5 ran these chains literally -- one dependent v_add_f64 (10 cycles) per term, 141 K terms
// per CG tick: 0.6 ms. But the rounding a term suffers depends only on the BINADE of the running sum:
//
//   Lemma. Let s be a double with 2^e <= |s| < 2^(e+1), u = 2^(e-52) its ulp, t any double with |t| < 2^(e-1), and suppose the exact
//   s + t lies strictly inside the binade. Then fl(s + t) = s + round_u(t), where round_u rounds to the nearest multiple of u --
//   unless t sits exactly half way between two multiples (the tie is then broken by the parity of s / u, which only the chain knows).
//   round_u(t) = (t + 1.5 * 2^e) - 1.5 * 2^e for s > 0 (the magic-constant form; sign-mirrored for s < 0).
//
// So inside a stretch where the running sum stays in one binade, the chain is the EXACT sum of grid-rounded terms -- sums of multiples
// of u below 2^(e+1) are exact in any order -- and a wave can do it with a scan. The stretch ends where a prefix leaves the binade, a
// term ties, or a term is too large for the magic constant; there the literal chain runs for one sub-block and the scan resumes
// behind it on the new grid. Nothing is assumed: every sub-block's prefix minimum and maximum are checked against the binade's edges
// with the exact running sum at its start, and whatever fails (or cannot be checked: zero / subnormal / non-finite sums) is added
// literally. The result is the sequential loop's, bit for bit, for ANY input (tests/test_seqfold.py: 10^6 random and adversarial
// vectors on the host model below; tools/seqfold_selftest.hip: the wave code on the GPU).
//
// Layout: a wave takes a chunk of 64 * K consecutive terms, lane L the K terms [L K, (L + 1) K) -- a SUB-BLOCK. Terms behind the end of
// the vector are -0.0 (x + (-0.0) == x for every x, -0.0 included).
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define SGF_HD __host__ __device__ __forceinline__
#define SGF_UNROLL _Pragma("unroll")
#else
#define SGF_HD static inline
#define SGF_UNROLL
#endif

#ifndef SGF_K
#define SGF_K 16                // terms per lane and chunk (chunk = 1024 terms = mlx_ro_kernels.h's RO_CHUNK)
#endif
#define SGF_BUDGET_HOSTILE 2    // ... when the chunk before ran out of budget
#define SGF_BUDGET 24           // sub-blocks of a chunk (of 64) that may fail their check before the rest of the chunk is added literally: a failed
                                // check costs about two literal sub-blocks, so the grid pays until roughly every second sub-block fails

struct SgfGrid {
    double lo, hi;              // the binade [lo, hi) of |s|
    double magic;               // 1.5 * lo: (t + magic) - magic = t rounded to a multiple of u = lo * 2^-52
    double hu;                  // u / 2: |t - round_u(t)| == hu is a tie
    double half;                // lo / 2: the magic constant needs |t| < half
    int ok;                     // 0: |s| is zero, subnormal-adjacent, infinite or NaN -- no grid, the literal chain
};

SGF_HD double sgf_from_bits(uint64_t b) { double d; memcpy(&d, &b, sizeof d); return d; }
SGF_HD uint64_t sgf_bits(double d) { uint64_t b; memcpy(&b, &d, sizeof b); return b; }

SGF_HD SgfGrid sgf_grid(double s)
{
    SgfGrid g;
    const int ex = (int)((sgf_bits(s) >> 52) & 0x7ff);
    g.ok = ex >= 54 && ex <= 2045;      // u / 2 = 2^(ex - 1023 - 53) must be a normal number, hi = 2^(ex - 1022) finite
    const int exc = g.ok ? ex : 1023;
    g.lo = sgf_from_bits((uint64_t)exc << 52);
    g.hi = 2.0 * g.lo;
    g.magic = 1.5 * g.lo;
    g.hu = sgf_from_bits((uint64_t)(exc - 53) << 52);
    g.half = 0.5 * g.lo;
    return g;
}

struct SgfLane {
    double R;                   // sum of the sub-block's grid-rounded terms (signed: the terms are not mirrored)
    double mn, mx;              // minimum / maximum over its non-empty prefixes
    int bad;                    // a tie, a term too large for the magic constant, a NaN, or a prefix beyond the exact range
};

// One sub-block under the grid g; sg = +1.0 / -1.0 = the sign of the running sum. Nine operations per term and no comparison: what
// can go wrong is ACCUMULATED (the largest |t|; the largest |t - r| - u/2, which is zero exactly at a tie and positive wherever the
// magic constant left its binade and rounded on another grid) and judged once behind the loop -- the first form compared three times
// per term and the fold was bound by those scalar round trips. NaNs: max / min drop them, the partial sum P keeps them.
// Must be compiled without contraction / reassociation (-ffp-contract=off; the library and the tests are).
SGF_HD SgfLane sgf_prepare(const double *t, int K, double sg, const SgfGrid &g)
{
    SgfLane L;
    const double M = g.magic * sg;                  // (t + M) - M rounds t to a multiple of u, for a running sum of either sign
    double P = 0.0, mn = g.hi, mx = -g.hi, amax = 0.0, zmax = -1.0;
    SGF_UNROLL
    for (int i = 0; i < K; i++) {
        const double x = t[i];
        const double r = (x + M) - M;
        const double d = x - r;
        zmax = fmax(zmax, fabs(d) - g.hu);          // == 0: half way between two multiples of u (the chain decides); > 0: r is not round_u(x)
        amax = fmax(amax, fabs(x));
        P = P + r;
        mn = fmin(mn, P);
        mx = fmax(mx, P);
    }
    // |x| < lo / 2 keeps x + M inside the binade of M (both sides); a prefix beyond +-lo need not be exact (and no valid prefix lies there)
    L.bad = !(amax < g.half) || !(zmax < 0.0) || !(P == P) || !(mn > -g.lo) || !(mx < g.lo);
    L.R = P; L.mn = mn; L.mx = mx;
    return L;
}

// S = the exact running sum in front of the sub-block (signed): every prefix strictly inside the binade?
SGF_HD int sgf_valid(const SgfLane &L, double S, double sg, const SgfGrid &g)
{
    if (L.bad) return 0;
    return sg > 0.0 ? ((S + L.mn > g.lo) && (S + L.mx < g.hi)) : ((S + L.mx < -g.lo) && (S + L.mn > -g.hi));
}

// the literal chain over one sub-block
SGF_HD double sgf_literal(double s, const double *t, int K)
{
    for (int i = 0; i < K; i++) s = s + t[i];
    return s;
}

// ---- host model of the wave algorithm (64 lanes emulated; same decisions, same arithmetic): the CPU property test's subject, and
// the documentation of sgf_wave_fold below. t: 64 * K terms of the chunk (padded with -0.0), s: the running sum in front of it.
#if !defined(__HIP_DEVICE_COMPILE__)
static inline double sgf_model_chunk(double s, const double *t, int K, int *violations /* may be NULL */, int *literal_blocks /* may be NULL */, int *hostile)
{
    // hostile (carried from chunk to chunk of one vector): the chunk before ran out of budget -- a chain that hovers around zero, say --
    // so this one gets two tries instead of SGF_BUDGET before it falls back to the literal chain (never a matter of the result)
    int pos = 0, budget = *hostile ? SGF_BUDGET_HOSTILE : SGF_BUDGET;
    *hostile = 0;
    while (pos < 64) {
        const SgfGrid g = sgf_grid(s);
        if (budget <= 0) {
            // too many failed checks in this chunk: the chain itself for what is left of it
            for (int k = pos; k < 64; k++) { s = sgf_literal(s, t + (size_t)k * K, K); if (literal_blocks) (*literal_blocks)++; }
            pos = 64;
            *hostile = 1;
            continue;
        }
        if (!g.ok) {
            // no grid (zero / tiny / non-finite sum): sub-blocks of zeros leave the sum alone (but -0.0 + (+0.0) = +0.0); the first
            // sub-block with a term that is not a zero runs literally
            int k = pos, pz = 0;
            for (; k < 64; k++) {
                int nz = 0, p = 0;
                for (int i = 0; i < K; i++) { const double x = t[(size_t)k * K + i]; nz |= !(x == 0.0); p |= (x == 0.0 && !signbit(x)); }
                if (nz) break;
                pz |= p;
            }
            if (s == 0.0 && pz) s = 0.0;
            if (k < 64) { s = sgf_literal(s, t + (size_t)k * K, K); if (literal_blocks) (*literal_blocks)++; }
            pos = k < 64 ? k + 1 : 64;
            continue;
        }
        const double sg = (s < 0.0) ? -1.0 : 1.0;
        SgfLane L[64];
        for (int k = pos; k < 64; k++) L[k] = sgf_prepare(t + (size_t)k * K, K, sg, g);
        double S = s;                           // exclusive prefix (exact while every sub-block before was valid)
        int k = pos;
        for (; k < 64; k++) {
            if (!sgf_valid(L[k], S, sg, g)) break;
            S = S + L[k].R;
        }
        s = S;
        if (k == 64) { pos = 64; break; }
        s = sgf_literal(s, t + (size_t)k * K, K);
        if (violations) (*violations)++;
        if (literal_blocks) (*literal_blocks)++;
        pos = k + 1;
        budget--;
    }
    return s;
}

// a whole vector: chunks of 64 * K terms, the tail padded with -0.0
static inline double sgf_model_fold(double s, const double *t, size_t n, int K, int *violations, int *literal_blocks)
{
    double buf[64 * 64];
    const size_t ch = (size_t)64 * K;
    int hostile = 0;
    for (size_t base = 0; base < n; base += ch) {
        const size_t cnt = (n - base < ch) ? n - base : ch;
        for (size_t i = 0; i < ch; i++) buf[i] = (i < cnt) ? t[base + i] : -0.0;
        s = sgf_model_chunk(s, buf, K, violations, literal_blocks, &hostile);
    }
    return s;
}
#endif

// ---- the wave code ----------------------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
// inclusive prefix sum over the 64 lanes on the VALU (DPP row shifts inside rows of 16, then the row broadcasts of gfx9): each lane
// receives only sums of CONTIGUOUS lane ranges ending at it -- the property the exactness argument needs
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double sgf_dpp_or_zero(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sgf_wave_inclusive_scan(double v)
{
#pragma clang fp contract(off)
    v = v + sgf_dpp_or_zero<0x111, 0xF>(v);      // row_shr:1 (lanes without a source read 0)
    v = v + sgf_dpp_or_zero<0x112, 0xF>(v);      // row_shr:2
    v = v + sgf_dpp_or_zero<0x114, 0xF>(v);      // row_shr:4
    v = v + sgf_dpp_or_zero<0x118, 0xF>(v);      // row_shr:8
    v = v + sgf_dpp_or_zero<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
    v = v + sgf_dpp_or_zero<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double sgf_bcast(double x, int src)       // lane `src` (wave-uniform) to every lane
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// s: the running sum in front of the chunk (the same value in every lane); hostile: see below; t[K]: this lane's sub-block (terms behind the vector's end
// = -0.0). Returns the running sum behind the chunk in every lane. All 64 lanes must be active. MUL: the euclideanNorm form -- a
// sub-block with has_mul set holds updates `sum = c + sum * m` (m != 1 where the running scale changes): never on the grid.
// (timing-experiment builds, tools/ablate_build.sh -DMLX_PHASE_TIMING -DMLX_SGF_STATS: how often the checks fail -- g_phase[13..15] in mlx_kernels.hip)
#ifdef MLX_SGF_STATS
// (per pass of k_ro_step: g_sgf_pass = 0 CG pass A (d.Hd), 1 CG pass B (r'.r'), 2 the other CG passes, 3 EVAL row folds, 4 EVAL pass over n;
//  g_phase[3 pass + (slot - 13)]: literal sub-blocks (no grid / budget), failed checks, grid iterations)
extern __device__ unsigned long long g_phase[16];
__shared__ int g_sgf_pass;
#define SGF_COUNT(slot) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_phase[3 * g_sgf_pass + ((slot) - 13)], 1ull); } while (0)
#define SGF_PASS(p) do { __syncthreads(); if (threadIdx.x == 0) g_sgf_pass = (p); __syncthreads(); } while (0)
#else
#define SGF_COUNT(slot) do { } while (0)
#define SGF_PASS(p) do { } while (0)
#endif
template <int K, bool MUL>
__device__ __forceinline__ double sgf_wave_fold(double s, const double (&t)[K], const double (&m)[K], bool has_mul, int &hostile)
{
#pragma clang fp contract(off)
    const int lane = (int)(threadIdx.x & 63);
    // hostile (the caller carries it from chunk to chunk of one vector): the chunk before ran out of budget -- a chain hovering around
    // zero changes binade all the time -- so this one gets two tries before the literal chain; a failed check costs about as much as
    // four literal sub-blocks, and round 6's first form spent 24 of them per chunk on such chains (the intercept's column of X'c)
    int pos = 0, budget = hostile ? SGF_BUDGET_HOSTILE : SGF_BUDGET;
    hostile = 0;
    // sub-block k's chain on its own lane, then to everybody. (Every lane running it on lane k's terms read as scalars was measured:
    // v_readlane with a run-time lane select costs ~70 cycles per term, profiles/r6_notes.md.)
    auto literal = [&](int k) {
        double x = s;
        if (lane == k) {
#pragma unroll
            for (int i = 0; i < K; i++) x = MUL ? t[i] + x * m[i] : x + t[i];
        }
        s = sgf_bcast(x, k);
    };
    const unsigned long long mulmask = MUL ? __ballot(has_mul) : 0ull;
    SgfLane L;
    L.R = 0.0; L.mn = 0.0; L.mx = 0.0; L.bad = 1;
    double lo_have = 0.0, sg_have = 0.0;
    while (pos < 64) {
        const SgfGrid g = sgf_grid(s);
        if (budget <= 0) {                        // too many failed checks in this chunk: the chain itself for what is left of it
            for (int k = pos; k < 64; k++) { literal(k); SGF_COUNT(13); }
            pos = 64;
            hostile = 1;
            continue;
        }
        if (!g.ok) {
            // No grid: the sum is zero, next to the subnormals, infinite or NaN. Sub-blocks that hold only zeros leave it alone -- except
            // that -0.0 + (+0.0) = +0.0 -- so skip to the first sub-block with a term that is not a zero and run that one literally.
            // (Sums that stay zero for a whole vector are common: the boundary dots of the first CG step, s = 0.)
            bool nz = MUL && has_mul, pz = false;
#pragma unroll
            for (int i = 0; i < K; i++) { nz = nz || !(t[i] == 0.0); pz = pz || (t[i] == 0.0 && !signbit(t[i])); }
            const unsigned long long nzm = __ballot(nz && lane >= pos);
            const int k = nzm ? __ffsll((long long)nzm) - 1 : 64;
            const unsigned long long pzm = __ballot(pz && lane >= pos && lane < k);
            if (s == 0.0 && pzm != 0ull) s = 0.0;
            if (k < 64) { literal(k); SGF_COUNT(13); }
            pos = k < 64 ? k + 1 : 64;
            continue;
        }
        SGF_COUNT(15);
        const double sg = (s < 0.0) ? -1.0 : 1.0;
        if (g.lo != lo_have || sg != sg_have) { L = sgf_prepare(t, K, sg, g); lo_have = g.lo; sg_have = sg; }      // (a tie leaves the grid as it is)
        // (a sub-block flagged bad contributes nothing: its R may be inexact, and the first failing lane's own prefix must stay exact;
        //  incl - Rk: both multiples of u below 2 lo while the lanes before are valid: exact)
        const double Rk = (lane >= pos && !L.bad) ? L.R : 0.0;
        const double incl = sgf_wave_inclusive_scan(Rk);
        const double S = s + (incl - Rk);
        const bool inval = lane >= pos && (!sgf_valid(L, S, sg, g) || (MUL && has_mul));
        const unsigned long long bad = __ballot(inval);
        if (bad == 0ull) { s = sgf_bcast(S + L.R, 63); pos = 64; break; }
        const int k = __ffsll((long long)bad) - 1;
        s = sgf_bcast(S, k);                      // exact in front of sub-block k
        literal(k);
        SGF_COUNT(14);
        pos = k + 1;
        // (a sub-block that only holds a scale change of euclideanNorm is not a failed check: the budget is for chains that will not settle)
        if (!(MUL && ((mulmask >> k) & 1ull))) budget--;
    }
    return s;
}
#endif
