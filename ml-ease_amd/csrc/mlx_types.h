// mlx_types.h -- device-visible descriptors shared by the kernels and the C-ABI host code.
//
// HBM layout (DESIGN.md "Data layout"): per local partition one PartDev (row data, uploaded once,
// shared by every lambda and every ADMM iteration); per (partition, lambda) problem one ProbDev
// (TRON scalars + pointers to its n_local-sized fp64 work vectors).
#pragma once
#include <stdint.h>

enum { PH_DONE = 0, PH_EVAL0 = 1, PH_CG = 2, PH_EVAL = 3 };
enum { ST_OK = 0, ST_NAN = 1, ST_TICKCAP = 2, ST_SYNC = 3 /* a work unit of the one-launch reference-order column pass gave up waiting for its problem's earlier row blocks */ };
enum { STEP_NP = 8 };       // doubles per workgroup and phase in ProbDev::pA/pB/pC

struct PartDev {
    int32_t l;             // rows
    int32_t n_local;       // features incl. intercept (intercept = local index n_local-1, implicit column of 1.0)
    int32_t n_feat;        // n_local - 1
    int32_t dense;         // 1: X dense tile, 0: CSR+CSC
    int32_t nblk;          // partial sums per X pass over this partition (sizes the partial buffers): dense = row units, CSR = row chunks
    int32_t rows_per_blk;  // rows per unit / chunk
    int32_t n_rowparts;    // loss / coefficient-sum partials a pass leaves in lossp / csump: dense = nblk (per unit), sliced CSR on the
                           // tick kernels = n_rgroups (per 64-row group), otherwise nblk (per chunk of the fallback / one-launch kernels)
    int32_t n_units;       // dense: row units of rows_per_blk rows
    int32_t units_per_wg;  // dense: 1 or 2 consecutive units per workgroup of the pass (set at finalize from the handle's work). nblk =
                           // stored partials = workgroups: unit sums (1) or pair sums (2); units are always ADDED in pairs, so the
                           // result does not depend on it
    int32_t pos, neg;      // #y==+1, #y==-1 (llf/LibLinear.java:272-276)
    int64_t ld;            // dense row stride in floats (multiple of 4, zero padded)
    int64_t nnz;
    const float *X;        // dense [l][ld]
    const int32_t *rp;     // CSR row pointer [l+1] (intercept excluded)
    const int32_t *ci;     // CSR local column ids [nnz]
    const float *val;      // CSR values [nnz] or nullptr (binary.feature)
    // Column side (X'c). Rows are cut into n_rblk row blocks of rblk_rows rows (one block's slice of the row
    // coefficients fits in LDS); the entries of column j inside block b form 0+ work items of <= SEG entries, rows
    // ascending. Items are numbered block-major, each block padded to a multiple of 64 items (padding items are empty).
    const int32_t *cri;        // [nnz] row ids in item order (block-major, column-major inside a block)
    const float *cval;         // [nnz] values in the same order, or nullptr
    const int32_t *item_ptr;   // [n_items+1] entry offsets of the items
    // Item sums are STORED column-major: item t writes its sum to slot item_dst[t] (-1 = empty/padding item), and the slots
    // of column j are col_ptr[j] .. col_ptr[j+1]-1 in (block, segment) order -- one range per column whatever n_rblk is, and
    // consecutive columns read consecutive slots.
    const int32_t *item_dst;   // [n_items]
    const int32_t *col_ptr;    // [n_feat+1]
    int32_t n_items;           // incl. padding items
    int32_t n_slots;           // real items = col_ptr[n_feat]
    int32_t n_rblk, rblk_rows;
    const int32_t *items_short;  // real item ids with <= 64 entries (8-lane groups; fallback kernels)
    const int32_t *items_long;   // real item ids with 65..SEG entries (one wave each)
    int32_t n_short, n_long;
    // Sliced-ELL copies for the LDS passes; sell != 0 when built (row-side padding <= 2x nnz).
    // Row side: slices 0..n_hs-1 = the n_hs*slw most frequent columns, slw each (gathered from LDS), the later slices = 65 535
    // columns each (gathered from global memory), n_rgroups groups of 64 rows; block (s, g) at rs_ptr[s*n_rgroups + g] holds, in packs of 4 per lane,
    // the slice-local uint16 column ids of the 64 rows' entries in slice s, padded with slw (slice 0: the LDS slot behind
    // the staged slice, which holds 0.0) / 0xFFFF (cold slices).
    int32_t sell;
    int32_t n_cs, slw, n_rgroups, n_cslices;
    int32_t n_hs;              // the first n_hs row-side slices are hot (slw columns each, staged in LDS); the rest are cold
    int32_t rgroups_per_chunk; // row groups one row-pass workgroup owns (set at finalize); nblk = chunks
    const int32_t *rs_ptr;     // [n_cs*n_rgroups + 1] entry offsets
    const uint16_t *rs_idx;
    const float *rs_val;       // values in the same layout or nullptr (binary.feature)
    // Column side: slice s = items 64s .. 64s+63 (one row block each), entry-major; row ids RELATIVE to the item's row
    // block as uint16 (index into the LDS-staged coefficients), padded with rblk_rows (zero slot).
    const int32_t *cs_ptr;     // [n_cslices+1]
    const uint16_t *cs_idx;
    const float *cs_val;
    int32_t n_cunits;          // work units of the LDS column pass: unit u = slices [cw_slice[u], cw_slice[u+1]) of block cw_blk[u]
    const int32_t *cw_blk;     // [n_cunits]
    const int32_t *cw_slice;   // [n_cunits+1]
    // Reference-order numerics (MLX_NUMERICS_REFERENCE_ORDER, mlx_ro_kernels.h): a column's sum is ONE sequential chain over its
    // rows (XTv, llf/LogisticRegressionL2.java:140-145). Items are unsplit inside a row block; the item of block b starts from its own
    // hand-over slot parts[t] -- the sum its column had reached in the earlier blocks, or the 0.0 of the slab's memset where no earlier
    // item hands a sum on -- and ONE word per item says where its sum goes (round 6; three words before): >= 0 the column's item in the
    // next block it has rows in (its slot), ~column for the column's last item (X'c of that column, stored densely), INT32_MIN for padding.
    const int32_t *item_init;  // (unused since round 6: nullptr)
    const int32_t *item_last;  // (unused since round 6: nullptr)
    const int32_t *item_chain; // [n_items] or nullptr: the word described above
    int32_t rowgroup;      // lanes per row in the CSR row pass (8..64)
    const int8_t *y;       // +1/-1
    const float *wt;       // instance weight
    const float *off;      // offset
    const int32_t *l2g;    // local -> global
    const double *c0;      // X' t0 with t0_i = wt_i (sigma(y_i off_i) - 1) y_i : data part of grad(0), fixed per partition
};

struct ProbDev {
    int32_t part;          // local partition index
    int32_t lambda_idx;
    int32_t phase;
    int32_t dsel;          // which wd[] buffer belongs to the last accepted point
    int32_t iter;          // Tron's iter (accepted steps + 1)
    int32_t max_iter;
    int32_t cg_iter;       // CG steps of the current trcg call
    int32_t newton, accepted, cg_total, ticks;
    int32_t status;
    double f, delta, gnorm, gnorm1, eps, rTr, cgtol, prered, gs;
    double pinv;           // scalar prior precision 1/(1/rho) (used when pinv_vec == nullptr)
    const double *pinv_vec;    // per-coordinate 1/priorVar (mlx_solve_one) or nullptr
    double *w, *w_new, *g, *s, *r, *d, *Hd, *m;   // n_local each
    // multi-workgroup step (k_step_a/b/c, CSR tick path): the residual is double buffered (rb[rsel] = r of the current CG
    // step, rb[rsel^1] receives r - alpha*Hd, so the trust-region boundary case can still see the old r), per-workgroup
    // partial sums of the three phases, and the scalars the fused phases carry from one tick to the next.
    double *rowtmp, *c0f;  // reference-order numerics only: per-row losses [l]; [n_local]: grad(0)'s data part (one-launch kernel) /
                           // the dense X'c the chained column pass leaves (tick kernels: xtc)
    double *rb[2];
    int32_t rsel;
    double *pA, *pB, *pC;  // [n_step_wg][STEP_NP] each
    double gsq;            // sum g_j^2 at the last accepted point (= rTr of the next trcg call)
    double head;           // k_step_head: sum of the first STEP_HEAD terms of the dot phase A is about to add (its rounding grid)
    double snorm;          // ||s|| at the end of the last trcg call
    double *wd[2];         // [l] wt_i * D_i at the accepted / trial point
    double *coef;          // [l] row coefficients of the current pass (CSR path only)
    double *parts;         // dense: [nblk][n_local] partial X'c ; CSR: itemsum [n_items]
    double *lossp;         // [nblk] partial loss sums
    double *csump;         // [nblk] partial sums of coef (CSR intercept column)
};

struct ConsDev {           // consensus parameters per lambda
    double weight;         // L2: (double)(float)(N*rho/(lambda+N*rho)) ; L1: lambda/(rho*N)
};
