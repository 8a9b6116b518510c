// mlx_kernels.h -- launch wrappers of the gfx950 kernels in mlx_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PartDev;
struct ProbDev;

// One X pass for the problems in qlist (device array of problem indices). Returns -1 if unsupported width.
int mlxk_xpass_dense(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int maxblk,
                     int max_nfeat, bool stream_once);
int mlxk_xpass_csr(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int maxblk,
                   int max_short, int max_long, int rowgroup, bool hasval, bool sell, int max_cunits, int max_rblk_rows, int row_slw, int row_ngc, bool stream_once,
                   int which /* 1 = row pass, 2 = column pass, 3 = both */,
                   int cold_groups /* > 0: the row pass's cold slices run as their own launch (k_rowcold) over that many row groups */,
                   int ro_blocks /* > 0: reference-order numerics (mlx_ro_kernels.h); the column pass runs once per row block, that many times */,
                   int ro_units_blk = 0 /* ... and the most column work units any one row block of a partition has: the grid of one block's launch */,
                   int *coldone = nullptr /* reference order: [total problems] ints -> all row blocks of the column pass in ONE launch (the row pass clears a problem's counter, the column pass's units count themselves) */);
// reference-order numerics on DENSE tiles (mlx_ro_dense.h): which & 1 = Xv, one lane per row; which & 2 = XTv, one lane per column
// over all rows (+ the intercept's column and the loss sum as two more chains)
// claim: two zeroed ints of device memory per concurrently running launch (the column kernel's work counter; it clears them itself)
void mlxk_ro_dense_passes(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int max_l, int max_nfeat,
                          bool stream_once, int which, int *claim);
// TRON/CG step of the reference-order numerics: one workgroup per problem, every reduction folded in index order
void mlxk_ro_step(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int *done_counter, bool exact_norms);
// TRON/CG control flow for the problems in qlist: one workgroup per problem (dense tiles)
void mlxk_tron_step(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int threads,
                    int *done_counter);
// the same for CSR problems, split over column chunks of `ch` columns (max_nwg chunks for the widest problem):
// four launches per tick, which = 0 (A), 1 (B), 2 (C), 3 (commit)
// emu: the dots Tron.dot computes sequentially (d.Hd, r.r) as grid-rounded sums (grid_of_sum in mlx_kernels.hip)
void mlxk_step_phase(hipStream_t st, int which, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, int ch,
                     int max_nwg, int *done_counter, bool emu);
// whole solves of small CSR problems in one launch (one workgroup per problem runs the tick loop)
// (qlist: the problems to solve, one workgroup each)
void mlxk_solve_small(hipStream_t st, const PartDev *parts, ProbDev *probs, const int *qlist, int nq, bool hasval,
                      int max_ticks, int *done_counter, int lds_doubles, bool faithful, int xl, int lds_bytes_xl, bool grid_dots);
void mlxk_collect_c0(hipStream_t st, const PartDev *parts, const ProbDev *probs, const int *qlist, int nq,
                     double *const *c0_ptrs, bool ro);
void mlxk_setup(hipStream_t st, const PartDev *parts, ProbDev *probs, int nprob, int n_lambda, int n_global,
                int max_nlocal, const float *z32, const float *u, const double *pinv_l, double epsilon, int max_iter);
void mlxk_setup_naive(hipStream_t st, const PartDev *parts, ProbDev *probs, int nprob, int max_nlocal,
                      const double *pinv_l, const double *pinv_ovr, double *pinv_buf, double prior_mean,
                      double epsilon, int max_iter);
void mlxk_outputs_naive(hipStream_t st, const PartDev *parts, const ProbDev *probs, int nprob, int n_lambda,
                        int n_global, int max_nlocal, float *B);
void mlxk_outputs(hipStream_t st, const PartDev *parts, const ProbDev *probs, int nprob, int n_lambda, int n_global,
                  int max_nlocal, bool any_absent, const float *z32, const float *u, float *B, float *UPX);
void mlxk_partial_means(hipStream_t st, int nlocal, int n_lambda, int n_global, double invN, const float *B,
                        const float *u, double *xbar, double *ubar);
void mlxk_z_update(hipStream_t st, int n_lambda, int n_global, int regularizer, int penalize_intercept,
                   const double *weight_l, const double *cmap, const double *xbar, const double *ubar, double *Z,
                   float *z32, unsigned long long *diffbits);
void mlxk_u_update(hipStream_t st, int nlocal, int n_lambda, int n_global, const float *UPX, const double *Z, float *u);
void mlxk_test_loglik(hipStream_t st, int l, int n_lambda, int n_global, const int64_t *rp, const int32_t *gi,
                      const double *val, const int8_t *y, const double *wt, const double *off, const double *Z, double *part);
void mlxk_round_z(hipStream_t st, int64_t n, const double *Z, float *z32);

// posterior variance (LibLinear.train computePosteriorVar): densify a CSR partition, weighted column sums, fp64-MFMA Gram
void mlxk_densify(hipStream_t st, int l, const int32_t *rp, const int32_t *ci, const float *val, float *X, int64_t ld);
void mlxk_hess_colsums(hipStream_t st, const float *X, int64_t ld, int l, const double *wd, double *part, int nchunk,
                       int rows_per_chunk, double *out /* [2*ld + 1]: s1, s2, sum wd */);
void mlxk_hess_diag_items(hipStream_t st, int n_items, const int32_t *item_ptr, const int32_t *item_dst, const int32_t *cri,
                          const float *cval, const double *wd, double *slots /* [n_slots] */);
void mlxk_gram_f64(hipStream_t st, const float *X, int64_t ld, int l, const double *wd, const int *blocks_xy, int nblocks,
                   int ksplit, int rows_per_split, double *P, int npad, int nf /* column nf = implicit ones (intercept) */);
void mlxk_gram_finish(hipStream_t st, const double *P, int ksplit, int npad, int nf, const double *pinv, double *H);
// one idle wave for `ticks` of the wall clock (stream / hardware-queue probe of mlx_api.hip)
void mlxk_spin(hipStream_t st, long long ticks);
// RegressionTest scoring: one float prediction per row
void mlxk_score_rows(hipStream_t st, int l, const int64_t *rp, const int32_t *gi, const double *val, const double *off,
                     const double *z, double base, float *pred);
