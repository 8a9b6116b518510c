/*
 * mlease_admm.h -- C-ABI of the MI355X-native ADMM L2-logistic trainer (libmlease_hip.so).
 *
 * Drop-in boundary for ONE path of linkedin/ml-ease (seam S3):
 * everything between "launch the AdmmMapper/AdmmReducer MapReduce job" and "z, u and
 * maxdiff of this iteration are known" inside RegressionAdmmTrain.run, i.e.
 *
 *   AvroUtils.runAvroJob(conf)            jobs/RegressionAdmmTrain.java:357
 *     AdmmReducer.reduce                  jobs/RegressionAdmmTrain.java:641-718
 *       LibLinear.train                   liblinearfunc/LibLinear.java:221-398
 *         Tron.tron / trcg                de/bwaldvogel/liblinear/Tron.java:30-179
 *           LogisticRegressionL2.fun/grad/Hv  liblinearfunc/LogisticRegressionL2.java:156-248
 *   LinearModelUtils.meanModel x2         jobs/RegressionAdmmTrain.java:362-364
 *   z-update (L2)                         jobs/RegressionAdmmTrain.java:365-405
 *   computeU                              jobs/RegressionAdmmTrain.java:736-765
 *   maxdiff / mindiff                     jobs/RegressionAdmmTrain.java:455-472
 *
 * The reference has no FFI of its own (100 % Java); these entry points are what a JNI class
 * bound into a patched RegressionAdmmTrain.run would call (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - C linkage, plain pointers and sizes, no exceptions cross the boundary.
 *   - Every function returns MLX_OK (0) or a negative MLX_ERR_* code; mlx_last_error(h) gives
 *     the message. The Java shim maps any non-zero code to IOException("Model fitting error!")
 *     like jobs/RegressionAdmmTrain.java:713-716.
 *   - The caller owns all host buffers and may free them when the call returns; the library
 *     owns device memory behind the opaque handle.
 *   - One handle drives ONE GPU; one host thread per handle; calls are blocking; a handle is not
 *     thread-safe; distinct handles are independent (multi-GPU = one handle per GPU, partitions
 *     k -> rank k mod G, consensus means summed across handles, see mlx_admm_solve_local).
 *   - Index spaces: GLOBAL coefficient index j in [0, n_global); the intercept "(INTERCEPT)"
 *     (liblinearfunc/LibLinearDataset.java:92) is ALWAYS global index n_global-1. LOCAL index in
 *     [0, n_local) per partition, the partition's first-seen feature order
 *     (liblinearfunc/LibLinearDataset.java:467-478); the intercept is local index n_local-1 and is NOT
 *     stored in the row data: the library appends it with value bias = 1.0 exactly as
 *     LibLinearDataset.finish does (liblinearfunc/LibLinearDataset.java:592-615).
 *   - lambda order: lambda[] must be sorted ascending; problem (partition k, lambda index li) is
 *     the reduce task with key k*n_lambda+li (jobs/RegressionAdmmTrain.java:636-638,647-650).
 */
#ifndef MLEASE_ADMM_H
#define MLEASE_ADMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mlx_context *mlx_handle;

enum {
    MLX_OK = 0,
    MLX_ERR_INVALID = -1,        /* bad argument / call order */
    MLX_ERR_HIP = -2,            /* a HIP runtime call failed */
    MLX_ERR_NO_DEVICE = -3,      /* no usable gfx950 device */
    MLX_ERR_MODEL_FITTING = -4,  /* a solve did not terminate (tick cap) or produced NaN */
    MLX_ERR_MISSING_MODELS = -5, /* "Some models failed!" (utils/LinearModelUtils.java:80-83) */
    MLX_ERR_COMM = -6            /* RCCL failure */
};

/* Counters of one mlx_admm_iterate / mlx_admm_solve_local call (this handle's partitions only). */
typedef struct mlx_stats {
    double maxdiff;          /* max_lambda ||z - z_prev||_inf  (jobs/RegressionAdmmTrain.java:455-472) */
    double mindiff;          /* min_lambda ...                                                         */
    int64_t solves;          /* (partition, lambda) problems solved = LibLinear.train calls            */
    int64_t newton_iters;    /* trcg calls (accepted + rejected TRON iterations)                       */
    int64_t accepted;        /* accepted TRON steps                                                    */
    int64_t cg_iters;        /* CG steps = Hv evaluations                                              */
    int64_t x_passes_ref;    /* passes over X the REFERENCE would make: 3 + sum(2cg+1+acc) per solve   */
    int64_t x_passes_dev;    /* passes over X this library made: 1 + sum(cg+1) per solve               */
    int64_t ticks;           /* lock-step device ticks (one X pass of every unfinished problem)        */
    double alg_bytes_dev;    /* algorithmic HBM bytes of the X-pass kernels launched       */
    double xpass_ms;         /* profiling on: SUM of the X-pass launches' own mark-to-mark intervals (HIP events on the   */
                             /* launch's tick stream; with two tick streams the intervals of the halves overlap and       */
                             /* include queue wait -- use *_busy_ms for bandwidth, or mlx_set_profiling(h, 2)); else 0    */
    double total_ms;         /* device time of the whole call (HIP events)                             */
    int64_t xpass_launches;  /* dense-pass + row-pass launches of the call: per tick one per tick stream and storage      */
                             /* class (two tick streams: 2 per tick); 0 for the one-launch solves of small CSR problems   */
    double rowpass_ms;       /* CSR path, profiling on: summed intervals of the row-pass launches      */
    double colpass_ms;       /*   ... of the column-pass launches (xpass_ms = dense + row + column)    */
    double step_ms;          /*   ... of the TRON/CG step launches                                     */
    /* With several tick streams the launches of a class overlap each other: the *_ms fields above sum the launches' own       */
    /* durations (what a kernel trace lists); the *_busy_ms fields are the time during which AT LEAST ONE launch of the class  */
    /* was running (union of the intervals on the device clock). One tick stream: busy == ms.                                  */
    double xpass_busy_ms;    /* all X-pass classes together                                            */
    double rowpass_busy_ms, colpass_busy_ms, step_busy_ms;
} mlx_stats;

/* ---- lifetime ---------------------------------------------------------------------------- */
int mlx_create(int device_id, mlx_handle *out);
int mlx_destroy(mlx_handle h);
const char *mlx_last_error(mlx_handle h);          /* valid until the next call on h; h may be NULL */
/* Run all work of this handle on an existing hipStream_t (e.g. a torch side stream); NULL = the handle's own stream (what
 * mlx_create made; torch's DEFAULT stream is the NULL pointer, so passing it means "own stream" too).
 * The handle ticks the halves of its problem list on two streams. The HIP runtime multiplexes a process's streams onto a few
 * hardware queues, and two streams of one queue run in order: mlx_create / mlx_set_stream therefore test the pair (one idle 60 us
 * wave on each: do they overlap?) and re-create the second stream until it sits on another queue (MLX_NO_STREAM_PROBE=1: no test).
 * Measured without the test: an 8-problem handle 1.8 k instead of 2.8 k solves/s whenever the pair shared a queue -- which happened
 * or not depending on the other streams alive in the process (profiles/r4_notes.md). */
/* The test launches two idle waves and SYNCHRONIZES both streams: on a caller-owned stream it is skipped when that stream is
 * capturing a graph or still has work queued (the pair is then used untested; mlx_get_option "tick_streams" /
 * "stream_probe_rejects" report what the handle runs on). */
int mlx_set_stream(mlx_handle h, void *hip_stream);
/* 1 = time the launch classes with HIP events (stats.*_ms / *_busy_ms; one mark per launch class and tick stream);
 * 2 = the same with ALL ticks on one stream, so that a launch's duration is the kernel's alone (measurement only: slower). */
int mlx_set_profiling(mlx_handle h, int enable);

/* ---- numerics contract and other per-handle behaviour a host chooses -----------------------------------------------
 * MLX_NUMERICS_FAST (default): every reduction of the reference -- the row / column sums of Xv / XTv
 * (liblinearfunc/LogisticRegressionL2.java:115-150), Tron.dot and euclideanNorm (de/bwaldvogel/liblinear/Tron.java:204-252), the loss
 * sum of fun (:172-189) -- is a fixed parallel tree (the two dots of a CG step grid-rounded); everything else
 * is the reference's arithmetic statement for statement. Results are bit-reproducible and independent of the GPU count, and
 * differ from the Java code by summation order only.
 * MLX_NUMERICS_REFERENCE_ORDER: those reductions run as the reference's SEQUENTIAL loops -- feature ids and rows in the caller's
 * order (liblinearfunc/LibLinearDataset.java:464-482), a row's entries in ascending id, a column's entries in row order, dots and
 * norms in index order -- on the same tick kernels (csrc/mlx_ro_kernels.h), with exp / log1p evaluated by portable +,-,*,/
 * sequences (csrc/portable_math.h; device and host libm differ in the last bit). Every output is then bit-identical to the
 * reference algorithm evaluated with the same elementary functions (oracle/liboracle_pm.so). Slower (the sequential chains).
 * Dense tiles stay tiles (round 6, csrc/mlx_ro_dense.h): one lane per row for Xv, one lane per column walking all rows for XTv --
 * two reads of the tile per tick where the fast contract reads it once; the zeros a tile holds add +-0.0 to a running sum, which
 * keeps every bit, so the sums equal the entry-by-entry sums over the non-zeros.
 * MLX_NUMERICS_REFERENCE_ORDER_ONE_LAUNCH: the same arithmetic on the one-workgroup-per-problem verification kernel (one thread per
 * reduction; orders of magnitude slower) -- the independent cross-check of the mode above; one row block per partition.
 * Call before the first mlx_add_partition_* (the HBM layout depends on it). The Java driver's job key: mlease.numerics. */
enum { MLX_NUMERICS_FAST = 0, MLX_NUMERICS_REFERENCE_ORDER = 1, MLX_NUMERICS_REFERENCE_ORDER_ONE_LAUNCH = 2 };
int mlx_set_numerics(mlx_handle h, int32_t mode);
/* String-keyed options, so that a host can choose per handle what the MLX_* environment variables choose per process (those only
 * seed the defaults at mlx_create; no reference counterpart -- the Java job passes everything through JobConf strings likewise):
 *   "numerics"            fast | reference_order | reference_order_one_launch          (= mlx_set_numerics)
 *   "tick_streams"        1..4   HIP streams the halves of the problem list tick on (default 2; re-tests the hardware queues)
 *   "stream_probe"        0 | 1  test that the tick streams sit on different hardware queues (default 1)
 *   "grid_rounded_dots"   0 | 1  d.Hd / r.r of every CSR solve (tick kernels and the one-launch solver of small partitions) as grid-rounded sums (default 1; 0 = plain trees, A/B and tests)
 *   "ro_exact_norms"      0 | 1  reference-order numerics: a CG step runs euclideanNorm's recurrence for both of its norm TESTS always (default 0: only
 *                                when the sum of squares it has anyway lies too close to the threshold to decide the test -- the outcome is the same by
 *                                construction, the tests compare the two settings bit for bit)
 *   "profile_one_stream"  0 | 1  with profiling on, all ticks on one stream (= mlx_set_profiling(h, 2))
 *   "one_launch_small"    0 | 1  small CSR problems solve in one launch (default 1); before mlx_finalize
 *   "small_ticks"         ticks one such launch may run before the host looks (default 16384)
 *   "comm_always"         0 | 1  run the RCCL exchange at nranks == 1 too (tests)
 *   "trace"               0 | 1  tick progress and stream-probe results on stderr
 * mlx_get_option additionally answers "numerics_kernels" (after mlx_finalize: fast | reference_order_ticks |
 * reference_order_one_launch: what the handle actually runs), "dense_tiles" (how many of the handle's partitions are stored as dense
 * tiles), "stream_probe_rejects" and "tick_log": the last solve's batches of
 * four lock-step ticks as "ticks:done:us;..." -- ticks queued, problems finished (read one batch late) and microseconds since the
 * solve's first launch when the GPU had finished that batch (monitoring: how the active set shrinks over a solve; give a buffer
 * of a few KB). Unknown key: MLX_ERR_INVALID. */
int mlx_set_option(mlx_handle h, const char *key, const char *value);
int mlx_get_option(mlx_handle h, const char *key, char *out, size_t out_len);

/* ---- problem definition -------------------------------------------------------------------
 * num_blocks  = num.blocks of the job: the FIXED divisor of the consensus mean
 *               (consumers/MeanLinearModelConsumer.java:61), also when this handle holds only a shard.
 * lambda/rho  = float32 as parsed by the driver (jobs/RegressionAdmmTrain.java:164-183), ascending lambda.
 * lambda_map  = NULL, or per-global-feature lambda (NaN = use the global lambda), the dense form of
 *               the lambda.map file (jobs/RegressionAdmmTrain.java:383-386); entry n_global-1 ignored.
 * Replaces: JobConf keys num.blocks/lambda/rho/penalize.intercept (jobs/RegressionAdmmTrain.java:138-185,302). */
int mlx_set_problem(mlx_handle h, int32_t n_global, int32_t n_lambda, const float *lambda,
                    const float *rho, int32_t num_blocks, int32_t penalize_intercept,
                    const float *lambda_map);

/* regularizer: 2 = L2 consensus shrinkage (default; jobs/RegressionAdmmTrain.java:378-405),
 * 1 = L1 iterative thresholding (:406-451). Call between mlx_set_problem and mlx_finalize. */
int mlx_set_regularizer(mlx_handle h, int32_t regularizer);

/* One partition = the rows one AdmmReducer receives (jobs/RegressionAdmmTrain.java:686-690), uploaded
 * ONCE and shared by all lambdas and all iterations. Replaces LibLinearDataset.addInstanceAvro+finish
 * (liblinearfunc/LibLinearDataset.java:413-484,586-658; binary: LibLinearBinaryDataset.java:426-515).
 *   row_ptr[l+1], col_idx[nnz]: CSR, 0-based local ids in [0, n_local-1), sorted per row, intercept excluded
 *   val[nnz]   float32 feature values, or NULL for binary.feature (all 1)
 *   y[l]       +1 / -1  (response 0 and -1 both map to -1, LibLinearDataset.java:419-423)
 *   weight[l], offset[l] float32 (RegressionPrepareOutput.avsc:28-33); NULL = all 1 / all 0
 *   local_to_global[n_local] with local_to_global[n_local-1] == n_global-1
 * partition_id is the GLOBAL id in [0, num_blocks); a handle may hold any subset.
 * Storage is the library's choice: rows that are at least 30 % filled (n_local <= 2049, strictly increasing column ids,
 * more than 65536 non-zeros) are kept as a dense tile exactly as if mlx_add_partition_dense had been called. */
int mlx_add_partition_csr(mlx_handle h, int32_t partition_id, int32_t l, int32_t n_local, int64_t nnz,
                          const int64_t *row_ptr, const int32_t *col_idx, const float *val,
                          const int8_t *y, const float *weight, const float *offset,
                          const int32_t *local_to_global);

/* `count` partitions in one call: identical in effect to `count` calls of mlx_add_partition_csr in argument order, but the
 * host-side preparation (frequency relabelling, column items, sliced copies) of different partitions runs on a pool of
 * threads; uploads follow sequentially. Every argument is an array of `count` entries; val, weight, offset may be NULL
 * as a whole or per entry. */
int mlx_add_partitions_csr(mlx_handle h, int32_t count, const int32_t *partition_id, const int32_t *l, const int32_t *n_local,
                           const int64_t *nnz, const int64_t *const *row_ptr, const int32_t *const *col_idx,
                           const float *const *val, const int8_t *const *y, const float *const *weight,
                           const float *const *offset, const int32_t *const *local_to_global);

/* Dense tile form for partitions in which every row carries every feature (BASELINE config #2):
 * X is row-major [l][ld] float32, the first n_feat columns used; n_local = n_feat+1.
 * x_on_device != 0: X, y, weight, offset are DEVICE pointers (synthetic data generated on the GPU). */
int mlx_add_partition_dense(mlx_handle h, int32_t partition_id, int32_t l, int32_t n_feat, int64_t ld,
                            const float *X, const int8_t *y, const float *weight, const float *offset,
                            const int32_t *local_to_global, int32_t x_on_device);

/* Allocate solver state after the last mlx_add_partition_*; z = 0, u = 0 (iteration 1 of
 * jobs/RegressionAdmmTrain.java:155,184,310-312). */
int mlx_finalize(mlx_handle h);

/* Resume / warm start: z[n_lambda][n_global] (double, the driver's z) and
 * u[local partitions in add order][n_lambda][n_global] float32 (the u file); NULL leaves as is. */
int mlx_set_state(mlx_handle h, const double *z, const float *u);

/* ---- one ADMM iteration ---------------------------------------------------------------------
 * mlx_admm_iterate == mlx_admm_solve_local + (RCCL all-reduce of the means if mlx_comm_init was
 * called) + mlx_admm_consensus_finish; it is one trip of the loop body
 * jobs/RegressionAdmmTrain.java:357-472.
 *   liblinear_epsilon = Double.parseDouble(String.valueOf(float eps)) as the reducer sees it (:346,:702)
 *   rho_adapt_rate    = conf RHO_ADAPT_RATE (:316,:326,:621,:653-658); 1.0f = none */
int mlx_admm_iterate(mlx_handle h, double liblinear_epsilon, float rho_adapt_rate, mlx_stats *stats);

/* Split form for callers that run the exchange step themselves (one process per GPU under
 * torch.distributed): solve all local (partition, lambda) problems and leave this shard's partial
 * means  xbar = sum_k (1/num_blocks) f32(beta_k),  ubar = sum_k (1/num_blocks) u_k  in one device
 * buffer of 2*n_lambda*n_global doubles ([xbar | ubar]); the caller sums that buffer over all
 * shards (ncclAllReduce / all_reduce(SUM)) and then calls mlx_admm_consensus_finish.
 * Stream ordering: mlx_admm_solve_local has completed on return (the buffer is final). The caller's collective
 * must in turn be COMPLETE before mlx_admm_consensus_finish is called, unless it was enqueued on the very stream
 * given to mlx_set_stream: the handle's kernels are ordered only against that stream. */
int mlx_admm_solve_local(mlx_handle h, double liblinear_epsilon, float rho_adapt_rate, mlx_stats *stats);
int mlx_consensus_buffer(mlx_handle h, void **device_ptr, size_t *count_doubles);
int mlx_admm_consensus_finish(mlx_handle h, mlx_stats *stats);

/* ---- mean-model warm start (initialize.boost.rate > 0 and regularizer == 2) -----------------
 * Replaces the RegressionNaiveTrain job that jobs/RegressionAdmmTrain.java:236-276 launches before iteration 1
 * and the meanModel call that turns its part files into the initial z:
 *   per (lambda, partition): LibLinear.train(dataset, null, null, priorVarMap, prior.mean, 1/lambda,
 *   "epsilon=<liblinear.epsilon>") from w = 0 (jobs/RegressionNaiveTrain.java:318-404; priorVarMap = 1/lambda.map[k]
 *   where mapped, 100000 for the intercept unless penalize.intercept, :311-320), model stored as float32, then
 *   z = sum_k (1/num.blocks) model_k (utils/LinearModelUtils.java:68-86).
 * mlx_naive_init == mlx_naive_solve_local + (RCCL all-reduce if mlx_comm_init was called) + mlx_naive_finish;
 * the split form leaves this shard's partial mean in the xbar half of mlx_consensus_buffer (ubar half = 0) for
 * the caller's own all-reduce. Afterwards z = the mean (double), u = 0; run iteration 1 with
 * rho_adapt_rate = initialize.boost.rate (:313-317).
 *   liblinear_epsilon = Double.parseDouble(String.valueOf(conf float liblinear.epsilon)), 0.01 when the job
 *   file does not set it (:246-249);  prior_mean = (double)(float) prior.mean (default 0). */
int mlx_naive_init(mlx_handle h, double liblinear_epsilon, double prior_mean, mlx_stats *stats);
int mlx_naive_solve_local(mlx_handle h, double liblinear_epsilon, double prior_mean, mlx_stats *stats);
int mlx_naive_finish(mlx_handle h);

/* ---- results ------------------------------------------------------------------------------- */
/* Driver z in double and as the float32 the final-model file holds (models/LinearModel.java:703,716). */
int mlx_get_z(mlx_handle h, double *z_double /* may be NULL */, float *z_float /* may be NULL */);
/* iter-i/model parity dumps: the reducer outputs of the last iteration for one local partition
 * (index in add order) and lambda: model, uplusx (jobs/RegressionAdmmTrain.java:706-711) and the u
 * that computeU derives for the next iteration (:752-757). Any pointer may be NULL. */
int mlx_get_partition_model(mlx_handle h, int32_t local_index, int32_t lambda_index, float *beta,
                            float *uplusx, float *u_next);
/* Per-problem counters of the last solve: out[q*4 + {0,1,2,3}] = newton_iters, accepted, cg_iters,
 * x_passes_ref for q = local_index*n_lambda + lambda_index. */
int mlx_get_solve_counters(mlx_handle h, int32_t *out);
/* Dimensions a host needs to size (or check) its arrays: out[0..5] = n_global, n_lambda, partitions held by this handle,
 * num.blocks, and -- for the local partition `local_index` (add order; -1: zeros) -- its n_local and its row count l.
 * No reference counterpart: the Java code sizes everything by HashMap; the JNI glue (jni/mlease_jni.c) uses it to turn a
 * wrongly sized Java array into an IllegalArgumentException instead of a native out-of-bounds access. */
int mlx_get_dims(mlx_handle h, int32_t local_index, int32_t out[6]);

/* ---- test log-likelihood per iteration (jobs/RegressionAdmmTrain.java:766-811, updateLogLikBestModel :812-845) ----
 * Upload the test rows once (the reference re-reads the first file under test.path, <= 1 000 000 rows, every
 * iteration): CSR with GLOBAL feature ids (-1 = name absent from the model: skipped as LinearModel.eval does,
 * models/LinearModel.java:251), val[nnz] the feature values as the DOUBLES Util.getDoubleAvro yields (evalInstanceAvro does
 * not cast them to float, models/LinearModel.java:530-534; only the training rows are, jobs/RegressionPrepare.java:145), NULL for
 * binary.feature, response[l] as read (1 / 0 / -1), weight[l] and offset[l] likewise doubles (NULL = 1 / 0). */
int mlx_set_test_data(mlx_handle h, int32_t l, int64_t nnz, const int64_t *row_ptr, const int32_t *global_idx,
                      const double *val, const int8_t *response, const double *weight, const double *offset);
/* loglik_sum[n_lambda] = sum_i evalInstanceAvro(record_i, loglik=true, 1, ignore_value) with the CURRENT driver z
 * (double); the caller divides by its sum of weights n (:792-807). */
int mlx_test_loglik(mlx_handle h, double *loglik_sum);

/* ---- unit-test seam S2 == LibLinear.train (liblinearfunc/LibLinear.java:200-228) ---------------
 * Solve ONE problem on local partition `local_index` with explicit dense arrays in the partition's
 * LOCAL index space: w[n_local] holds initParam on entry and the TRON result on exit;
 * prior_var[n_local] (per-coordinate), epsilon as parsed from the option string. */
int mlx_solve_one(mlx_handle h, int32_t local_index, double *w, const double *prior_mean,
                  const double *prior_var, double epsilon, int32_t max_iter, int32_t *counters4,
                  double *f_out, double *gnorm_out, double *gnorm1_out);

/* ---- RegressionTest scoring (jobs/RegressionTest.java:147-175, the AdmmTestMapper) -----------------------------
 * pred[i] = (float) model.evalInstanceAvro(record_i, loglik = false, ignore_value) = (float)(offset_i + eval(features_i))
 * (models/LinearModel.java:241-257,491-541) for l rows in CSR form with GLOBAL feature ids (-1 = a name the model does
 * not hold: skipped), val as doubles (see mlx_set_test_data) or NULL for binary.feature, offset NULL = 0. `model` is one record of the final-model /
 * best-model file as dense float32 [n_global] (intercept last), exactly the values LinearModel reads back from avro.
 * Needs only mlx_create (no training partitions); sums run in record order, one thread per row. */
int mlx_score_rows(mlx_handle h, int32_t n_global, const float *model, int32_t l, int64_t nnz, const int64_t *row_ptr,
                   const int32_t *global_idx, const double *val, const double *offset, float *pred);

/* ---- posterior variance at the mode: the computePosteriorVar tail of LibLinear.train ----------
 * (liblinearfunc/LibLinear.java:221-228 with computePosteriorVar = true, body :314-337; the ADMM reducer passes false,
 * jobs/ItemModelTrain.java:269 is the reference's caller). On local partition `local_index`, arrays in the
 * partition's LOCAL index space like mlx_solve_one:
 *   full == 0: post_var[j] = 1 / hessianDiagonal(w)[j]          (liblinearfunc/LogisticRegressionL2.java:304-327)
 *   full != 0: H = diag(1/prior_var) + X' D X                   (:258-297; D_ii = weight_i p_i (1 - p_i)), built on the
 *              GPU as an fp64-MFMA Gram matrix, then commons-math3 3.2 CholeskyDecomposition + getInverse():
 *              post_var_matrix[n_local * n_local] (row-major, may be NULL) and post_var = its diagonal.
 * Errors: MLX_ERR_MODEL_FITTING when the Cholesky checks fail (the reference throws NonPositiveDefiniteMatrixException);
 * MLX_ERR_INVALID for CSR partitions too large to densify (n_local > 8192). gram_ms (may be NULL) receives the HIP-event
 * time of the Gram kernel when full != 0. */
int mlx_posterior_variance(mlx_handle h, int32_t local_index, const double *w, const double *prior_var, int32_t full,
                           double *post_var, double *post_var_matrix, double *gram_ms);

/* ---- multi-GPU exchange inside the library (RCCL over xGMI) ---------------------------------
 * Replaces the gather of iter-i/model + iter-i/u files to the driver (SURVEY 2a collective table). */
#define MLX_UNIQUE_ID_BYTES 128
int mlx_comm_get_unique_id(char out[MLX_UNIQUE_ID_BYTES]);
int mlx_comm_init(mlx_handle h, const char unique_id[MLX_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank);

/* Library build info, e.g. "mlease_hip gfx950 <date>". */
const char *mlx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MLEASE_ADMM_H */
