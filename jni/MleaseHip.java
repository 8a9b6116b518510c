// src/main/java/com/linkedin/mlease/regression/gpu/MleaseHip.java
//
// JNI view of include/mlease_admm.h (libmlease_hip.so): the class a maintainer adds to linkedin/ml-ease to call the
// MI355X path from RegressionAdmmTrain.run (INTEGRATION.md section 3 shows the patch). One native per C entry point,
// same order as the header. One instance = one mlx_handle = one GPU = one host thread.
//
// Error mapping (the C glue, jni/mlease_jni.c):
//   MLX_ERR_MODEL_FITTING, MLX_ERR_HIP, MLX_ERR_NO_DEVICE, MLX_ERR_COMM
//       -> IOException("Model fitting error!", new RuntimeException(<library message>))
//          -- what AdmmReducer.reduce throws for any failure of liblinear.train (jobs/RegressionAdmmTrain.java:713-716)
//   MLX_ERR_MISSING_MODELS -> RuntimeException("Some models failed!")           (utils/LinearModelUtils.java:80-83)
//   MLX_ERR_INVALID        -> IllegalArgumentException(<library message>)
//
// NOT compiled in this repository's image (no JDK); tests/test_jni_glue.py keeps the C side honest against a stub jni.h
// and checks that every native below has its Java_... function and every header entry point is bound.
package com.linkedin.mlease.regression.gpu;

import java.io.IOException;

public final class MleaseHip implements AutoCloseable
{
  static
  {
    System.loadLibrary("mlease_jni");     // libmlease_jni.so, linked against libmlease_hip.so
  }

  /** mlx_stats of the last solve (include/mlease_admm.h); filled by the iterate/solve natives. */
  public static final class Stats
  {
    public double maxdiff, mindiff;
    public long solves, newtonIters, accepted, cgIters, xPassesRef, xPassesDev, ticks;
    public double algBytesDev, xpassMs, totalMs;
    public long xpassLaunches;
    public double rowpassMs, colpassMs, stepMs;
    public double xpassBusyMs, rowpassBusyMs, colpassBusyMs, stepBusyMs;
  }

  private long handle;                    // mlx_handle

  public MleaseHip(int deviceId) throws IOException
  {
    handle = create(deviceId);
  }

  @Override
  public void close()
  {
    if (handle != 0)
    {
      destroy(handle);
      handle = 0;
    }
  }

  // ---- lifetime -------------------------------------------------------------------------------------------------
  private static native long create(int deviceId) throws IOException;                       // mlx_create
  private static native void destroy(long handle);                                          // mlx_destroy
  public native void setStream(long hipStreamOrZero) throws IOException;                    // mlx_set_stream
  public native void setProfiling(boolean enable) throws IOException;                       // mlx_set_profiling
  public static native String version();                                                    // mlx_version
  /** Numerics contract: 0 = fast (the library's default; the drop-in job -- INTEGRATION.md section 3, the native CLI -- asks for 1 unless
   *  mlease.numerics=fast), 1 = reference order (every sum a sequential loop as in bw/Tron.java and
   *  liblinearfunc/LogisticRegressionL2.java; bit-identical to the Java algorithm up to Math.exp / Math.log1p), 2 = the same on the
   *  one-launch verification kernel. Before the first partition is added. Job key: mlease.numerics. */
  public static final int NUMERICS_FAST = 0, NUMERICS_REFERENCE_ORDER = 1, NUMERICS_REFERENCE_ORDER_ONE_LAUNCH = 2;
  public native void setNumerics(int mode) throws IOException;                              // mlx_set_numerics
  public native void setOption(String key, String value) throws IOException;                // mlx_set_option
  public native String getOption(String key) throws IOException;                            // mlx_get_option

  // ---- problem definition ---------------------------------------------------------------------------------------
  /** lambda ascending; lambdaMap = null or per-global-feature lambda with NaN = "use the global lambda". */
  public native void setProblem(int nGlobal, float[] lambdaAscending, float[] rho, int numBlocks,
                                boolean penalizeIntercept, float[] lambdaMapOrNull) throws IOException;   // mlx_set_problem
  public native void setRegularizer(int regularizer) throws IOException;                    // mlx_set_regularizer

  /** Rows of ONE partition, local ids as LibLinearDataset assigns them, intercept NOT included (mlx_add_partition_csr). */
  public native void addPartitionCsr(int partitionId, int nLocal, long[] rowPtr, int[] colIdx, float[] valOrNull,
                                     byte[] y, float[] weightOrNull, float[] offsetOrNull, int[] localToGlobal)
      throws IOException;

  /** Several partitions per call; the host-side preparation runs on a thread pool (mlx_add_partitions_csr). */
  public native void addPartitionsCsr(int[] partitionId, int[] nLocal, long[][] rowPtr, int[][] colIdx,
                                      float[][] valOrNull, byte[][] y, float[][] weightOrNull,
                                      float[][] offsetOrNull, int[][] localToGlobal) throws IOException;

  /** Dense tile [l][ld] float32, the first nFeat columns used (mlx_add_partition_dense, host buffers). */
  public native void addPartitionDense(int partitionId, int l, int nFeat, long ld, float[] x, byte[] y,
                                       float[] weightOrNull, float[] offsetOrNull, int[] localToGlobal)
      throws IOException;

  public native void finalizeProblem() throws IOException;                                  // mlx_finalize
  public native void setState(double[] zOrNull, float[] uOrNull) throws IOException;        // mlx_set_state

  // ---- one ADMM iteration -------------------------------------------------------------------------------------------
  /** One trip of RegressionAdmmTrain.java:357-472 (mlx_admm_iterate). */
  public native Stats admmIterate(double liblinearEpsilon, float rhoAdaptRate) throws IOException;
  public native Stats admmSolveLocal(double liblinearEpsilon, float rhoAdaptRate) throws IOException;    // mlx_admm_solve_local
  /** Device pointer and length (doubles) of the [xbar | ubar] buffer for a caller-run all-reduce (mlx_consensus_buffer). */
  public native long[] consensusBuffer() throws IOException;
  public native Stats admmConsensusFinish() throws IOException;                             // mlx_admm_consensus_finish

  // ---- mean-model warm start (initialize.boost.rate; RegressionAdmmTrain.java:236-276) -------------------------------
  public native Stats naiveInit(double liblinearEpsilon, double priorMean) throws IOException;           // mlx_naive_init
  public native Stats naiveSolveLocal(double liblinearEpsilon, double priorMean) throws IOException;     // mlx_naive_solve_local
  public native void naiveFinish() throws IOException;                                      // mlx_naive_finish

  // ---- results --------------------------------------------------------------------------------------------------------
  public native void getZ(double[] zDoubleOrNull, float[] zFloatOrNull) throws IOException; // mlx_get_z
  public native void getPartitionModel(int localIndex, int lambdaIndex, float[] betaOrNull, float[] uplusxOrNull,
                                       float[] uNextOrNull) throws IOException;             // mlx_get_partition_model
  /** out[q*4 + {0,1,2,3}] = newton, accepted, cg, xPassesRef for q = localIndex*nLambda + lambdaIndex. */
  public native void getSolveCounters(int[] out) throws IOException;                        // mlx_get_solve_counters
  /** {nGlobal, nLambda, local partitions, num.blocks, nLocal and rows of partition localIndex (0, 0 for -1)}. */
  public native int[] dims(int localIndexOrMinus1) throws IOException;                      // mlx_get_dims

  // ---- test log-likelihood per iteration (RegressionAdmmTrain.java:766-845) ------------------------------------------
  public native void setTestData(long[] rowPtr, int[] globalIdx, double[] valOrNull, byte[] response,
                                 double[] weightOrNull, double[] offsetOrNull) throws IOException;       // mlx_set_test_data
  public native void testLoglik(double[] loglikSumPerLambda) throws IOException;            // mlx_test_loglik

  // ---- LibLinear.train seam, scoring, posterior variance -------------------------------------------------------------
  /** w holds initParam on entry, the TRON result on exit; returns {f, gnorm, gnorm1}; counters4 may be null. */
  public native double[] solveOne(int localIndex, double[] w, double[] priorMeanOrNull, double[] priorVar,
                                  double epsilon, int maxIter, int[] counters4OrNull) throws IOException; // mlx_solve_one
  /** AdmmTestMapper.map (RegressionTest.java:147-175) for l rows; model = one final-model record, intercept last. */
  public native void scoreRows(float[] modelInterceptLast, long[] rowPtr, int[] globalIdx, double[] valOrNull,
                               double[] offsetOrNull, float[] pred) throws IOException;     // mlx_score_rows
  /** LibLinear.train(..., computePosteriorVar = true) tail (LibLinear.java:314-337); returns the Gram-kernel ms (0 if !full). */
  public native double posteriorVariance(int localIndex, double[] w, double[] priorVar, boolean full,
                                         double[] postVar, double[] postVarMatrixOrNull) throws IOException;   // mlx_posterior_variance

  // ---- multi-GPU exchange inside the library (RCCL) -------------------------------------------------------------------
  public static native byte[] commGetUniqueId() throws IOException;                         // mlx_comm_get_unique_id
  public native void commInit(byte[] uniqueId128, int nranks, int rank) throws IOException; // mlx_comm_init
}
