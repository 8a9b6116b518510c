/*
 * mlease_jni.c -- JNI glue between com.linkedin.mlease.regression.gpu.MleaseHip (jni/MleaseHip.java) and the C-ABI of
 * libmlease_hip.so (include/mlease_admm.h). One Java_... function per native, 1:1 with the header's entry points.
 *
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/mlease_jni.c -o libmlease_jni.so \
 *       -Lml-ease_amd/csrc -lmlease_hip -Wl,-rpath,'$ORIGIN'
 *
 * This repository's image has no JDK: the file is syntax- and type-checked against tests/jni_stub/jni.h (the JNI types and
 * the function table entries used here) by tests/test_jni_glue.py, not linked.
 *
 * Array arguments are pinned with Get<Type>ArrayElements for the duration of ONE library call and released right after
 * it (JNI_ABORT for inputs): every mlx_* call has copied what it needs to the device when it returns, which is the
 * ownership rule of the header. Error mapping: see throw_for().
 */
#include <jni.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mlease_admm.h"

#define JFN(ret, name) JNIEXPORT ret JNICALL Java_com_linkedin_mlease_regression_gpu_MleaseHip_##name

/* ---- helpers ------------------------------------------------------------------------------------------------------ */
static mlx_handle handle_of(JNIEnv *env, jobject self)
{
    jclass cls = (*env)->GetObjectClass(env, self);
    jfieldID fid = (*env)->GetFieldID(env, cls, "handle", "J");
    return (mlx_handle)(intptr_t)(*env)->GetLongField(env, self, fid);
}

/* MLX_ERR_MODEL_FITTING / HIP / NO_DEVICE / COMM -> IOException("Model fitting error!", cause), as AdmmReducer.reduce
 * wraps any failure of liblinear.train (jobs/RegressionAdmmTrain.java:713-716);
 * MLX_ERR_MISSING_MODELS -> RuntimeException("Some models failed!") (utils/LinearModelUtils.java:80-83);
 * MLX_ERR_INVALID -> IllegalArgumentException(message). Returns rc so callers can `if (throw_for(...)) return`. */
static int throw_for(JNIEnv *env, mlx_handle h, int rc)
{
    if (rc == MLX_OK) return 0;
    const char *msg = mlx_last_error(h);
    if (!msg) msg = "";
    /* a failing FindClass / GetMethodID / NewObject leaves ITS exception pending (NoClassDefFoundError, OutOfMemoryError):
     * that one then reaches Java instead -- nothing below may be called with a NULL class or object */
    if (rc == MLX_ERR_INVALID) {
        jclass iae = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (iae) (*env)->ThrowNew(env, iae, msg);
    } else if (rc == MLX_ERR_MISSING_MODELS) {
        jclass rte = (*env)->FindClass(env, "java/lang/RuntimeException");
        if (rte) (*env)->ThrowNew(env, rte, "Some models failed!");
    } else {
        jclass rte = (*env)->FindClass(env, "java/lang/RuntimeException");
        jclass ioe = rte ? (*env)->FindClass(env, "java/io/IOException") : NULL;
        jmethodID rctor = ioe ? (*env)->GetMethodID(env, rte, "<init>", "(Ljava/lang/String;)V") : NULL;
        jmethodID ictor = rctor ? (*env)->GetMethodID(env, ioe, "<init>", "(Ljava/lang/String;Ljava/lang/Throwable;)V") : NULL;
        if (ictor) {
            char buf[640];
            snprintf(buf, sizeof buf, "mlease_hip error %d: %s", rc, msg);
            jstring m1 = (*env)->NewStringUTF(env, buf);
            jobject cause = m1 ? (*env)->NewObject(env, rte, rctor, m1) : NULL;
            jstring m2 = cause ? (*env)->NewStringUTF(env, "Model fitting error!") : NULL;
            jobject ex = m2 ? (*env)->NewObject(env, ioe, ictor, m2, cause) : NULL;
            if (ex) (*env)->Throw(env, (jthrowable)ex);
        }
    }
    return rc;
}

/* ---- argument validation: every array a native reads or writes is checked BEFORE it is pinned -- non-null where the C-ABI
 * requires it, and long enough for what the library will read or write (lengths derived from rowPtr / mlx_get_dims) -- and
 * a violation becomes IllegalArgumentException, never a native out-of-bounds access. */
static int bad_arg(JNIEnv *env, const char *fmt, ...)
{
    char buf[320];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    jclass iae = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (iae) (*env)->ThrowNew(env, iae, buf);          /* (FindClass failing leaves its own exception pending) */
    return 1;
}
/* length of a possibly-null array, -1 for null */
static jlong alen(JNIEnv *env, jarray a) { return a ? (jlong)(*env)->GetArrayLength(env, a) : -1; }
/* `a` must be non-null (unless optional) and hold at least `need` elements */
static int need_len(JNIEnv *env, const char *name, jarray a, jlong need, int optional)
{
    if (!a) return optional ? 0 : bad_arg(env, "%s is null", name);
    const jlong have = alen(env, a);
    if (need < 0 || have < need) return bad_arg(env, "%s has %lld elements, %lld needed", name, (long long)have, (long long)need);
    return 0;
}
/* number of non-zeros a CSR row pointer announces: rowPtr[l], read without pinning; -1 (exception thrown) if unusable */
static jlong csr_nnz(JNIEnv *env, jlongArray rowPtr, jlong *l_out)
{
    if (!rowPtr) { bad_arg(env, "rowPtr is null"); return -1; }
    const jlong n = alen(env, rowPtr);
    if (n < 1) { bad_arg(env, "rowPtr is empty (needs l + 1 entries)"); return -1; }
    jlong last = 0;
    (*env)->GetLongArrayRegion(env, rowPtr, (jsize)(n - 1), 1, &last);
    if (last < 0 || last >= ((jlong)1 << 31)) { bad_arg(env, "rowPtr[l] = %lld is not a valid entry count", (long long)last); return -1; }
    *l_out = n - 1;
    return last;
}
static int dims_of(JNIEnv *env, mlx_handle h, jint local_index, int32_t d[6])
{
    return throw_for(env, h, mlx_get_dims(h, (int32_t)local_index, d));
}

/* pinned views of (possibly null) primitive arrays */
#define PIN(T, J, arr) ((arr) ? (*env)->Get##J##ArrayElements(env, (arr), NULL) : (T *)NULL)
#define UNPIN_IN(J, arr, p) do { if (arr) (*env)->Release##J##ArrayElements(env, (arr), (p), JNI_ABORT); } while (0)
#define UNPIN_OUT(J, arr, p) do { if (arr) (*env)->Release##J##ArrayElements(env, (arr), (p), 0); } while (0)

static jobject stats_to_java(JNIEnv *env, const mlx_stats *s)
{
    jclass cls = (*env)->FindClass(env, "com/linkedin/mlease/regression/gpu/MleaseHip$Stats");
    if (!cls) return NULL;
    jmethodID ctor = (*env)->GetMethodID(env, cls, "<init>", "()V");
    jobject o = ctor ? (*env)->NewObject(env, cls, ctor) : NULL;
    if (!o) return NULL;
#define SETD(name, v) (*env)->SetDoubleField(env, o, (*env)->GetFieldID(env, cls, name, "D"), (jdouble)(v))
#define SETJ(name, v) (*env)->SetLongField(env, o, (*env)->GetFieldID(env, cls, name, "J"), (jlong)(v))
    SETD("maxdiff", s->maxdiff); SETD("mindiff", s->mindiff);
    SETJ("solves", s->solves); SETJ("newtonIters", s->newton_iters); SETJ("accepted", s->accepted);
    SETJ("cgIters", s->cg_iters); SETJ("xPassesRef", s->x_passes_ref); SETJ("xPassesDev", s->x_passes_dev);
    SETJ("ticks", s->ticks);
    SETD("algBytesDev", s->alg_bytes_dev); SETD("xpassMs", s->xpass_ms); SETD("totalMs", s->total_ms);
    SETJ("xpassLaunches", s->xpass_launches);
    SETD("rowpassMs", s->rowpass_ms); SETD("colpassMs", s->colpass_ms); SETD("stepMs", s->step_ms);
    SETD("xpassBusyMs", s->xpass_busy_ms); SETD("rowpassBusyMs", s->rowpass_busy_ms); SETD("colpassBusyMs", s->colpass_busy_ms);
    SETD("stepBusyMs", s->step_busy_ms);
#undef SETD
#undef SETJ
    return o;
}

/* ---- lifetime ----------------------------------------------------------------------------------------------------- */
JFN(jlong, create)(JNIEnv *env, jclass cls, jint deviceId)
{
    (void)cls;
    mlx_handle h = NULL;
    int rc = mlx_create((int)deviceId, &h);
    if (throw_for(env, NULL, rc)) return 0;
    return (jlong)(intptr_t)h;
}

JFN(void, destroy)(JNIEnv *env, jclass cls, jlong handle)
{
    (void)env; (void)cls;
    mlx_destroy((mlx_handle)(intptr_t)handle);
}

JFN(void, setStream)(JNIEnv *env, jobject self, jlong hipStream)
{
    mlx_handle h = handle_of(env, self);
    throw_for(env, h, mlx_set_stream(h, (void *)(intptr_t)hipStream));
}

JFN(void, setProfiling)(JNIEnv *env, jobject self, jboolean enable)
{
    mlx_handle h = handle_of(env, self);
    throw_for(env, h, mlx_set_profiling(h, enable ? 1 : 0));
}

/* numerics contract / per-handle options (include/mlease_admm.h: mlx_set_numerics, mlx_set_option, mlx_get_option); the job key
 * mlease.numerics of the patched driver goes through setOption("numerics", ...) */
JFN(void, setNumerics)(JNIEnv *env, jobject self, jint mode)
{
    mlx_handle h = handle_of(env, self);
    throw_for(env, h, mlx_set_numerics(h, (int32_t)mode));
}

JFN(void, setOption)(JNIEnv *env, jobject self, jstring key, jstring value)
{
    mlx_handle h = handle_of(env, self);
    if (!key || !value) { bad_arg(env, "setOption: key or value is null"); return; }
    const char *k = (*env)->GetStringUTFChars(env, key, NULL), *v = (*env)->GetStringUTFChars(env, value, NULL);
    const int rc = (k && v) ? mlx_set_option(h, k, v) : MLX_ERR_INVALID;
    if (v) (*env)->ReleaseStringUTFChars(env, value, v);
    if (k) (*env)->ReleaseStringUTFChars(env, key, k);
    throw_for(env, h, rc);
}

JFN(jstring, getOption)(JNIEnv *env, jobject self, jstring key)
{
    mlx_handle h = handle_of(env, self);
    if (!key) { bad_arg(env, "getOption: key is null"); return NULL; }
    enum { CAP = 1 << 17 };                    /* ("tick_log" of a long solve is a few KB) */
    char *buf = (char *)malloc(CAP);
    if (!buf) { bad_arg(env, "getOption: out of memory"); return NULL; }
    const char *k = (*env)->GetStringUTFChars(env, key, NULL);
    const int rc = k ? mlx_get_option(h, k, buf, CAP) : MLX_ERR_INVALID;
    if (k) (*env)->ReleaseStringUTFChars(env, key, k);
    if (rc) { free(buf); throw_for(env, h, rc); return NULL; }
    jstring out = (*env)->NewStringUTF(env, buf);
    free(buf);
    return out;
}

JFN(jstring, version)(JNIEnv *env, jclass cls)
{
    (void)cls;
    return (*env)->NewStringUTF(env, mlx_version());
}

/* ---- problem definition --------------------------------------------------------------------------------------------- */
JFN(void, setProblem)(JNIEnv *env, jobject self, jint nGlobal, jfloatArray lambda, jfloatArray rho, jint numBlocks,
                      jboolean penalizeIntercept, jfloatArray lambdaMap)
{
    mlx_handle h = handle_of(env, self);
    if (need_len(env, "lambda", lambda, 1, 0)) return;
    const jsize nl = (*env)->GetArrayLength(env, lambda);
    if (need_len(env, "rho", rho, nl, 0) || need_len(env, "lambdaMap", lambdaMap, nGlobal, 1)) return;
    jfloat *la = PIN(jfloat, Float, lambda), *rh = PIN(jfloat, Float, rho), *lm = PIN(jfloat, Float, lambdaMap);
    int rc = mlx_set_problem(h, (int32_t)nGlobal, (int32_t)nl, la, rh, (int32_t)numBlocks, penalizeIntercept ? 1 : 0, lm);
    UNPIN_IN(Float, lambda, la); UNPIN_IN(Float, rho, rh); UNPIN_IN(Float, lambdaMap, lm);
    throw_for(env, h, rc);
}

JFN(void, setRegularizer)(JNIEnv *env, jobject self, jint regularizer)
{
    mlx_handle h = handle_of(env, self);
    throw_for(env, h, mlx_set_regularizer(h, (int32_t)regularizer));
}

JFN(void, addPartitionCsr)(JNIEnv *env, jobject self, jint pid, jint nLocal, jlongArray rowPtr, jintArray colIdx,
                           jfloatArray val, jbyteArray y, jfloatArray wt, jfloatArray off, jintArray l2g)
{
    mlx_handle h = handle_of(env, self);
    jlong l = 0;
    const jlong nnz = csr_nnz(env, rowPtr, &l);
    if (nnz < 0) return;
    if (need_len(env, "colIdx", colIdx, nnz, nnz == 0) || need_len(env, "val", val, nnz, 1) || need_len(env, "y", y, l, 0) ||
        need_len(env, "weight", wt, l, 1) || need_len(env, "offset", off, l, 1) || need_len(env, "localToGlobal", l2g, nLocal, 0))
        return;
    jlong *rp = PIN(jlong, Long, rowPtr);
    jint *ci = PIN(jint, Int, colIdx), *map = PIN(jint, Int, l2g);
    jfloat *v = PIN(jfloat, Float, val), *w = PIN(jfloat, Float, wt), *o = PIN(jfloat, Float, off);
    jbyte *yy = PIN(jbyte, Byte, y);
    int rc = mlx_add_partition_csr(h, (int32_t)pid, (int32_t)l, (int32_t)nLocal, nnz, (const int64_t *)rp, (const int32_t *)ci,
                                   v, (const int8_t *)yy, w, o, (const int32_t *)map);
    UNPIN_IN(Long, rowPtr, rp); UNPIN_IN(Int, colIdx, ci); UNPIN_IN(Int, l2g, map);
    UNPIN_IN(Float, val, v); UNPIN_IN(Float, wt, w); UNPIN_IN(Float, off, o); UNPIN_IN(Byte, y, yy);
    throw_for(env, h, rc);
}

JFN(void, addPartitionsCsr)(JNIEnv *env, jobject self, jintArray partitionId, jintArray nLocal, jobjectArray rowPtr,
                            jobjectArray colIdx, jobjectArray val, jobjectArray y, jobjectArray wt, jobjectArray off,
                            jobjectArray l2g)
{
    mlx_handle h = handle_of(env, self);
    if (need_len(env, "partitionId", partitionId, 0, 0)) return;
    const jsize n = (*env)->GetArrayLength(env, partitionId);
    if (need_len(env, "nLocal", nLocal, n, 0) || need_len(env, "rowPtr", rowPtr, n, 0) || need_len(env, "colIdx", colIdx, n, 0) ||
        need_len(env, "val", val, n, 1) || need_len(env, "y", y, n, 0) || need_len(env, "weight", wt, n, 1) ||
        need_len(env, "offset", off, n, 1) || need_len(env, "localToGlobal", l2g, n, 0))
        return;
    /* 7 local references per partition stay alive across the library call (their arrays are pinned through it) */
    if ((*env)->EnsureLocalCapacity(env, 7 * n + 16) != 0) return;           /* OutOfMemoryError pending */
    /* pass 1: every partition's arrays checked before anything is pinned */
    {
        jint nl1 = 0;
        for (jsize k = 0; k < n; k++) {
            jlongArray r = (jlongArray)(*env)->GetObjectArrayElement(env, rowPtr, k);
            jlong lk = 0;
            const jlong nz = csr_nnz(env, r, &lk);
            int bad = nz < 0;
            (*env)->GetIntArrayRegion(env, nLocal, k, 1, &nl1);
            jarray e;
#define CHK(arr, name, need, opt) if (!bad) { e = (arr) ? (jarray)(*env)->GetObjectArrayElement(env, (arr), k) : NULL; bad = need_len(env, name, e, (need), (opt)); if (e) (*env)->DeleteLocalRef(env, e); }
            CHK(colIdx, "colIdx[k]", nz, nz == 0); CHK(val, "val[k]", nz, 1); CHK(y, "y[k]", lk, 0); CHK(wt, "weight[k]", lk, 1);
            CHK(off, "offset[k]", lk, 1); CHK(l2g, "localToGlobal[k]", nl1, 0);
#undef CHK
            if (r) (*env)->DeleteLocalRef(env, r);
            if (bad) return;
        }
    }
    jint *pid = PIN(jint, Int, partitionId), *nloc = PIN(jint, Int, nLocal);
    /* per-partition pinned views; val / wt / off may be null as a whole or per entry */
    int32_t *ls = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    int64_t *nnz = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    const int64_t **rp = (const int64_t **)calloc((size_t)n + 1, sizeof(void *));
    const int32_t **ci = (const int32_t **)calloc((size_t)n + 1, sizeof(void *)), **mp = (const int32_t **)calloc((size_t)n + 1, sizeof(void *));
    const float **vv = (const float **)calloc((size_t)n + 1, sizeof(void *)), **ww = (const float **)calloc((size_t)n + 1, sizeof(void *)),
                **oo = (const float **)calloc((size_t)n + 1, sizeof(void *));
    const int8_t **yy = (const int8_t **)calloc((size_t)n + 1, sizeof(void *));
    jobject *arrs = (jobject *)calloc(7 * ((size_t)n + 1), sizeof(jobject));
    for (jsize k = 0; k < n; k++) {
        jobject *a = arrs + 7 * k;
        a[0] = (*env)->GetObjectArrayElement(env, rowPtr, k);
        a[1] = (*env)->GetObjectArrayElement(env, colIdx, k);
        a[2] = val ? (*env)->GetObjectArrayElement(env, val, k) : NULL;
        a[3] = (*env)->GetObjectArrayElement(env, y, k);
        a[4] = wt ? (*env)->GetObjectArrayElement(env, wt, k) : NULL;
        a[5] = off ? (*env)->GetObjectArrayElement(env, off, k) : NULL;
        a[6] = (*env)->GetObjectArrayElement(env, l2g, k);
        ls[k] = (int32_t)((*env)->GetArrayLength(env, (jarray)a[0]) - 1);
        rp[k] = (const int64_t *)PIN(jlong, Long, (jlongArray)a[0]);
        ci[k] = (const int32_t *)PIN(jint, Int, (jintArray)a[1]);
        vv[k] = PIN(jfloat, Float, (jfloatArray)a[2]);
        yy[k] = (const int8_t *)PIN(jbyte, Byte, (jbyteArray)a[3]);
        ww[k] = PIN(jfloat, Float, (jfloatArray)a[4]);
        oo[k] = PIN(jfloat, Float, (jfloatArray)a[5]);
        mp[k] = (const int32_t *)PIN(jint, Int, (jintArray)a[6]);
        nnz[k] = rp[k] ? rp[k][ls[k]] : 0;
    }
    int rc = mlx_add_partitions_csr(h, (int32_t)n, (const int32_t *)pid, ls, (const int32_t *)nloc, nnz, rp, ci, val ? vv : NULL, yy,
                                    wt ? ww : NULL, off ? oo : NULL, mp);
    for (jsize k = 0; k < n; k++) {
        jobject *a = arrs + 7 * k;
        UNPIN_IN(Long, (jlongArray)a[0], (jlong *)rp[k]); UNPIN_IN(Int, (jintArray)a[1], (jint *)ci[k]);
        UNPIN_IN(Float, (jfloatArray)a[2], (jfloat *)vv[k]); UNPIN_IN(Byte, (jbyteArray)a[3], (jbyte *)yy[k]);
        UNPIN_IN(Float, (jfloatArray)a[4], (jfloat *)ww[k]); UNPIN_IN(Float, (jfloatArray)a[5], (jfloat *)oo[k]);
        UNPIN_IN(Int, (jintArray)a[6], (jint *)mp[k]);
        for (int t = 0; t < 7; t++) if (a[t]) (*env)->DeleteLocalRef(env, a[t]);
    }
    UNPIN_IN(Int, partitionId, pid); UNPIN_IN(Int, nLocal, nloc);
    free(ls); free(nnz); free((void *)rp); free((void *)ci); free((void *)mp); free((void *)vv); free((void *)ww); free((void *)oo);
    free((void *)yy); free(arrs);
    throw_for(env, h, rc);
}

JFN(void, addPartitionDense)(JNIEnv *env, jobject self, jint pid, jint l, jint nFeat, jlong ld, jfloatArray x, jbyteArray y,
                             jfloatArray wt, jfloatArray off, jintArray l2g)
{
    mlx_handle h = handle_of(env, self);
    if (l < 0 || nFeat < 0 || ld < nFeat) { bad_arg(env, "addPartitionDense: l=%d nFeat=%d ld=%lld", (int)l, (int)nFeat, (long long)ld); return; }
    if (need_len(env, "x", x, l > 0 ? (jlong)(l - 1) * ld + nFeat : 0, 0) || need_len(env, "y", y, l, 0) || need_len(env, "weight", wt, l, 1) ||
        need_len(env, "offset", off, l, 1) || need_len(env, "localToGlobal", l2g, (jlong)nFeat + 1, 0))
        return;
    jfloat *xx = PIN(jfloat, Float, x), *w = PIN(jfloat, Float, wt), *o = PIN(jfloat, Float, off);
    jbyte *yy = PIN(jbyte, Byte, y);
    jint *map = PIN(jint, Int, l2g);
    int rc = mlx_add_partition_dense(h, (int32_t)pid, (int32_t)l, (int32_t)nFeat, (int64_t)ld, xx, (const int8_t *)yy, w, o,
                                     (const int32_t *)map, 0);
    UNPIN_IN(Float, x, xx); UNPIN_IN(Float, wt, w); UNPIN_IN(Float, off, o); UNPIN_IN(Byte, y, yy); UNPIN_IN(Int, l2g, map);
    throw_for(env, h, rc);
}

JFN(void, finalizeProblem)(JNIEnv *env, jobject self)
{
    mlx_handle h = handle_of(env, self);
    throw_for(env, h, mlx_finalize(h));
}

JFN(void, setState)(JNIEnv *env, jobject self, jdoubleArray z, jfloatArray u)
{
    mlx_handle h = handle_of(env, self);
    int32_t d[6];
    if (dims_of(env, h, -1, d)) return;
    const jlong zl = (jlong)d[0] * d[1];
    if (need_len(env, "z", z, zl, 1) || need_len(env, "u", u, zl * d[2], 1)) return;
    jdouble *zz = PIN(jdouble, Double, z);
    jfloat *uu = PIN(jfloat, Float, u);
    int rc = mlx_set_state(h, zz, uu);
    UNPIN_IN(Double, z, zz); UNPIN_IN(Float, u, uu);
    throw_for(env, h, rc);
}

/* ---- one ADMM iteration ------------------------------------------------------------------------------------------- */
JFN(jobject, admmIterate)(JNIEnv *env, jobject self, jdouble eps, jfloat rate)
{
    mlx_handle h = handle_of(env, self);
    mlx_stats st;
    if (throw_for(env, h, mlx_admm_iterate(h, (double)eps, (float)rate, &st))) return NULL;
    return stats_to_java(env, &st);
}

JFN(jobject, admmSolveLocal)(JNIEnv *env, jobject self, jdouble eps, jfloat rate)
{
    mlx_handle h = handle_of(env, self);
    mlx_stats st;
    if (throw_for(env, h, mlx_admm_solve_local(h, (double)eps, (float)rate, &st))) return NULL;
    return stats_to_java(env, &st);
}

JFN(jlongArray, consensusBuffer)(JNIEnv *env, jobject self)
{
    mlx_handle h = handle_of(env, self);
    void *ptr = NULL;
    size_t cnt = 0;
    if (throw_for(env, h, mlx_consensus_buffer(h, &ptr, &cnt))) return NULL;
    jlong out[2] = {(jlong)(intptr_t)ptr, (jlong)cnt};
    jlongArray arr = (*env)->NewLongArray(env, 2);
    if (arr) (*env)->SetLongArrayRegion(env, arr, 0, 2, out);
    return arr;
}

JFN(jobject, admmConsensusFinish)(JNIEnv *env, jobject self)
{
    mlx_handle h = handle_of(env, self);
    mlx_stats st;
    if (throw_for(env, h, mlx_admm_consensus_finish(h, &st))) return NULL;
    return stats_to_java(env, &st);
}

/* ---- mean-model warm start ------------------------------------------------------------------------------------------ */
JFN(jobject, naiveInit)(JNIEnv *env, jobject self, jdouble eps, jdouble priorMean)
{
    mlx_handle h = handle_of(env, self);
    mlx_stats st;
    if (throw_for(env, h, mlx_naive_init(h, (double)eps, (double)priorMean, &st))) return NULL;
    return stats_to_java(env, &st);
}

JFN(jobject, naiveSolveLocal)(JNIEnv *env, jobject self, jdouble eps, jdouble priorMean)
{
    mlx_handle h = handle_of(env, self);
    mlx_stats st;
    if (throw_for(env, h, mlx_naive_solve_local(h, (double)eps, (double)priorMean, &st))) return NULL;
    return stats_to_java(env, &st);
}

JFN(void, naiveFinish)(JNIEnv *env, jobject self)
{
    mlx_handle h = handle_of(env, self);
    throw_for(env, h, mlx_naive_finish(h));
}

/* ---- results -------------------------------------------------------------------------------------------------------- */
JFN(void, getZ)(JNIEnv *env, jobject self, jdoubleArray zDouble, jfloatArray zFloat)
{
    mlx_handle h = handle_of(env, self);
    int32_t d[6];
    if (dims_of(env, h, -1, d)) return;
    const jlong zl = (jlong)d[0] * d[1];
    if (need_len(env, "zDouble", zDouble, zl, 1) || need_len(env, "zFloat", zFloat, zl, 1)) return;
    jdouble *zd = PIN(jdouble, Double, zDouble);
    jfloat *zf = PIN(jfloat, Float, zFloat);
    int rc = mlx_get_z(h, zd, zf);
    UNPIN_OUT(Double, zDouble, zd); UNPIN_OUT(Float, zFloat, zf);
    throw_for(env, h, rc);
}

JFN(void, getPartitionModel)(JNIEnv *env, jobject self, jint localIndex, jint lambdaIndex, jfloatArray beta, jfloatArray uplusx,
                             jfloatArray uNext)
{
    mlx_handle h = handle_of(env, self);
    int32_t d[6];
    if (dims_of(env, h, -1, d)) return;
    if (need_len(env, "beta", beta, d[0], 1) || need_len(env, "uplusx", uplusx, d[0], 1) || need_len(env, "uNext", uNext, d[0], 1)) return;
    jfloat *b = PIN(jfloat, Float, beta), *x = PIN(jfloat, Float, uplusx), *u = PIN(jfloat, Float, uNext);
    int rc = mlx_get_partition_model(h, (int32_t)localIndex, (int32_t)lambdaIndex, b, x, u);
    UNPIN_OUT(Float, beta, b); UNPIN_OUT(Float, uplusx, x); UNPIN_OUT(Float, uNext, u);
    throw_for(env, h, rc);
}

JFN(void, getSolveCounters)(JNIEnv *env, jobject self, jintArray out)
{
    mlx_handle h = handle_of(env, self);
    int32_t d[6];
    if (dims_of(env, h, -1, d)) return;
    if (need_len(env, "out", out, 4 * (jlong)d[1] * d[2], 0)) return;
    jint *o = PIN(jint, Int, out);
    int rc = mlx_get_solve_counters(h, (int32_t *)o);
    UNPIN_OUT(Int, out, o);
    throw_for(env, h, rc);
}

JFN(jintArray, dims)(JNIEnv *env, jobject self, jint localIndexOrMinus1)
{
    mlx_handle h = handle_of(env, self);
    int32_t d[6];
    if (dims_of(env, h, localIndexOrMinus1, d)) return NULL;
    jintArray arr = (*env)->NewIntArray(env, 6);
    if (arr) (*env)->SetIntArrayRegion(env, arr, 0, 6, (const jint *)d);
    return arr;
}

/* ---- test log-likelihood ---------------------------------------------------------------------------------------------- */
JFN(void, setTestData)(JNIEnv *env, jobject self, jlongArray rowPtr, jintArray globalIdx, jdoubleArray val, jbyteArray response,
                       jdoubleArray weight, jdoubleArray offset)
{
    mlx_handle h = handle_of(env, self);
    jlong l = 0;
    const jlong nnz = csr_nnz(env, rowPtr, &l);
    if (nnz < 0) return;
    if (need_len(env, "globalIdx", globalIdx, nnz, nnz == 0) || need_len(env, "val", val, nnz, 1) || need_len(env, "response", response, l, 0) ||
        need_len(env, "weight", weight, l, 1) || need_len(env, "offset", offset, l, 1))
        return;
    jlong *rp = PIN(jlong, Long, rowPtr);
    jint *gi = PIN(jint, Int, globalIdx);
    jdouble *v = PIN(jdouble, Double, val);
    jbyte *r = PIN(jbyte, Byte, response);
    jdouble *w = PIN(jdouble, Double, weight), *o = PIN(jdouble, Double, offset);
    int rc = mlx_set_test_data(h, (int32_t)l, nnz, (const int64_t *)rp, (const int32_t *)gi, v, (const int8_t *)r, w, o);
    UNPIN_IN(Long, rowPtr, rp); UNPIN_IN(Int, globalIdx, gi); UNPIN_IN(Double, val, v); UNPIN_IN(Byte, response, r);
    UNPIN_IN(Double, weight, w); UNPIN_IN(Double, offset, o);
    throw_for(env, h, rc);
}

JFN(void, testLoglik)(JNIEnv *env, jobject self, jdoubleArray loglikSum)
{
    mlx_handle h = handle_of(env, self);
    int32_t d[6];
    if (dims_of(env, h, -1, d)) return;
    if (need_len(env, "loglikSumPerLambda", loglikSum, d[1], 0)) return;
    jdouble *s = PIN(jdouble, Double, loglikSum);
    int rc = mlx_test_loglik(h, s);
    UNPIN_OUT(Double, loglikSum, s);
    throw_for(env, h, rc);
}

/* ---- LibLinear.train seam, scoring, posterior variance ------------------------------------------------------------------ */
JFN(jdoubleArray, solveOne)(JNIEnv *env, jobject self, jint localIndex, jdoubleArray w, jdoubleArray priorMean, jdoubleArray priorVar,
                            jdouble epsilon, jint maxIter, jintArray counters4)
{
    mlx_handle h = handle_of(env, self);
    int32_t d[6];
    if (dims_of(env, h, localIndex, d)) return NULL;
    if (need_len(env, "w", w, d[4], 0) || need_len(env, "priorMean", priorMean, d[4], 1) || need_len(env, "priorVar", priorVar, d[4], 0) ||
        need_len(env, "counters4", counters4, 4, 1))
        return NULL;
    jdouble *ww = PIN(jdouble, Double, w), *pm = PIN(jdouble, Double, priorMean), *pv = PIN(jdouble, Double, priorVar);
    jint *c4 = PIN(jint, Int, counters4);
    double f = 0, gn = 0, gn1 = 0;
    int rc = mlx_solve_one(h, (int32_t)localIndex, ww, pm, pv, (double)epsilon, (int32_t)maxIter, (int32_t *)c4, &f, &gn, &gn1);
    UNPIN_OUT(Double, w, ww); UNPIN_IN(Double, priorMean, pm); UNPIN_IN(Double, priorVar, pv); UNPIN_OUT(Int, counters4, c4);
    if (throw_for(env, h, rc)) return NULL;
    jdouble out[3] = {f, gn, gn1};
    jdoubleArray arr = (*env)->NewDoubleArray(env, 3);
    if (arr) (*env)->SetDoubleArrayRegion(env, arr, 0, 3, out);
    return arr;
}

JFN(void, scoreRows)(JNIEnv *env, jobject self, jfloatArray model, jlongArray rowPtr, jintArray globalIdx, jdoubleArray val,
                     jdoubleArray offset, jfloatArray pred)
{
    mlx_handle h = handle_of(env, self);
    if (need_len(env, "model", model, 1, 0)) return;
    const jsize ng = (*env)->GetArrayLength(env, model);
    jlong l = 0;
    const jlong nnz = csr_nnz(env, rowPtr, &l);
    if (nnz < 0) return;
    if (need_len(env, "globalIdx", globalIdx, nnz, nnz == 0) || need_len(env, "val", val, nnz, 1) || need_len(env, "offset", offset, l, 1) ||
        need_len(env, "pred", pred, l, 0))
        return;
    jfloat *m = PIN(jfloat, Float, model), *p = PIN(jfloat, Float, pred);
    jlong *rp = PIN(jlong, Long, rowPtr);
    jint *gi = PIN(jint, Int, globalIdx);
    jdouble *o = PIN(jdouble, Double, offset), *v = PIN(jdouble, Double, val);
    int rc = mlx_score_rows(h, (int32_t)ng, m, (int32_t)l, nnz, (const int64_t *)rp, (const int32_t *)gi, v, o, p);
    UNPIN_IN(Float, model, m); UNPIN_IN(Double, val, v); UNPIN_OUT(Float, pred, p); UNPIN_IN(Long, rowPtr, rp);
    UNPIN_IN(Int, globalIdx, gi); UNPIN_IN(Double, offset, o);
    throw_for(env, h, rc);
}

JFN(jdouble, posteriorVariance)(JNIEnv *env, jobject self, jint localIndex, jdoubleArray w, jdoubleArray priorVar, jboolean full,
                                jdoubleArray postVar, jdoubleArray postVarMatrix)
{
    mlx_handle h = handle_of(env, self);
    int32_t d[6];
    if (dims_of(env, h, localIndex, d)) return 0.0;
    if (need_len(env, "w", w, d[4], 0) || need_len(env, "priorVar", priorVar, d[4], 0) || need_len(env, "postVar", postVar, d[4], 0) ||
        need_len(env, "postVarMatrix", postVarMatrix, (jlong)d[4] * d[4], !full))
        return 0.0;
    jdouble *ww = PIN(jdouble, Double, w), *pv = PIN(jdouble, Double, priorVar), *out = PIN(jdouble, Double, postVar),
            *mat = PIN(jdouble, Double, postVarMatrix);
    double ms = 0;
    int rc = mlx_posterior_variance(h, (int32_t)localIndex, ww, pv, full ? 1 : 0, out, mat, &ms);
    UNPIN_IN(Double, w, ww); UNPIN_IN(Double, priorVar, pv); UNPIN_OUT(Double, postVar, out); UNPIN_OUT(Double, postVarMatrix, mat);
    throw_for(env, h, rc);
    return (jdouble)ms;
}

/* ---- RCCL ---------------------------------------------------------------------------------------------------------------- */
JFN(jbyteArray, commGetUniqueId)(JNIEnv *env, jclass cls)
{
    (void)cls;
    char id[MLX_UNIQUE_ID_BYTES];
    if (throw_for(env, NULL, mlx_comm_get_unique_id(id))) return NULL;
    jbyteArray arr = (*env)->NewByteArray(env, MLX_UNIQUE_ID_BYTES);
    if (arr) (*env)->SetByteArrayRegion(env, arr, 0, MLX_UNIQUE_ID_BYTES, (const jbyte *)id);
    return arr;
}

JFN(void, commInit)(JNIEnv *env, jobject self, jbyteArray uniqueId, jint nranks, jint rank)
{
    mlx_handle h = handle_of(env, self);
    char id[MLX_UNIQUE_ID_BYTES];
    memset(id, 0, sizeof id);
    if (need_len(env, "uniqueId", uniqueId, 1, 0)) return;
    jsize n = (*env)->GetArrayLength(env, uniqueId);
    if (n > MLX_UNIQUE_ID_BYTES) n = MLX_UNIQUE_ID_BYTES;
    (*env)->GetByteArrayRegion(env, uniqueId, 0, n, (jbyte *)id);
    throw_for(env, h, mlx_comm_init(h, id, (int32_t)nranks, (int32_t)rank));
}
