/*
 * admm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, IEEE-double, single-threaded-per-solve restatement of the one hot
 * path of linkedin/ml-ease that this repository replaces on MI355X:
 *
 *   AdmmReducer.reduce  -> LibLinear.train -> Tron.tron/trcg -> LogisticRegressionL2.{fun,grad,Hv}
 *   + the driver-side consensus (meanModel x2, z-update, computeU, maxdiff, eps schedule).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library. The product path (ml-ease_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or KATs for this
 * path (no src/test) and cannot be executed here (no JVM).
 * This file is therefore pinned only by (a) line-by-line review against the
 * reference files cited on every function, (b) agreement to <=1e-12 with an
 * independent NumPy restatement (oracle/admm_numpy.py), (c) mathematical pins
 * in tests/ (finite differences, KKT at exit, approach to the centralised optimum), and (d) -- the closest thing to the reference
 * this image can run -- the C++ liblinear the reference's vendored Java port derives from, reached through scikit-learn's
 * `liblinear` solver: where the objectives coincide (prior mean 0, variance 1, bias last and penalised) the TRON trajectories
 * agree in iteration counts and to ~1e-15 in the coefficients at every tolerance (tests/test_oracle.py::
 * test_oracle_tron_follows_c_liblinear_through_scikit_learn). tools/make_java_golden.sh pins it to the Java itself wherever a JDK
 * exists (tests/test_java_golden.py).
 *
 * Path aliases used in citations (all under /root/reference/src/main/java/):
 *   bw/   = de/bwaldvogel/liblinear/
 *   llf/  = com/linkedin/mlease/regression/liblinearfunc/
 *   jobs/ = com/linkedin/mlease/regression/jobs/
 *   models/, consumers/, utils/ = com/linkedin/mlease/{models,regression/consumers,utils}/
 *
 * Data layout mirrors the Java: one array of {int index(1-based); double value}
 * per row (bw/FeatureNode.java:6-7), rows sorted by index, intercept appended
 * last with value bias=1.0 (llf/LibLinearDataset.java:481-482,606-615).
 * The binary variant stores indices only (llf/LibLinearBinaryDataset.java:461-510).
 * Summation order inside every loop is the Java loop order.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle_hooks.h"      /* orc_dataset, the portable exp / log1p of the verification twin, the experiment hooks */

/* which exp / log1p this build evaluates: tests assert "portable" on liboracle_pm.so and "libm" on liboracle.so, so a
 * twin built without -DORC_PORTABLE_MATH (or with the define lost behind another guard) cannot pass for the other */
const char *orc_math_kind(void)
{
#ifdef ORC_PORTABLE_MATH
    return "portable";
#else
    return "libm";
#endif
}

typedef struct orc_tron_stats {
    int newton_iters;    /* accepted + rejected trcg calls (loop trips of bw/Tron.java:66) */
    int accepted;        /* accepted steps */
    int cg_iters;        /* sum of cg_iter */
    int x_passes;        /* Xv + XTv calls, not counting the void Xv(0) */
    int fun_evals, grad_evals, hv_evals;
    double f, gnorm, gnorm1;
} orc_tron_stats;

/* ------------------------------------------------------------------ dataset */

/* Build from CSR (0-based local column ids, intercept NOT included; it is appended
 * here as LibLinearDataset.finish() does, llf/LibLinearDataset.java:592-615).
 * val==NULL -> binary dataset. Rows are sorted by index like
 * llf/LibLinearDataset.java:481-482 (stable insertion sort; duplicates kept). */
orc_dataset *orc_dataset_create(int l, int n_local, const int64_t *row_ptr, const int32_t *col_idx,
                                const float *val, const int8_t *y, const float *weight,
                                const float *offset)
{
    orc_dataset *d = (orc_dataset *)calloc(1, sizeof(*d));
    int64_t nnz = row_ptr[l] - row_ptr[0];
    d->l = l; d->n = n_local; d->binary = (val == NULL);
    d->rp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(l + 1));
    if (d->binary) d->idx = (int *)malloc(sizeof(int) * (size_t)(nnz + l));
    else d->nodes = (orc_node *)malloc(sizeof(orc_node) * (size_t)(nnz + l));
    d->y = (int *)malloc(sizeof(int) * (size_t)l);
    d->weight = (double *)malloc(sizeof(double) * (size_t)l);
    d->offset = (double *)malloc(sizeof(double) * (size_t)l);
    int64_t p = 0;
    for (int i = 0; i < l; i++) {
        d->rp[i] = p;
        int64_t b = row_ptr[i], e = row_ptr[i + 1];
        int64_t start = p;
        for (int64_t k = b; k < e; k++) {
            int id = col_idx[k] + 1;
            double v = d->binary ? 1.0 : (double)val[k];   /* float widened: llf/LibLinearDataset.java:460 */
            int64_t q = p;                                  /* insertion keeps first-seen order among equals */
            if (d->binary) {
                while (q > start && d->idx[q - 1] > id) { d->idx[q] = d->idx[q - 1]; q--; }
                d->idx[q] = id;
            } else {
                while (q > start && d->nodes[q - 1].index > id) { d->nodes[q] = d->nodes[q - 1]; q--; }
                d->nodes[q].index = id; d->nodes[q].value = v;
            }
            p++;
        }
        /* intercept last: index n, value bias = 1.0 (jobs/RegressionAdmmTrain.java:680-684) */
        if (d->binary) d->idx[p] = n_local; else { d->nodes[p].index = n_local; d->nodes[p].value = 1.0; }
        p++;
        d->y[i] = (y[i] == 1) ? 1 : -1;
        d->weight[i] = weight ? (double)weight[i] : 1.0;    /* defaults llf/LibLinearDataset.java:636-649 */
        d->offset[i] = offset ? (double)offset[i] : 0.0;    /* :621-634 */
    }
    d->rp[l] = p;
    return d;
}

void orc_dataset_destroy(orc_dataset *d)
{
    if (!d) return;
    free(d->rp); free(d->nodes); free(d->idx); free(d->y); free(d->weight); free(d->offset); free(d);
}

/* Summation-order experiments live in oracle/experiments.c and reach the functions below through orc_hooks (oracle_hooks.h):
 * all NULL unless a tool installs them; no parity check does. */
orc_exp_hooks orc_hooks = {0};
static _Thread_local int t_site = 0;   /* call site of the next Tron.dot (see orc_exp_hooks.dot) */

/* ------------------------------------------------- LogisticRegressionL2 object */

typedef struct orc_func {
    const orc_dataset *data;
    double *weight;          /* Cp/Cn * data.weight, llf/LogisticRegressionL2.java:93-99 (Cp=Cn=1 in ADMM) */
    double *z, *D, *wa;      /* state arrays :89-91; wa is the per-call temp of Hv :236 */
    const double *priorMean;
    double *priorVar_inv;    /* :107-109 */
    double multiplier;       /* :105, =1 (llf/LibLinear.java:280) */
    orc_tron_stats *st;
} orc_func;

static orc_func *func_create(const orc_dataset *d, const double *priorMean, const double *priorVar,
                             double multiplier, double Cp, double Cn, orc_tron_stats *st)
{
    orc_func *f = (orc_func *)calloc(1, sizeof(*f));
    int l = d->l, n = d->n;
    f->data = d; f->priorMean = priorMean; f->multiplier = multiplier; f->st = st;
    f->z = (double *)calloc((size_t)l, sizeof(double));
    f->D = (double *)calloc((size_t)l, sizeof(double));
    f->wa = (double *)calloc((size_t)l, sizeof(double));
    f->weight = (double *)malloc(sizeof(double) * (size_t)l);
    for (int i = 0; i < l; i++) f->weight[i] = (d->y[i] == 1 ? Cp : Cn) * d->weight[i];
    f->priorVar_inv = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) f->priorVar_inv[i] = 1.0 / priorVar[i];
    return f;
}

static void func_destroy(orc_func *f)
{
    free(f->z); free(f->D); free(f->wa); free(f->weight); free(f->priorVar_inv); free(f);
}

/* llf/LogisticRegressionL2.java:115-129 ; binary llf/LogisticRegressionL2BinaryFeature.java:57-72 */
static void Xv(orc_func *f, const double *v, double *out)
{
    const orc_dataset *d = f->data;
    if (orc_hooks.Xv && orc_hooks.Xv(d, v, out)) return;
    for (int i = 0; i < d->l; i++) {
        double acc = 0;
        if (d->binary) for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++) acc += v[d->idx[k] - 1];
        else for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++) acc += v[d->nodes[k].index - 1] * d->nodes[k].value;
        out[i] = acc;
    }
}

/* llf/LogisticRegressionL2.java:131-150 ; binary :74-93 */
static void XTv(orc_func *f, const double *v, double *out)
{
    const orc_dataset *d = f->data;
    for (int i = 0; i < d->n; i++) out[i] = 0;
    if (orc_hooks.XTv && orc_hooks.XTv(d, v, out)) return;
    for (int i = 0; i < d->l; i++) {
        if (d->binary) for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++) out[d->idx[k] - 1] += v[i];
        else for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++) out[d->nodes[k].index - 1] += v[i] * d->nodes[k].value;
    }
}

/* llf/LogisticRegressionL2.java:156-193 */
static double fun(orc_func *f, const double *w, int count_pass)
{
    const orc_dataset *d = f->data;
    double s = 0;
    Xv(f, w, f->z);
    if (f->st) { f->st->fun_evals++; if (count_pass) f->st->x_passes++; }
    if (orc_hooks.fun_sums && orc_hooks.fun_sums(d, f->weight, f->z, w, f->priorMean, f->priorVar_inv, &s)) return f->multiplier * s;
    for (int i = 0; i < d->l; i++) {
        f->z[i] += d->offset[i];
        double yz = d->y[i] * f->z[i];
        if (yz >= 0) s += f->weight[i] * log1p(exp(-yz));
        else s += f->weight[i] * (-yz + log1p(exp(yz)));
    }
    s = 2.0 * s;
    for (int i = 0; i < d->n; i++) {
        double t = w[i] - f->priorMean[i];
        s += t * t * f->priorVar_inv[i];
    }
    s /= 2.0;
    return f->multiplier * s;
}

/* llf/LogisticRegressionL2.java:199-225 -- consumes z[] left by the preceding fun() */
static void grad(orc_func *f, const double *w, double *g)
{
    const orc_dataset *d = f->data;
    for (int i = 0; i < d->l; i++) {
        f->z[i] = 1 / (1 + exp(-d->y[i] * f->z[i]));
        f->D[i] = f->z[i] * (1 - f->z[i]);
        f->z[i] = f->weight[i] * (f->z[i] - 1) * d->y[i];
    }
    XTv(f, f->z, g);
    if (f->st) { f->st->grad_evals++; f->st->x_passes++; }
    for (int i = 0; i < d->n; i++)
        g[i] = ((w[i] - f->priorMean[i]) * f->priorVar_inv[i] + g[i]) * f->multiplier;
}

/* llf/LogisticRegressionL2.java:231-248 -- uses D[] left by the last grad() */
static void Hv(orc_func *f, const double *s, double *Hs)
{
    const orc_dataset *d = f->data;
    Xv(f, s, f->wa);
    for (int i = 0; i < d->l; i++) f->wa[i] = f->weight[i] * f->D[i] * f->wa[i];
    XTv(f, f->wa, Hs);
    if (f->st) { f->st->hv_evals++; f->st->x_passes += 2; }
    for (int i = 0; i < d->n; i++) Hs[i] = (s[i] * f->priorVar_inv[i] + Hs[i]) * f->multiplier;
}

/* --------------------------------------------------------------- bw/Tron.java */

static void daxpy(int n, double c, const double *v1, double *v2)     /* :190-197 */
{
    if (c == 0) return;
    for (int i = 0; i < n; i++) v2[i] += c * v1[i];
}
static double dot(int n, const double *a, const double *b)           /* :204-213 */
{
    if (orc_hooks.dot) { double r; if (orc_hooks.dot(n, a, b, t_site, &r)) return r; }
    double p = 0;
    for (int i = 0; i < n; i++) p += a[i] * b[i];
    return p;
}
static double euclideanNorm(int n, const double *v)                  /* :220-252 */
{
    if (n < 1) return 0;
    if (n == 1) return fabs(v[0]);
    if (orc_hooks.norm) { double r; if (orc_hooks.norm(n, v, &r)) return r; }
    double scale = 0, sum = 1;
    for (int i = 0; i < n; i++) {
        if (v[i] != 0) {
            double a = fabs(v[i]);
            if (scale < a) { double t = scale / a; sum = 1 + sum * (t * t); scale = a; }
            else { double t = a / scale; sum += t * t; }
        }
    }
    return scale * sqrt(sum);
}
static void scale_(int n, double c, double *v)                       /* :259-265 */
{
    if (c == 1.0) return;
    for (int i = 0; i < n; i++) v[i] *= c;
}

/* bw/Tron.java:126-179 */
static int trcg(orc_func *fo, int n, double delta, const double *g, double *s, double *r,
                double *d, double *Hd)
{
    double one = 1, rTr, rnewTrnew, cgtol;
    for (int i = 0; i < n; i++) { s[i] = 0; r[i] = -g[i]; d[i] = r[i]; }
    cgtol = 0.1 * euclideanNorm(n, g);
    int cg_iter = 0;
    t_site = 0; rTr = dot(n, r, r);
    while (1) {
        if (euclideanNorm(n, r) <= cgtol) break;
        cg_iter++;
        Hv(fo, d, Hd);
        t_site = 1;
        double alpha = rTr / dot(n, d, Hd);
        daxpy(n, alpha, d, s);
        if (euclideanNorm(n, s) > delta) {
            alpha = -alpha;
            daxpy(n, alpha, d, s);
            t_site = 3;
            double std = dot(n, s, d), sts = dot(n, s, s), dtd = dot(n, d, d);
            double dsq = delta * delta;
            double rad = sqrt(std * std + dtd * (dsq - sts));
            if (std >= 0) alpha = (dsq - sts) / (std + rad);
            else alpha = (rad - std) / dtd;
            daxpy(n, alpha, d, s);
            alpha = -alpha;
            daxpy(n, alpha, Hd, r);
            break;
        }
        alpha = -alpha;
        daxpy(n, alpha, Hd, r);
        t_site = 2; rnewTrnew = dot(n, r, r);
        double beta = rnewTrnew / rTr;
        scale_(n, beta, d);
        daxpy(n, one, r, d);
        rTr = rnewTrnew;
    }
    return cg_iter;
}

/* bw/Tron.java:30-124 (LinkedIn-modified: warm start, gnorm1 at w=0, :47-60) */
static void tron(orc_func *fo, int n, double eps, int max_iter, double *w, orc_tron_stats *st)
{
    double eta0 = 1e-4, eta1 = 0.25, eta2 = 0.75;
    double sigma1 = 0.25, sigma2 = 0.5, sigma3 = 4;
    double delta, snorm, one = 1.0, alpha, f, fnew, prered, actred, gs;
    int search = 1, iter = 1, cg_iter;
    double *s = (double *)calloc((size_t)n, sizeof(double));
    double *r = (double *)calloc((size_t)n, sizeof(double));
    double *w_new = (double *)calloc((size_t)n, sizeof(double));
    double *g = (double *)calloc((size_t)n, sizeof(double));
    double *d = (double *)calloc((size_t)n, sizeof(double));
    double *Hd = (double *)calloc((size_t)n, sizeof(double));

    for (int i = 0; i < n; i++) s[i] = 0;
    f = fun(fo, s, 0);                 /* Xv(0): algorithmically void pass, not counted */
    grad(fo, s, g);
    double gnorm1 = euclideanNorm(n, g);

    f = fun(fo, w, 1);
    grad(fo, w, g);
    delta = euclideanNorm(n, g);
    double gnorm = delta;

    if (gnorm <= eps * gnorm1) search = 0;
    iter = 1;
    while (iter <= max_iter && search != 0) {
        cg_iter = trcg(fo, n, delta, g, s, r, d, Hd);
        if (st) { st->newton_iters++; st->cg_iters += cg_iter; }
        memcpy(w_new, w, sizeof(double) * (size_t)n);
        daxpy(n, one, s, w_new);
        t_site = 4; gs = dot(n, g, s);
        t_site = 5; prered = -0.5 * (gs - dot(n, s, r));
        fnew = fun(fo, w_new, 1);
        actred = f - fnew;
        snorm = euclideanNorm(n, s);
        if (iter == 1) delta = fmin(delta, snorm);
        if (fnew - f - gs <= 0) alpha = sigma3;
        else alpha = fmax(sigma1, -0.5 * (gs / (fnew - f - gs)));
        if (actred < eta0 * prered) delta = fmin(fmax(alpha, sigma1) * snorm, sigma2 * delta);
        else if (actred < eta1 * prered) delta = fmax(sigma1 * delta, fmin(alpha * snorm, sigma2 * delta));
        else if (actred < eta2 * prered) delta = fmax(sigma1 * delta, fmin(alpha * snorm, sigma3 * delta));
        else delta = fmax(delta, fmin(alpha * snorm, sigma3 * delta));
        if (actred > eta0 * prered) {
            iter++;
            memcpy(w, w_new, sizeof(double) * (size_t)n);
            f = fnew;
            grad(fo, w, g);
            if (st) st->accepted++;
            gnorm = euclideanNorm(n, g);
            if (gnorm <= eps * gnorm1) break;
        }
        if (f < -1.0e+32) break;
        if (fabs(actred) <= 0 && prered <= 0) break;
        if (fabs(actred) <= 1.0e-12 * fabs(f) && fabs(prered) <= 1.0e-12 * fabs(f)) break;
    }
    if (st) { st->f = f; st->gnorm = gnorm; st->gnorm1 = gnorm1; }
    free(s); free(r); free(w_new); free(g); free(d); free(Hd);
}

/* ---------------------------------------------------- unit-test seams (S1) */

/* One-shot evaluation of fun / grad / Hv at w (and direction s) for finite-difference tests. */
void orc_eval(const orc_dataset *d, const double *w, const double *priorMean, const double *priorVar,
              const double *s, double *f_out, double *g_out, double *Hs_out)
{
    orc_func *fo = func_create(d, priorMean, priorVar, 1.0, 1.0, 1.0, NULL);
    *f_out = fun(fo, w, 0);
    grad(fo, w, g_out);
    if (s && Hs_out) Hv(fo, s, Hs_out);
    func_destroy(fo);
}

/* LibLinear.train numeric core, llf/LibLinear.java:221-312: w[] holds init on entry
 * (already scattered to local index, :236-245) and the TRON result on exit.
 * eps_tron = epsilon * min(pos,neg) / l (:272-276,310-311). */
void orc_train(const orc_dataset *d, double *w, const double *priorMean, const double *priorVar,
               double epsilon, int max_iter, orc_tron_stats *st)
{
    int pos = 0;
    for (int i = 0; i < d->l; i++) if (d->y[i] == 1) pos++;
    int neg = d->l - pos;
    if (st) memset(st, 0, sizeof(*st));
    orc_func *fo = func_create(d, priorMean, priorVar, 1.0, 1.0, 1.0, st);
    tron(fo, d->n, epsilon * (pos < neg ? pos : neg) / d->l, max_iter, w, st);
    func_destroy(fo);
}

/* ------------------------------------------------- posterior variance (LibLinear.train with computePosteriorVar) */

/* llf/LogisticRegressionL2.java:258-297 `hessian` (binary: llf/LogisticRegressionL2BinaryFeature.java:134-178):
 * H = diag(1/priorVar) + X' D X with D_ii = weight_i p_i (1 - p_i), p_i = 1/(1+exp(-y_i (w.x_i + offset_i))); the lower
 * triangle is accumulated row by row, then mirrored. H is n x n row-major (zeroed here: Java's new double[n][n]). */
static double row_q(const orc_func *f, const double *w, int i)
{
    const orc_dataset *d = f->data;
    double score = 0;
    for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++)
        score += d->binary ? w[d->idx[k] - 1] : w[d->nodes[k].index - 1] * d->nodes[k].value;
    score += d->offset[i];
    double p = 1.0 / (1.0 + exp(-d->y[i] * score));
    return f->weight[i] * p * (1 - p);
}

static void hessian(orc_func *f, const double *w, double *H)
{
    const orc_dataset *d = f->data;
    int n = d->n;
    memset(H, 0, sizeof(double) * (size_t)n * (size_t)n);
    for (int k = 0; k < n; k++) H[(size_t)k * n + k] = f->priorVar_inv[k];
    for (int i = 0; i < d->l; i++) {
        double D_ii = row_q(f, w, i);
        for (int64_t a = d->rp[i]; a < d->rp[i + 1]; a++) {
            int m = (d->binary ? d->idx[a] : d->nodes[a].index) - 1;
            double vm = d->binary ? 1.0 : d->nodes[a].value;
            for (int64_t b = d->rp[i]; b < d->rp[i + 1]; b++) {
                int nn = (d->binary ? d->idx[b] : d->nodes[b].index) - 1;
                if (d->binary) H[(size_t)m * n + nn] += D_ii;
                else H[(size_t)m * n + nn] += D_ii * vm * d->nodes[b].value;
                if (m == nn) break;
            }
        }
    }
    for (int m = 0; m < n; m++)
        for (int nn = m + 1; nn < n; nn++) H[(size_t)m * n + nn] = H[(size_t)nn * n + m];
}

/* llf/LogisticRegressionL2.java:304-327 `hessianDiagonal` */
static void hessianDiagonal(orc_func *f, const double *w, double *H)
{
    const orc_dataset *d = f->data;
    for (int k = 0; k < d->n; k++) H[k] = f->priorVar_inv[k];
    for (int i = 0; i < d->l; i++) {
        double q = row_q(f, w, i);
        for (int64_t a = d->rp[i]; a < d->rp[i + 1]; a++) {
            if (d->binary) H[d->idx[a] - 1] += q;      /* the binary class inherits the same loop with value 1 */
            else H[d->nodes[a].index - 1] += q * d->nodes[a].value * d->nodes[a].value;
        }
    }
}

/* org.apache.commons:commons-math3:3.2 (pom.xml:113-117; not vendored in the reference tree), restated from its
 * published algorithm: CholeskyDecomposition(matrix) with the default thresholds (relative symmetry 1e-15, absolute
 * positivity 1e-10) followed by getSolver().getInverse() = solve(identity). A is n x n row-major and is overwritten
 * by L^T (upper triangle); X receives the inverse. Returns 0, or -1 not symmetric / -2 not positive definite. */
static int cholesky_inverse(int n, double *A, double *X)
{
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) {
            double lIJ = A[(size_t)i * n + j], lJI = A[(size_t)j * n + i];
            double maxDelta = 1.0e-15 * fmax(fabs(lIJ), fabs(lJI));
            if (fabs(lIJ - lJI) > maxDelta) return -1;
            A[(size_t)j * n + i] = 0;
        }
    for (int i = 0; i < n; i++) {
        double *ltI = A + (size_t)i * n;
        if (ltI[i] <= 1.0e-10) return -2;
        ltI[i] = sqrt(ltI[i]);
        double inverse = 1.0 / ltI[i];
        for (int q = n - 1; q > i; q--) {
            ltI[q] *= inverse;
            double *ltQ = A + (size_t)q * n;
            for (int p = q; p < n; p++) ltQ[p] -= ltI[q] * ltI[p];
        }
    }
    memset(X, 0, sizeof(double) * (size_t)n * (size_t)n);
    for (int i = 0; i < n; i++) X[(size_t)i * n + i] = 1.0;
    for (int j = 0; j < n; j++) {                      /* L Y = I */
        const double *lJ = A + (size_t)j * n;
        double lJJ = lJ[j];
        double *xJ = X + (size_t)j * n;
        for (int k = 0; k < n; k++) xJ[k] /= lJJ;
        for (int i = j + 1; i < n; i++) {
            double *xI = X + (size_t)i * n;
            double lJI = lJ[i];
            for (int k = 0; k < n; k++) xI[k] -= xJ[k] * lJI;
        }
    }
    for (int j = n - 1; j >= 0; j--) {                 /* L^T X = Y */
        double lJJ = A[(size_t)j * n + j];
        double *xJ = X + (size_t)j * n;
        for (int k = 0; k < n; k++) xJ[k] /= lJJ;
        for (int i = 0; i < j; i++) {
            double *xI = X + (size_t)i * n;
            double lIJ = A[(size_t)i * n + j];
            for (int k = 0; k < n; k++) xI[k] -= xJ[k] * lIJ;
        }
    }
    return 0;
}

/* the decomposition + inverse on its own (tests compare the product's threaded version against it); A is overwritten */
int orc_cholesky_inverse(int n, double *A, double *X) { return cholesky_inverse(n, A, X); }

/* llf/LibLinear.java:314-337: posterior variance at the mode w. full == 0: postVar = 1 / hessianDiagonal;
 * full != 0: postVarMatrix = inverse(hessian) (n x n, may be NULL if only the diagonal is wanted), postVar = its diagonal.
 * hess_out (n x n, optional) receives the Hessian itself for tests. */
int orc_posterior_variance(const orc_dataset *d, const double *w, const double *priorVar, int full,
                           double *post_var, double *post_var_matrix, double *hess_out)
{
    int n = d->n, rc = 0;
    double *zero = (double *)calloc((size_t)n, sizeof(double));
    orc_func *fo = func_create(d, zero, priorVar, 1.0, 1.0, 1.0, NULL);
    if (!full) {
        hessianDiagonal(fo, w, post_var);
        for (int i = 0; i < n; i++) post_var[i] = 1.0 / post_var[i];
    } else {
        double *H = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
        double *V = post_var_matrix ? post_var_matrix : (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
        hessian(fo, w, H);
        if (hess_out) memcpy(hess_out, H, sizeof(double) * (size_t)n * (size_t)n);
        rc = cholesky_inverse(n, H, V);
        if (rc == 0) for (int i = 0; i < n; i++) post_var[i] = V[(size_t)i * n + i];
        if (!post_var_matrix) free(V);
        free(H);
    }
    func_destroy(fo);
    free(zero);
    return rc;
}

/* ------------------------------------------------------------- ADMM driver */

typedef struct orc_admm {
    int nblocks;         /* num.blocks: GLOBAL divisor of the mean (consumers/MeanLinearModelConsumer.java:61) */
    int nlocal;          /* partitions held by this instance (== nblocks unless sharded) */
    int n_global;        /* features + intercept; intercept is global index n_global-1 */
    int nlambda;
    float *lambda, *rho; /* sorted ascending by lambda (jobs/RegressionAdmmTrain.java:636-638) */
    int penalize_intercept;
    int regularizer;     /* 2 = L2 (default), 1 = L1 (jobs/RegressionAdmmTrain.java:143-147,378,406) */
    float *lambda_map;   /* NULL or [n_global] per-feature lambda, NaN = none (lambda.map, :188-197,383-386) */
    orc_dataset **ds;    /* [nlocal] (borrowed) */
    int **l2g;           /* [nlocal][n_local] local->global */
    double *Z;           /* [nlambda][n_global] driver z, double (jobs/...:155,365-405) */
    float *u;            /* [nlocal][nlambda][n_global] u file of the current iteration (float32 on disk) */
    float *B;            /* model file  (float32) */
    float *UPX;          /* uplusx file (float32) */
    int iter_done;       /* iterations completed */
    double *xbar, *ubar; /* [nlambda][n_global] partial / full means */
    orc_tron_stats *stats; /* [nlocal][nlambda] of last iteration */
} orc_admm;

orc_admm *orc_admm_create(int nblocks, int nlocal, int n_global, int nlambda, const float *lambda,
                          const float *rho, int penalize_intercept)
{
    orc_admm *a = (orc_admm *)calloc(1, sizeof(*a));
    a->nblocks = nblocks; a->nlocal = nlocal; a->n_global = n_global; a->nlambda = nlambda;
    a->penalize_intercept = penalize_intercept;
    a->regularizer = 2;
    a->lambda_map = NULL;
    a->lambda = (float *)malloc(sizeof(float) * (size_t)nlambda);
    a->rho = (float *)malloc(sizeof(float) * (size_t)nlambda);
    memcpy(a->lambda, lambda, sizeof(float) * (size_t)nlambda);
    memcpy(a->rho, rho, sizeof(float) * (size_t)nlambda);
    a->ds = (orc_dataset **)calloc((size_t)nlocal, sizeof(*a->ds));
    a->l2g = (int **)calloc((size_t)nlocal, sizeof(*a->l2g));
    size_t zl = (size_t)nlambda * (size_t)n_global, pl = (size_t)nlocal * zl;
    a->Z = (double *)calloc(zl, sizeof(double));
    a->xbar = (double *)calloc(zl, sizeof(double));
    a->ubar = (double *)calloc(zl, sizeof(double));
    a->u = (float *)calloc(pl, sizeof(float));
    a->B = (float *)calloc(pl, sizeof(float));
    a->UPX = (float *)calloc(pl, sizeof(float));
    a->stats = (orc_tron_stats *)calloc((size_t)nlocal * (size_t)nlambda, sizeof(orc_tron_stats));
    return a;
}

void orc_admm_destroy(orc_admm *a)
{
    if (!a) return;
    for (int k = 0; k < a->nlocal; k++) free(a->l2g[k]);
    free(a->lambda_map);
    free(a->lambda); free(a->rho); free(a->ds); free(a->l2g); free(a->Z); free(a->xbar); free(a->ubar);
    free(a->u); free(a->B); free(a->UPX); free(a->stats); free(a);
}

void orc_admm_set_options(orc_admm *a, int regularizer, const float *lambda_map)
{
    a->regularizer = regularizer;
    free(a->lambda_map);
    a->lambda_map = NULL;
    if (lambda_map) {
        a->lambda_map = (float *)malloc(sizeof(float) * (size_t)a->n_global);
        memcpy(a->lambda_map, lambda_map, sizeof(float) * (size_t)a->n_global);
    }
}

void orc_admm_set_partition(orc_admm *a, int k, orc_dataset *d, const int32_t *local_to_global)
{
    a->ds[k] = d;
    a->l2g[k] = (int *)malloc(sizeof(int) * (size_t)d->n);
    for (int j = 0; j < d->n; j++) a->l2g[k][j] = local_to_global[j];
}

/* One reducer invocation, jobs/RegressionAdmmTrain.java:641-718, for (partition k, lambda li). */
static void reduce_one(orc_admm *a, int k, int li, double epsilon, float rho_adapt_rate)
{
    const orc_dataset *d = a->ds[k];
    int n = d->n, ng = a->n_global;
    size_t off = ((size_t)k * (size_t)a->nlambda + (size_t)li) * (size_t)ng;
    const float *u = a->u + off;
    float *B = a->B + off, *UPX = a->UPX + off;
    const double *Z = a->Z + (size_t)li * (size_t)ng;
    const int *l2g = a->l2g[k];

    double rho = (double)a->rho[li];                                   /* :652 */
    if (rho_adapt_rate != 1.0f) rho = rho * (double)rho_adapt_rate;    /* :653-658 */

    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *pm = (double *)malloc(sizeof(double) * (size_t)n);
    double *pv = (double *)malloc(sizeof(double) * (size_t)n);
    for (int j = 0; j < n; j++) {
        double zt = (double)(float)Z[l2g[j]];      /* init-value file is float32: :330-331, models/LinearModel.java:703,716 */
        double uj = (double)u[l2g[j]];
        w[j] = zt;                                 /* initvalue :692-693 */
        pm[j] = -1.0 * uj + 1.0 * zt;              /* priormean.linearCombine(-1, 1, initvalue) :695-697 */
        pv[j] = 1.0 / rho;                         /* defaultPriorVar :705 ; llf/LibLinear.java:243-245 */
    }
    orc_train(d, w, pm, pv, epsilon, 10000, &a->stats[(size_t)k * (size_t)a->nlambda + (size_t)li]);

    /* absent features keep priorMean (llf/LibLinear.java:373-383); outputs float32 (:706-711) */
    for (int gj = 0; gj < ng; gj++) {
        double zt = (double)(float)Z[gj], uj = (double)u[gj];
        double beta = -1.0 * uj + 1.0 * zt;
        B[gj] = (float)beta;
        UPX[gj] = (float)(1.0 * uj + 1.0 * beta);   /* uplusx.linearCombine(1, 1, model) :709-711 */
    }
    for (int j = 0; j < n; j++) {
        int gj = l2g[j];
        double uj = (double)u[gj];
        B[gj] = (float)w[j];
        UPX[gj] = (float)(1.0 * uj + 1.0 * w[j]);
    }
    free(w); free(pm); free(pv);
}

/* All local reducers of one iteration + this shard's partial means.
 * xbar = sum_k (1/N) B_k, ubar = sum_k (1/N) u_k, sequential in partition order:
 * consumers/MeanLinearModelConsumer.java:61 ; models/LinearModel.java:181-201. */
void orc_admm_solve_local(orc_admm *a, double epsilon, float rho_adapt_rate, int nthreads)
{
    int np = a->nlocal * a->nlambda;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int q = 0; q < np; q++) reduce_one(a, q / a->nlambda, q % a->nlambda, epsilon, rho_adapt_rate);
    (void)nthreads;

    size_t ng = (size_t)a->n_global;
    double b = 1.0 / a->nblocks;
    for (int li = 0; li < a->nlambda; li++) {
        double *xb = a->xbar + (size_t)li * ng, *ub = a->ubar + (size_t)li * ng;
        for (size_t j = 0; j < ng; j++) { xb[j] = 0; ub[j] = 0; }
        for (int k = 0; k < a->nlocal; k++) {
            size_t off = ((size_t)k * (size_t)a->nlambda + (size_t)li) * ng;
            for (size_t j = 0; j < ng; j++) {
                xb[j] = 1.0 * xb[j] + b * (double)a->B[off + j];
                ub[j] = 1.0 * ub[j] + b * (double)a->u[off + j];
            }
        }
    }
}

double *orc_admm_xbar(orc_admm *a) { return a->xbar; }
double *orc_admm_ubar(orc_admm *a) { return a->ubar; }

/* z-update (L2), maxdiff/mindiff, and the u file of the NEXT iteration.
 * jobs/RegressionAdmmTrain.java:365-405 (z), :455-472 (diffs), :736-765 (computeU). */
void orc_admm_finish(orc_admm *a, double *maxdiff_out, double *mindiff_out)
{
    size_t ng = (size_t)a->n_global;
    double mindiff = 99999999, maxdiff = 0;
    for (int li = 0; li < a->nlambda; li++) {
        double *Z = a->Z + (size_t)li * ng;
        const double *xb = a->xbar + (size_t)li * ng, *ub = a->ubar + (size_t)li * ng;
        float l = a->lambda[li], r = a->rho[li];
        double weight;
        if (a->regularizer == 2) weight = a->nblocks * r / (l + a->nblocks * r);   /* float arithmetic, then widened: :374-381 */
        else weight = l / (r * a->nblocks + 0.0);                                   /* L1 threshold :411 */
        double diff = 0;
        for (size_t j = 0; j < ng; j++) {
            double zn;
            int icpt = (j == ng - 1);
            if (icpt && !a->penalize_intercept) zn = xb[j] + ub[j];                  /* :392-403 / :438-449 */
            else if (a->regularizer == 2) {
                double c = weight;
                if (!icpt && a->lambda_map && !isnan(a->lambda_map[j]))              /* weightmap :383-386 */
                    c = a->nblocks * r / (a->lambda_map[j] + a->nblocks * r + 0.0);
                zn = 0 + c * xb[j]; zn = 1.0 * zn + c * ub[j];                       /* :387-391 */
            } else {
                zn = 0 + 1.0 * xb[j]; zn = 1.0 * zn + 1.0 * ub[j];                   /* :418-422 */
                if (!icpt) {                                                          /* iterative thresholding :424-436 */
                    if (zn > weight) zn = zn - weight;
                    else if (zn < -weight) zn = zn + weight;
                }
            }
            double dv = fabs(1 * Z[j] + -1 * zn);                               /* :463-464 */
            if (diff < dv) diff = dv;
            Z[j] = zn;
        }
        if (mindiff > diff) mindiff = diff;
        if (maxdiff < diff) maxdiff = diff;
    }
    /* u_k = f32( f32(u_k + beta_k) - Z ), Z in double: :752-757 */
    for (int k = 0; k < a->nlocal; k++)
        for (int li = 0; li < a->nlambda; li++) {
            size_t off = ((size_t)k * (size_t)a->nlambda + (size_t)li) * ng;
            const double *Z = a->Z + (size_t)li * ng;
            for (size_t j = 0; j < ng; j++)
                a->u[off + j] = (float)(1.0 * (double)a->UPX[off + j] + -1.0 * Z[j]);
        }
    a->iter_done++;
    if (maxdiff_out) *maxdiff_out = maxdiff;
    if (mindiff_out) *mindiff_out = mindiff;
}

void orc_admm_iterate(orc_admm *a, double epsilon, float rho_adapt_rate, int nthreads,
                      double *maxdiff_out, double *mindiff_out)
{
    orc_admm_solve_local(a, epsilon, rho_adapt_rate, nthreads);
    orc_admm_finish(a, maxdiff_out, mindiff_out);
}

/* Mean-model warm start, jobs/RegressionAdmmTrain.java:236-276: the RegressionNaiveTrain job per (lambda, partition)
 * (jobs/RegressionNaiveTrain.java:318-404: initParam null, priorMean map null, priorVar map = 1/lambda.map[k] with
 * the intercept forced to 100000 unless penalize.intercept :311-320, default prior mean = prior.mean, default prior
 * var = 1.0/lambda, "epsilon=" option) and this shard's part of meanModel (utils/LinearModelUtils.java:68-86 ->
 * consumers/MeanLinearModelConsumer.java:61: model files are float32, mean accumulates (1/nblocks) * value). */
static void naive_one(orc_admm *a, int k, int li, double epsilon, double prior_mean)
{
    const orc_dataset *d = a->ds[k];
    int n = d->n, ng = a->n_global;
    size_t off = ((size_t)k * (size_t)a->nlambda + (size_t)li) * (size_t)ng;
    float *B = a->B + off;
    const int *l2g = a->l2g[k];
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *pm = (double *)malloc(sizeof(double) * (size_t)n);
    double *pv = (double *)malloc(sizeof(double) * (size_t)n);
    for (int j = 0; j < n; j++) {
        int gj = l2g[j];
        w[j] = 0.0;
        pm[j] = prior_mean;
        pv[j] = 1.0 / (double)a->lambda[li];                                   /* :380 `1.0 / lambda` */
        if (a->lambda_map && !isnan(a->lambda_map[gj])) pv[j] = 1.0 / (double)a->lambda_map[gj];   /* :311-316 */
        if (gj == ng - 1 && !a->penalize_intercept) pv[j] = 100000.0;          /* :317-320 */
    }
    orc_train(d, w, pm, pv, epsilon, 10000, &a->stats[(size_t)k * (size_t)a->nlambda + (size_t)li]);
    for (int gj = 0; gj < ng; gj++) B[gj] = 0.0f;          /* features the partition never saw are not in its model */
    for (int j = 0; j < n; j++) B[l2g[j]] = (float)w[j];   /* models/LinearModel.java:703,716 */
    free(w); free(pm); free(pv);
}

void orc_admm_naive_solve_local(orc_admm *a, double epsilon, double prior_mean, int nthreads)
{
    int np = a->nlocal * a->nlambda;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int q = 0; q < np; q++) naive_one(a, q / a->nlambda, q % a->nlambda, epsilon, prior_mean);
    (void)nthreads;
    size_t ng = (size_t)a->n_global;
    double b = 1.0 / a->nblocks;
    for (int li = 0; li < a->nlambda; li++) {
        double *xb = a->xbar + (size_t)li * ng, *ub = a->ubar + (size_t)li * ng;
        for (size_t j = 0; j < ng; j++) { xb[j] = 0; ub[j] = 0; }
        for (int k = 0; k < a->nlocal; k++) {
            size_t off = ((size_t)k * (size_t)a->nlambda + (size_t)li) * ng;
            for (size_t j = 0; j < ng; j++) xb[j] = 1.0 * xb[j] + b * (double)a->B[off + j];
        }
    }
}

/* z = the mean model, kept in double by the driver (:267); the u file of iteration 1 is empty (:310-312). */
void orc_admm_naive_finish(orc_admm *a)
{
    size_t zl = (size_t)a->nlambda * (size_t)a->n_global;
    memcpy(a->Z, a->xbar, sizeof(double) * zl);
    memset(a->u, 0, sizeof(float) * zl * (size_t)a->nlocal);
}

void orc_admm_get_z(const orc_admm *a, double *Z_out, float *z32_out)
{
    size_t n = (size_t)a->nlambda * (size_t)a->n_global;
    for (size_t i = 0; i < n; i++) {
        if (Z_out) Z_out[i] = a->Z[i];
        if (z32_out) z32_out[i] = (float)a->Z[i];   /* final-model write, models/LinearModel.java:703,716 */
    }
}

void orc_admm_set_state(orc_admm *a, const double *Z, const float *u)
{
    size_t zl = (size_t)a->nlambda * (size_t)a->n_global;
    if (Z) memcpy(a->Z, Z, sizeof(double) * zl);
    if (u) memcpy(a->u, u, sizeof(float) * zl * (size_t)a->nlocal);
}

void orc_admm_get_partition_model(const orc_admm *a, int k, int li, float *beta, float *uplusx, float *u_next)
{
    size_t ng = (size_t)a->n_global;
    size_t off = ((size_t)k * (size_t)a->nlambda + (size_t)li) * ng;
    if (beta) memcpy(beta, a->B + off, sizeof(float) * ng);
    if (uplusx) memcpy(uplusx, a->UPX + off, sizeof(float) * ng);
    if (u_next) memcpy(u_next, a->u + off, sizeof(float) * ng);
}

void orc_admm_get_stats(const orc_admm *a, orc_tron_stats *out)
{
    memcpy(out, a->stats, sizeof(orc_tron_stats) * (size_t)a->nlocal * (size_t)a->nlambda);
}

/* Sum over test rows of LinearModel.evalInstanceAvro(record, loglik=true, num_click_replicates=1, ignore_value)
 * for one model z (global index, intercept last): models/LinearModel.java:491-554 with eval :241-257;
 * called per record by testloglik, jobs/RegressionAdmmTrain.java:779-791. gidx < 0 = name not in the model. */
double orc_test_loglik_sum(int n_global, const double *z, int l, const int64_t *row_ptr, const int32_t *gidx,
                           const double *val, const int8_t *response, const double *weight, const double *offset)
{
    double total = 0.0;
    for (int i = 0; i < l; i++) {
        double result = -log(1 - 1 + 1 * exp(-z[n_global - 1]));          /* :243-244 */
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; k++)
            if (gidx[k] >= 0) result += z[gidx[k]] * (val ? val[k] : 1.0);   /* :249-255 */
        double xbeta = (offset ? offset[i] : 0.0) + result;                /* :544 */
        double w = weight ? weight[i] : 1.0;
        if (response[i] == 1) total += -log1p(exp(-xbeta)) * w;            /* :545-552 */
        else total += -log1p(exp(xbeta)) * w;
    }
    return total;
}

/* RegressionTest's mapper, jobs/RegressionTest.java:147-175: pred = (float) evalInstanceAvro(record, loglik=false,
 * ignore_value) = (float)(offset + eval(keys, values, 1)) with eval of models/LinearModel.java:241-257 (result starts at
 * -log(1-1+1*exp(-intercept)), then += coef*value in record order; names the model does not hold, gidx < 0, are skipped).
 * z = the model as read from the final-model file (float32 widened), intercept last. */
void orc_score_rows(int n_global, const double *z, int l, const int64_t *row_ptr, const int32_t *gidx, const double *val,
                    const double *offset, float *pred)
{
    for (int i = 0; i < l; i++) {
        double result = -log(1 - 1 + 1 * exp(-z[n_global - 1]));
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; k++)
            if (gidx[k] >= 0) result += z[gidx[k]] * (val ? val[k] : 1.0);
        pred[i] = (float)((offset ? offset[i] : 0.0) + result);
    }
}

/* String.valueOf(float) -> Double.parseDouble round trip of liblinear.epsilon
 * (jobs/RegressionAdmmTrain.java:346,620,702 ; llf/LibLinear.java:128-131 ; utils/Util.java:145-155).
 * Shortest decimal that round-trips the float (== Float.toString digits for the
 * values the schedule produces; SURVEY R14 notes pre-JDK19 corner cases <=1e-7 rel). */
double orc_float_to_string_to_double(float e)
{
    char buf[64];
    for (int prec = 2; prec <= 9; prec++) {   /* Java prints at least two significant digits (1.4E-45, not 1E-45) */
        snprintf(buf, sizeof buf, "%.*g", prec, (double)e);
        if (strtof(buf, NULL) == e) break;
    }
    return strtod(buf, NULL);
}

/* The whole loop of RegressionAdmmTrain.run (jobs/...:278-497) without test-loglik:
 * returns the number of iterations executed. diffs[2*i], diffs[2*i+1] = maxdiff, mindiff;
 * eps_used[i] = epsilon seen by the reducers in iteration i+1. */
int orc_admm_run(orc_admm *a, int niter, double epsilon_stop, int aggressive, int nthreads,
                 double *diffs, double *eps_used)
{
    double mindiff = 99999999, maxdiff = 0;
    float liblinearEpsilon = 0.01f;                                   /* :279 */
    int i;
    for (i = 1; i <= niter; i++) {
        if (i > 1 && mindiff < 0.001 && !aggressive) liblinearEpsilon = liblinearEpsilon / 10;   /* :338-341 */
        else if (aggressive && i > 5) liblinearEpsilon = liblinearEpsilon / 10;                   /* :342-345 */
        double eps = orc_float_to_string_to_double(liblinearEpsilon);
        if (eps_used) eps_used[i - 1] = eps;
        orc_admm_iterate(a, eps, 1.0f, nthreads, &maxdiff, &mindiff);
        if (diffs) { diffs[2 * (i - 1)] = maxdiff; diffs[2 * (i - 1) + 1] = mindiff; }
        if (maxdiff < epsilon_stop && liblinearEpsilon <= 0.00001) { i++; break; }               /* :493-496 */
    }
    return i - 1;
}
