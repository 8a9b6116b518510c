/*
 * experiments.c -- SUMMATION-ORDER EXPERIMENTS on the CPU oracle (test infrastructure; never used by a parity check).
 *
 * tools/sum_order_experiment.py and tests/test_oracle.py measure how far ONE changed summation (a tree, a compensated sum, a
 * grid-rounded sum ...) moves the reference's TRON trajectory on one-hot data. The alternative arithmetic
 * lives here, in its own translation unit; oracle/admm_oracle.c -- the restatement of the reference every parity check compares
 * against -- only carries the hooks of oracle_hooks.h, NULL by default.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_hooks.h"

/* orc_set_sum_mode(m): bits of m pick the replaced summations (NOT the reference's arithmetic)
 * orc_set_sum_mode(m), default 0 = the reference's sequential loops, which is what every parity test compares against.
 *   bit 0 (1): Tron.dot as a compensated sum (TwoSum error-free transformation: the sum of the SAME rounded terms, rounded
 *              once -- order-independent up to second-order effects);
 *   bit 1 (2): the row sums of Xv and the column sums of XTv the same way;
 *   bit 2 (4): Tron.euclideanNorm as sqrt of the compensated sum of squares;
 *   bit 3 (8): the loss and prior sums of fun as compensated sums;
 *   bit 4 (16): Tron.euclideanNorm as sqrt of the plain sequential sum of squares (no running scale);
 *   bit 5 (32): Tron.dot as the GRID-ROUNDED sum a parallel kernel can compute: term j rounded to the ulp of the (exact) prefix
 *               sum's binade, the rounded terms added exactly -- an emulation of what the sequential loop does to small terms
 *               once the running sum is large (it absorbs their low bits), without the loop's dependency chain;
 *   bit 6 (64): Tron.dot as a pairwise tree (what a parallel reduction computes);
 *   bit 7 (128) / bit 8 (256): the grid-rounded sum with the grid taken from the prefix at the START of every block of 2048 / 64
 *               elements (cheaper for a kernel: no scan inside the block);
 *   bits 9-12 (512 ... 4096): variants of it (see dot());
 *   bit 13 (8192): the loss sum of fun as a tree over 256-row units (what the dense pass kernel computes);
 *   bit 14 (16384): the loss sum of fun grid-rounded: terms behind the first 256 rows rounded to the ulp of those rows' sum, then added
 *               exactly.
 * Used by tools/sum_order_experiment.py and tests/test_oracle.py to measure how far a summation order (or an exact sum) moves
 * the reference's TRON trajectory on one-hot data; never by a parity check. */
static int g_sum_mode = 0;
/* per call site of Tron.dot (0: r.r at the start of trcg, 1: d.Hd, 2: r.r in the loop, 3: the three boundary dots, 4: g.s, 5: s.r):
 * dot-related bits (1, 32, 64, 128, 256 ...) that replace those of the global mode at that site; -1 = use the global mode */
static int g_site_mode[6] = {-1, -1, -1, -1, -1, -1};
static inline void acc2(double *hi, double *lo, double x)
{
    double s = *hi + x;
    double bb = s - *hi;
    *lo += (*hi - (s - bb)) + (x - bb);
    *hi = s;
}

static int exp_Xv(const orc_dataset *d, const double *v, double *out)
{
    if (g_sum_mode & 2) {            /* experiment: compensated row sums */
        for (int i = 0; i < d->l; i++) {
            double hi = 0, lo = 0;
            if (d->binary) for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++) acc2(&hi, &lo, v[d->idx[k] - 1]);
            else for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++) acc2(&hi, &lo, v[d->nodes[k].index - 1] * d->nodes[k].value);
            out[i] = hi + lo;
        }
        return 1;
    }
    return 0;
}

static int exp_XTv(const orc_dataset *d, const double *v, double *out)
{
    if (g_sum_mode & 2) {            /* experiment: compensated column sums */
        double *lo = (double *)calloc((size_t)d->n, sizeof(double));
        for (int i = 0; i < d->l; i++) {
            if (d->binary) for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++) acc2(&out[d->idx[k] - 1], &lo[d->idx[k] - 1], v[i]);
            else for (int64_t k = d->rp[i]; k < d->rp[i + 1]; k++) acc2(&out[d->nodes[k].index - 1], &lo[d->nodes[k].index - 1], v[i] * d->nodes[k].value);
        }
        for (int i = 0; i < d->n; i++) out[i] += lo[i];
        free(lo);
        return 1;
    }
    return 0;
}

static int exp_fun_sums(const orc_dataset *d, const double *weight, double *z, const double *w, const double *priorMean,
                        const double *priorVar_inv, double *s_out)
{
    double s = 0;
    if (g_sum_mode & (8192 | 16384)) {      /* experiment: the loss sum as a kernel would add it */
        double tot = 0;
        if (g_sum_mode & 8192) {
            for (int u0 = 0; u0 < d->l; u0 += 256) {
                double t[256];
                int m = d->l - u0 < 256 ? d->l - u0 : 256;
                for (int k = 0; k < 256; k++) t[k] = 0;
                for (int k = 0; k < m; k++) {
                    int i = u0 + k;
                    z[i] += d->offset[i];
                    double yz = d->y[i] * z[i];
                    t[k] = (yz >= 0) ? weight[i] * log1p(exp(-yz)) : weight[i] * (-yz + log1p(exp(yz)));
                }
                for (int st = 128; st >= 1; st >>= 1) for (int k = 0; k < st; k++) t[k] += t[k + st];
                tot += t[0];
            }
        } else {
            double hh = 0, hl = 0, rh = 0, rl = 0, u = 0, magic = 0;
            for (int i = 0; i < d->l; i++) {
                z[i] += d->offset[i];
                double yz = d->y[i] * z[i];
                double x = (yz >= 0) ? weight[i] * log1p(exp(-yz)) : weight[i] * (-yz + log1p(exp(yz)));
                if (i < 256) { acc2(&hh, &hl, x); acc2(&rh, &rl, x); }
                else {
                    if (i == 256) { double h = hh + hl; int e; if (h > 0) { frexp(h, &e); u = ldexp(1.0, e - 53); magic = 1.5 * ldexp(1.0, 52) * u; } }
                    double xr = (u > 0 && fabs(x) < ldexp(1.0, 50) * u) ? ((x + magic) - magic) : x;
                    acc2(&rh, &rl, xr);
                }
            }
            tot = rh + rl;
        }
        s = 2.0 * tot;
        for (int i = 0; i < d->n; i++) {
            double t = w[i] - priorMean[i];
            s += t * t * priorVar_inv[i];
        }
        s /= 2.0;
        *s_out = s;
        return 1;
    }
    if (g_sum_mode & 8) {            /* experiment: compensated loss and prior sums */
        double hi = 0, lo = 0;
        for (int i = 0; i < d->l; i++) {
            z[i] += d->offset[i];
            double yz = d->y[i] * z[i];
            if (yz >= 0) acc2(&hi, &lo, weight[i] * log1p(exp(-yz)));
            else acc2(&hi, &lo, weight[i] * (-yz + log1p(exp(yz))));
        }
        double ph = 0, pl = 0;
        for (int i = 0; i < d->n; i++) {
            double t = w[i] - priorMean[i];
            acc2(&ph, &pl, t * t * priorVar_inv[i]);
        }
        s = 2.0 * (hi + lo);
        s += ph + pl;
        s /= 2.0;
        *s_out = s;
        return 1;
    }
    (void)s;
    return 0;
}

static int exp_dot(int n, const double *a, const double *b, int site, double *out)
{
    const int g_sum_mode_global = g_sum_mode;
    const int g_sum_mode = g_site_mode[site] >= 0 ? g_site_mode[site] : g_sum_mode_global;     /* (shadows the global) */
    if (g_sum_mode & 1) {            /* experiment: compensated */
        double hi = 0, lo = 0;
        for (int i = 0; i < n; i++) acc2(&hi, &lo, a[i] * b[i]);
        { *out = hi + lo; return 1; }
    }
    if (g_sum_mode & 32) {           /* experiment: grid-rounded terms */
        double ph = 0, pl = 0;       /* exact prefix (what a scan would provide) */
        double rh = 0, rl = 0;       /* exact sum of the rounded terms */
        for (int i = 0; i < n; i++) {
            double x = a[i] * b[i];
            acc2(&ph, &pl, x);
            double pre = ph + pl;
            if (pre != 0 && x != 0) {
                int e;
                frexp(pre, &e);                          /* |pre| in [2^(e-1), 2^e): ulp = 2^(e-53) */
                double u = ldexp(1.0, e - 53);
                double magic = 1.5 * ldexp(1.0, 52) * u; /* (x + magic) - magic rounds x to a multiple of u (|x| < 2^51 u) */
                double xr = (fabs(x) < ldexp(1.0, 50) * u) ? ((x + magic) - magic) : x;
                acc2(&rh, &rl, xr);
            } else acc2(&rh, &rl, x);
        }
        { *out = rh + rl; return 1; }
    }
    if (g_sum_mode & (128 | 256 | 512 | 1024 | 2048 | 4096)) {  /* experiment: grid-rounded terms, grid per block (512 / 1024: the prefix stops growing after 1 / 4 blocks; 2048 / 4096: ONE grid from the sum of the first 256 / 64 terms) */
        const int B = (g_sum_mode & 256) ? 64 : ((g_sum_mode & 2048) ? 256 : ((g_sum_mode & 4096) ? 64 : 2048));
        const int K = (g_sum_mode & (512 | 2048 | 4096)) ? 1 : ((g_sum_mode & 1024) ? 4 : (1 << 30));
        double ph = 0, pl = 0, rh = 0, rl = 0, pre_cap = 0;
        for (int c0 = 0; c0 < n; c0 += B) {
            double pre = ph + pl;
            if (c0 / B <= K) pre_cap = pre; else pre = pre_cap;
            int e = 0;
            double u = 0;
            if (pre != 0) { frexp(pre, &e); u = ldexp(1.0, e - 53); }
            double magic = 1.5 * ldexp(1.0, 52) * u;
            int m = n - c0 < B ? n - c0 : B;
            for (int q = 0; q < m; q++) {
                double x = a[c0 + q] * b[c0 + q];
                acc2(&ph, &pl, x);
                double xr = (u > 0 && fabs(x) < ldexp(1.0, 50) * u) ? ((x + magic) - magic) : x;
                acc2(&rh, &rl, xr);
            }
        }
        { *out = rh + rl; return 1; }
    }
    if (g_sum_mode & 64) {           /* experiment: pairwise tree over 8-strided blocks of 2048 */
        double tot = 0;
        for (int c0 = 0; c0 < n; c0 += 2048) {
            double t[256];
            int m = n - c0 < 2048 ? n - c0 : 2048;
            for (int k = 0; k < 256; k++) { double p = 0; for (int q = k; q < m; q += 256) p += a[c0 + q] * b[c0 + q]; t[k] = p; }
            for (int st = 128; st >= 1; st >>= 1) for (int k = 0; k < st; k++) t[k] += t[k + st];
            tot += t[0];
        }
        { *out = tot; return 1; }
    }
    return 0;
}

static int exp_norm(int n, const double *v, double *out)
{
    if (g_sum_mode & 4) {            /* experiment: sqrt of the compensated sum of squares */
        double hi = 0, lo = 0;
        for (int i = 0; i < n; i++) acc2(&hi, &lo, v[i] * v[i]);
        { *out = sqrt(hi + lo); return 1; }
    }
    if (g_sum_mode & 16) {           /* experiment: sqrt of the plain sum of squares */
        double p = 0;
        for (int i = 0; i < n; i++) p += v[i] * v[i];
        { *out = sqrt(p); return 1; }
    }
    return 0;
}

static void install(void)
{
    int any_site = 0;
    for (int i = 0; i < 6; i++) any_site |= (g_site_mode[i] >= 0);
    orc_hooks.Xv = (g_sum_mode & 2) ? exp_Xv : NULL;
    orc_hooks.XTv = (g_sum_mode & 2) ? exp_XTv : NULL;
    orc_hooks.fun_sums = (g_sum_mode & (8 | 8192 | 16384)) ? exp_fun_sums : NULL;
    orc_hooks.dot = ((g_sum_mode & (1 | 32 | 64 | 128 | 256 | 512 | 1024 | 2048 | 4096)) || any_site) ? exp_dot : NULL;
    orc_hooks.norm = (g_sum_mode & (4 | 16)) ? exp_norm : NULL;
}
void orc_set_sum_mode(int m) { g_sum_mode = m; install(); }
void orc_set_dot_site_mode(int site, int m) { if (site >= 0 && site < 6) g_site_mode[site] = m; install(); }
int orc_get_sum_mode(void) { return g_sum_mode; }
