/*
 * oracle_hooks.h -- types shared by the CPU oracle's translation units, and the HOOKS through which oracle/experiments.c
 * (summation-order experiments: tools/sum_order_experiment.py, tests/test_oracle.py) replaces single reductions of the
 * restatement. Test infrastructure, not product code. Every hook is NULL -- the reference's sequential loops, what every
 * parity check compares against -- unless orc_set_sum_mode / orc_set_dot_site_mode (experiments.c) install it; the hot
 * functions of admm_oracle.c carry one `if (hook)` line each and none of the alternative arithmetic.
 */
#ifndef ORACLE_HOOKS_H
#define ORACLE_HOOKS_H
#include <stdint.h>

#ifdef ORC_PORTABLE_MATH
/* Verification twin (liboracle_pm.so): exp / log1p from the +,-,*,/ implementations that the HIP library's reference-order
 * numerics evaluate as well, so that the two can be compared bit for bit (tests/test_gpu_parity.py). */
#include "portable_math.h"
#define exp pm_exp
#define log1p pm_log1p
#endif

typedef struct { int index; double value; } orc_node;   /* bw/FeatureNode.java */

typedef struct orc_dataset {
    int l, n;            /* rows; features incl. intercept (llf/LibLinearDataset.java:590-594) */
    int binary;          /* LibLinearBinaryDataset */
    int64_t *rp;         /* row pointer into nodes/idx, l+1 */
    orc_node *nodes;     /* non-binary */
    int *idx;            /* binary: 1-based indices */
    int *y;              /* +1/-1 (llf/LibLinearDataset.java:419-423) */
    double *weight, *offset;
} orc_dataset;

/* each hook returns 1 when it produced the result (0: run the reference's loop) */
typedef struct orc_exp_hooks {
    int (*Xv)(const orc_dataset *d, const double *v, double *out);
    int (*XTv)(const orc_dataset *d, const double *v, double *out);
    /* the sums of fun after Xv: z[] += offset (as the loop does), *s_out = (2 * loss + prior) / 2 before the multiplier */
    int (*fun_sums)(const orc_dataset *d, const double *weight, double *z, const double *w, const double *priorMean,
                    const double *priorVar_inv, double *s_out);
    /* site: 0 r.r at the start of trcg, 1 d.Hd, 2 r.r in the loop, 3 the three boundary dots, 4 g.s, 5 s.r */
    int (*dot)(int n, const double *a, const double *b, int site, double *out);
    int (*norm)(int n, const double *v, double *out);
} orc_exp_hooks;
extern orc_exp_hooks orc_hooks;

#endif
