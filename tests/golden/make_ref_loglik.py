#!/usr/bin/env python3
"""Reference run of BASELINE configs[1] on the CPU ORACLE -> tests/golden/c2_ref_loglik.json.

The oracle (oracle/admm_oracle.c, the C restatement of the reference's AdmmTrain path) runs the whole synthetic dense
job -- 1 000 000 rows x 1000 features, 64 partitions by row % 64, lambda = 1 (rho = 1), 20 ADMM iterations with the
driver's liblinear-epsilon schedule -- and records the mean test log-likelihood of the consensus model after every
iteration on 100 000 held-out rows (jobs/RegressionAdmmTrain.java:766-811). bench.py's metric (ii), "ADMM wall-clock
to the reference log-likelihood", uses the LAST value as its target; the data come from the integer generator of
tools/synth_data.py, which bench.py reproduces bit for bit on the GPU. (Fixture generator: lives beside the fixture, like
make_golden_c1.py; it is the only place outside tests/, smoke() and bench.py's CPU legs that runs the oracle.)

    python tests/golden/make_ref_loglik.py [--threads 8]          (about 25 GB of host memory, ~1 core-hour)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import oracle_lib as ol          # noqa: E402
import synth_data as sd          # noqa: E402
from mlease_amd.dataset import PartitionBlock   # noqa: E402

ROWS, NFEAT, PARTS, TEST_ROWS, ITERS = 1_000_000, 1000, 64, 100_000, 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--iters", type=int, default=ITERS)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "c2_ref_loglik.json"))
    ap.add_argument("--permute-seed", type=int, default=0,
                    help="!= 0: every partition's rows in a seeded random order (an order Hadoop does not define) -> the oracle's OWN spread; "
                         "write with --out tests/golden/c2_ref_loglik_rowperm.json (-> c2_ref_z_rowperm.npz beside it)")
    args = ap.parse_args()
    beta = sd.dense_beta(NFEAT)
    per = args.rows // PARTS
    blocks = []
    t0 = time.time()
    for k in range(PARTS):
        X, y = ol.synth_dense(k, per, NFEAT, beta, sd.SEED, stride=PARTS)
        if args.permute_seed:
            perm = np.random.default_rng([args.permute_seed, k]).permutation(per)
            X, y = np.ascontiguousarray(X[perm]), np.ascontiguousarray(y[perm])
        blocks.append(PartitionBlock(k, per, NFEAT + 1, np.arange(0, (per + 1) * NFEAT, NFEAT, dtype=np.int64),
                                     np.tile(np.arange(NFEAT, dtype=np.int32), per), X.reshape(-1), y,
                                     np.ones(per, np.float32), np.zeros(per, np.float32), np.arange(NFEAT + 1, dtype=np.int32)))
    oc = ol.OracleAdmm(blocks, NFEAT + 1, [1.0], [1.0])
    for b in blocks:                     # the oracle keeps its own row-sparse copy
        b.val = None; b.col_idx = None
    Xt, yt = ol.synth_dense(ROWS, TEST_ROWS, NFEAT, beta, sd.SEED)           # rows beyond the training range
    trp = np.arange(0, (TEST_ROWS + 1) * NFEAT, NFEAT, dtype=np.int64)
    tgi = np.tile(np.arange(NFEAT, dtype=np.int32), TEST_ROWS)
    resp = np.where(yt == 1, 1, 0).astype(np.int8)
    print("data ready in %.0f s" % (time.time() - t0), file=sys.stderr)
    e = np.float32(0.01)
    mindiff = 99999999.0
    lls, eps_used, diffs, counters = [], [], [], []
    z32_it, Z_it, cnt_it = [], [], []
    for it in range(1, args.iters + 1):
        if it > 1 and mindiff < 0.001:
            e = np.float32(e / np.float32(10))
        eps = ol.float_to_string_to_double(e)
        t1 = time.time()
        maxdiff, mindiff = oc.iterate(eps, 1.0, nthreads=args.threads)
        Z = oc.z()[0][0]
        ll = ol.test_loglik_sum(Z, trp, tgi, Xt.reshape(-1), resp) / TEST_ROWS
        st = oc.stats()
        lls.append(ll); eps_used.append(eps); diffs.append(maxdiff)
        counters.append([int(sum(s.newton_iters for s in st)), int(sum(s.cg_iters for s in st)), int(sum(s.x_passes for s in st))])
        z32_it.append(oc.z()[1][0].copy()); Z_it.append(Z.copy())
        cnt_it.append(np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in st], np.int32))
        print("iteration %2d: eps %g maxdiff %.6g loglik %.10f (%.0f s)" % (it, eps, maxdiff, ll, time.time() - t1), file=sys.stderr)
    out = {"what": "oracle/admm_oracle.c on BASELINE configs[1]: %d x %d, %d partitions (row %% %d), lambda=1, rho=1; "
                   "mean test loglik of z after each ADMM iteration on %d held-out rows" % (args.rows, NFEAT, PARTS, PARTS, TEST_ROWS),
           "generator": "tools/synth_data.py seed %d (training rows 0..%d stream 0, test rows %d.. stream 0)" % (sd.SEED, args.rows - 1, ROWS),
           "rows": args.rows, "features": NFEAT, "partitions": PARTS, "test_rows": TEST_ROWS, "iterations": args.iters,
           "loglik_by_iteration": lls, "ref_loglik": lls[-1], "epsilon_by_iteration": eps_used, "maxdiff_by_iteration": diffs,
           "newton_cg_xpasses_by_iteration": counters,
           "z32_final_sha1": __import__("hashlib").sha1(oc.z()[1].tobytes()).hexdigest(), "permute_seed": args.permute_seed}
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    # the consensus after every iteration (float32 as the final-model file holds it, and the driver's double) and every
    # solve's TRON counters: bench.py compares its full 20-iteration run against these, iteration by iteration
    np.savez_compressed(os.path.splitext(args.out)[0].replace("c2_ref_loglik", "c2_ref_z") + ".npz",
                        z32=np.stack(z32_it), Z=np.stack(Z_it), counters=np.stack(cnt_it))
    print("wrote", args.out, file=sys.stderr)


if __name__ == "__main__":
    main()
