#!/usr/bin/env python3
"""Reference run of BASELINE configs[2] on the CPU ORACLE -> tests/golden/c3_ref_loglik.json.

The one-hot job of bench.py's sparse leg (tools/synth_data.py: 10 M rows x 100 K binary features, 20 nnz/row, 256 partitions,
lambda = 1, rho = 1) from z = 0 for 20 ADMM iterations under the driver's epsilon schedule (jobs/RegressionAdmmTrain.java:279,
338-346), the mean test log-likelihood (jobs/RegressionAdmmTrain.java:766-811) of the consensus after every iteration on
`--test-rows` further rows of the same generator. Target of bench.py's metric (ii) on the sparse leg (`sparse.time_to_ref_loglik`).

    python tests/golden/make_ref_loglik_onehot.py [--threads 8]          (about 3 GB of host memory, a few core-minutes)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import mlease_amd  # noqa: F401,E402
from mlease_amd import admm  # noqa: E402
from mlease_amd.dataset import PartitionBlock  # noqa: E402
import oracle_lib as ol  # noqa: E402
import synth_data as sd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--partitions", type=int, default=256)
    ap.add_argument("--test-rows", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "c3_ref_loglik.json"))
    args = ap.parse_args()
    P, rows = args.partitions, args.rows // args.partitions
    t0 = time.time()
    blocks, ng = [], None
    for k in range(P):
        rp, ci, y, l2g, ng = sd.onehot_partition(k, rows)
        blocks.append(PartitionBlock(k, rows, len(l2g), rp, ci, None, y, np.ones(rows, np.float32), np.zeros(rows, np.float32), l2g))
    trp, tgi, tresp, _ = sd.onehot_test_rows(args.test_rows)
    print("data: %.0f s" % (time.time() - t0), file=sys.stderr)
    oc = ol.OracleAdmm(blocks, ng, [1.0], [1.0])
    e, mind = np.float32(0.01), 99999999.0
    lls, eps_used, diffs, secs = [], [], [], []
    for it in range(1, args.iters + 1):
        if it > 1 and mind < 0.001:
            e = np.float32(e / np.float32(10))
        eps = admm.float_string_roundtrip(e)
        t1 = time.time()
        maxdiff, mind = oc.iterate(eps, 1.0, nthreads=args.threads)
        ll = ol.test_loglik_sum(oc.z()[0][0], trp, tgi, None, tresp) / args.test_rows
        lls.append(ll); eps_used.append(eps); diffs.append(maxdiff); secs.append(time.time() - t1)
        print("iteration %2d: eps %g maxdiff %.6g loglik %.10f (%.0f s)" % (it, eps, maxdiff, ll, secs[-1]), file=sys.stderr)
    import hashlib
    out = {"job": "BASELINE configs[2]: synthetic one-hot %d rows x %d binary features, %d partitions, lambda 1, rho 1; "
                  "mean test loglik of z after each ADMM iteration on %d held-out rows" % (rows * P, ng - 1, P, args.test_rows),
           "generator": "tools/synth_data.py onehot_partition / onehot_test_rows (seed %d)" % sd.SEED,
           "oracle": "oracle/admm_oracle.c through tests/oracle_lib.py, %d threads" % args.threads,
           "rows": rows * P, "partitions": P, "test_rows": args.test_rows, "iterations": args.iters,
           "loglik_by_iteration": lls, "ref_loglik": lls[-1], "epsilon_by_iteration": eps_used, "maxdiff_by_iteration": diffs,
           "oracle_seconds_by_iteration": [round(s, 2) for s in secs],
           "z32_final_sha1": hashlib.sha1(oc.z()[1].tobytes()).hexdigest()}
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote %s: ref loglik %.10f" % (args.out, lls[-1]), file=sys.stderr)


if __name__ == "__main__":
    main()
