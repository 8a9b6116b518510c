#!/usr/bin/env python3
"""Generate the committed C1 fixtures (BASELINE.json configs[0]).

Run in the BUILD container (needs /root/reference/examples/sample-data.avro, which does not exist
on the GPU box):

    python tests/golden/make_golden_c1.py

Writes
  tests/golden/c1_partitions.npz  -- the 1000-row sample, prepared with the deterministic key
                                     row_index % 8 and indexed per partition by
                                     ml-ease_amd/dataset.py (CSR blocks, the C-ABI layout);
  tests/golden/c1_golden.npz      -- outputs of the C oracle (oracle/admm_oracle.c) on those blocks:
                                     lambda=1.0 (rho=1), 20 iterations: Z per iteration (double),
                                     maxdiff/mindiff, liblinear eps, per-partition beta/uplusx/u of
                                     iterations 1, 2, 20, TRON counters per solve per iteration;
                                     lambda={1,10,100,1000} (sample-config.job + one rho=10 case), 6 iterations.

The reference has NO golden vectors for this path (parity unpinned); these are oracle
outputs, cross-checked against the independent NumPy restatement before being written.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import mlease_amd  # noqa: E402,F401
from mlease_amd import avro_io, dataset  # noqa: E402
import oracle_lib as ol  # noqa: E402
import admm_numpy as an  # noqa: E402

SAMPLE = "/root/reference/examples/sample-data.avro"
OUT = os.path.dirname(os.path.abspath(__file__))
NBLOCKS = 8


def eps_schedule_step(e, i, mindiff, aggressive=False):
    """jobs/RegressionAdmmTrain.java:338-345 on a float32."""
    if i > 1 and mindiff < 0.001 and not aggressive:
        e = np.float32(e / np.float32(10))
    elif aggressive and i > 5:
        e = np.float32(e / np.float32(10))
    return e


def run_recording(blocks, ng, lambdas, rhos, niter, keep_models_at):
    oc = ol.OracleAdmm(blocks, ng, lambdas, rhos)
    nl, N = len(lambdas), len(blocks)
    Z = np.zeros((niter, nl, ng))
    diffs = np.zeros((niter, 2))
    eps = np.zeros(niter)
    counters = np.zeros((niter, N * nl, 4), np.int32)
    models = {}
    e = np.float32(0.01)
    mindiff = 99999999.0
    for i in range(1, niter + 1):
        e = eps_schedule_step(e, i, mindiff)
        eps[i - 1] = ol.float_to_string_to_double(e)
        maxdiff, mindiff = oc.iterate(eps[i - 1], 1.0, nthreads=8)
        Z[i - 1] = oc.z()[0]
        diffs[i - 1] = (maxdiff, mindiff)
        for q, s in enumerate(oc.stats()):
            counters[i - 1, q] = (s.newton_iters, s.accepted, s.cg_iters, s.x_passes)
        if i in keep_models_at:
            B = np.zeros((N, nl, ng), np.float32)
            U = np.zeros_like(B)
            Un = np.zeros_like(B)
            for k in range(N):
                for li in range(nl):
                    B[k, li], U[k, li], Un[k, li] = oc.partition_model(k, li)
            models[i] = (B, U, Un)
    return Z, diffs, eps, counters, models


def main():
    recs = avro_io.read_records(SAMPLE)
    rows = dataset.prepare_rows(recs, NBLOCKS, key_fn=lambda i, r: i % NBLOCKS)
    pd = dataset.build_partitions(rows, NBLOCKS)
    part = {"num_blocks": np.int32(NBLOCKS), "n_global": np.int32(pd.n_global),
            "feature_names": np.array(pd.feature_names)}
    for b in pd.blocks:
        p = "p%d_" % b.partition_id
        part[p + "row_ptr"] = b.row_ptr.astype(np.int32)
        part[p + "col_idx"] = b.col_idx.astype(np.int16)
        part[p + "val"] = b.val
        part[p + "y"] = b.y
        part[p + "weight"] = b.weight
        part[p + "offset"] = b.offset
        part[p + "l2g"] = b.local_to_global.astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "c1_partitions.npz"), **part)

    # --- config #1: lambda=1.0, num.blocks=8, 20 iterations
    Z, diffs, eps, counters, models = run_recording(pd.blocks, pd.n_global, [1.0], [1.0], 20, (1, 2, 20))
    # cross-check 1: the oracle's own run() loop gives the same trajectory
    oc = ol.OracleAdmm(pd.blocks, pd.n_global, [1.0], [1.0])
    done, d2, e2 = oc.run(20, nthreads=8)
    assert done == 20 and np.array_equal(d2, diffs) and np.array_equal(e2, eps)
    assert np.array_equal(oc.z()[0], Z[-1])
    # cross-check 2: independent NumPy restatement, float32 outputs identical
    parts = [an.partition_from_csr(b.row_ptr, b.col_idx, b.val, b.y, b.weight, b.offset, b.n_local,
                                   b.local_to_global) for b in pd.blocks]
    na = an.AdmmNumpy(parts, pd.n_global, [1.0])
    na.run(20)
    assert np.array_equal(na.Z.astype(np.float32), Z[-1].astype(np.float32)), "C and NumPy oracles disagree"

    # --- multi-lambda: sample-config.job's 1,10,100 plus 1000 (default rho switches to 10 above 100)
    lam = [1.0, 10.0, 100.0, 1000.0]
    rho = [1.0, 1.0, 1.0, 10.0]
    Zm, diffsm, epsm, countersm, _ = run_recording(pd.blocks, pd.n_global, lam, rho, 6, ())
    nam = an.AdmmNumpy(parts, pd.n_global, lam)
    nam.run(6)
    assert np.array_equal(nam.Z.astype(np.float32), Zm[-1].astype(np.float32)), "multi-lambda oracles disagree"

    gold = {"Z": Z, "diffs": diffs, "eps": eps, "counters": counters,
            "Zm": Zm, "diffsm": diffsm, "epsm": epsm, "countersm": countersm,
            "lambdas_m": np.asarray(lam, np.float32), "rhos_m": np.asarray(rho, np.float32)}
    for it, (B, U, Un) in models.items():
        gold["B_it%d" % it] = B
        gold["UPX_it%d" % it] = U
        gold["Unext_it%d" % it] = Un
    np.savez_compressed(os.path.join(OUT, "c1_golden.npz"), **gold)
    print("wrote fixtures:", {k: os.path.getsize(os.path.join(OUT, k)) for k in ("c1_partitions.npz", "c1_golden.npz")})
    print("final maxdiff", diffs[-1], "passes/solve it1", counters[0, :, 3], "it20", counters[-1, :, 3])


if __name__ == "__main__":
    main()
