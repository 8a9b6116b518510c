#!/usr/bin/env python3
"""Full-size one-hot partitions (configs[2] shape): HIP path vs the oracle vs the oracle on row-permuted data.
Shows whether a GPU/oracle difference is inside the reference algorithm's own order sensitivity."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, _p)
import mlease_amd  # noqa
from mlease_amd.hip_engine import HipAdmmEngine
import oracle_lib as ol
from fixtures import permute_rows
import bench_sparse as bs

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(bs.SEED)
blocks, ng = bs.gen(39063 * K, K, rng)
lam, rho = [1.0], [1.0]
def counters(oc): return np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in oc.stats()])
oc = ol.OracleAdmm(blocks, ng, lam, rho, num_blocks=256)
pb = [permute_rows(b, np.random.default_rng(7 + i)) for i, b in enumerate(blocks)]
op = ol.OracleAdmm(pb, ng, lam, rho, num_blocks=256)
eng = HipAdmmEngine(ng, lam, rho, 256)
for b in blocks: eng.add_partition(b)
eng.finalize()
def rel(a, b):
    fl = 1e-2 * np.max(np.abs(b)); return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), fl)))
for it in range(3):
    oc.solve_local(0.01, 1.0, nthreads=K); op.solve_local(0.01, 1.0, nthreads=K); eng.solve_local(0.01, 1.0)
    co, cp, cg = counters(oc), counters(op), eng.solve_counters()[:K]
    print("it", it, "oracle", co.tolist(), "perm", cp.tolist(), "gpu", cg.tolist())
    for k in range(K):
        bo = oc.partition_model(k, 0)[0]; bp = op.partition_model(k, 0)[0]; bg = eng.partition_model(k, 0)[0]
        print("  part %d: perm-vs-oracle %.3e   gpu-vs-oracle %.3e   max|beta| %.3f" % (k, rel(bp, bo), rel(bg, bo), np.max(np.abs(bo))))
    # keep all three on the oracle's state so each iteration compares like with like
    xb, ub = oc.partial_means(); oc.finish()
    Z = oc.z()[0]
    u = np.stack([oc.partition_model(k, 0)[2] for k in range(K)])[:, None, :]
    op.set_state(Z, u); eng.set_state(Z, u)

if len(sys.argv) > 2:
    # ---- whole ADMM runs (driver eps schedule) from z = u = 0: HIP vs oracle vs row-permuted oracle, per iteration
    from mlease_amd import admm
    NIT = int(sys.argv[2])
    test = blocks[-1]
    tr_blocks, tr_perm = blocks[:-1], pb[:-1]
    Kt = K - 1
    oc = ol.OracleAdmm(tr_blocks, ng, lam, rho)
    op = ol.OracleAdmm(tr_perm, ng, lam, rho)
    eng = HipAdmmEngine(ng, lam, rho, Kt)
    for b in tr_blocks: eng.add_partition(b)
    eng.finalize()
    gi = test.local_to_global[test.col_idx].astype(np.int32)
    trow = (test.row_ptr, gi, None, np.where(test.y == 1, 1, 0).astype(np.int8))
    eng.set_test_data(*trow)
    e = np.float32(0.01); md = [99999999.0] * 3
    print("ADMM from zero, %d partitions, test rows %d" % (Kt, test.l))
    for it in range(1, NIT + 1):
        if it > 1 and md[0] < 0.001: e = np.float32(e / np.float32(10))
        ee = admm.float_string_roundtrip(e)
        mo = oc.iterate(ee, 1.0, nthreads=Kt); mp = op.iterate(ee, 1.0, nthreads=Kt); st = eng.iterate(ee)
        md[0] = mo[1]
        zo, zp, zg = oc.z()[0][0], op.z()[0][0], eng.z()[0][0]
        sc = np.max(np.abs(zo))
        llo = ol.test_loglik_sum(zo, *trow, None, None) / test.l
        llg = float(eng.test_loglik_sums()[0]) / test.l
        print("it %2d eps %-8g maxdiff oracle %.4e gpu %.4e | |z_perm-z_orc|/max %.2e  |z_gpu-z_orc|/max %.2e | test loglik oracle %.8f gpu %.8f"
              % (it, ee, mo[0], st.maxdiff, np.max(np.abs(zp - zo)) / sc, np.max(np.abs(zg - zo)) / sc, llo, llg))
