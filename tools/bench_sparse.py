#!/usr/bin/env python3
"""bench_sparse.py -- development bench for the CSR path (BASELINE configs[2]: synthetic one-hot, 20 categorical
fields x 5000 levels = 100 000 binary features, Zipf(1.1) levels, ~20 nnz/row + intercept, rare positives).

Not the driver's bench (that is bench.py on configs[1]); this one exists to measure and tune the sparse kernels.

    python tools/bench_sparse.py [--rows 10000000] [--partitions 256] [--steps 3] [--warmup 1] [--lambdas 1.0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

SEED = 20260925
FIELDS, LEVELS = 20, 5000


def gen(rows, partitions, rng, valued=False):
    """-> list of PartitionBlock (binary, partition-local ids), n_global."""
    from mlease_amd.dataset import PartitionBlock
    p = np.arange(1, LEVELS + 1, dtype=np.float64) ** -1.1
    cdf = np.cumsum(p / p.sum())
    beta = rng.normal(0, 0.3, FIELDS * LEVELS)
    ng = FIELDS * LEVELS + 1
    blocks = []
    per = (rows + partitions - 1) // partitions
    for k in range(partitions):
        l = min(per, rows - k * per)
        lev = np.searchsorted(cdf, rng.random((l, FIELDS))).astype(np.int32)
        lev = np.minimum(lev, LEVELS - 1)
        gid = lev + (np.arange(FIELDS, dtype=np.int32) * LEVELS)[None, :]          # global feature id per entry
        logit = beta[gid].sum(axis=1) - 3.0
        y = np.where(rng.random(l) < 1 / (1 + np.exp(-logit)), 1, -1).astype(np.int8)
        uniq, inv = np.unique(gid.reshape(-1), return_inverse=True)                 # partition-local compaction
        ci = np.sort(inv.reshape(l, FIELDS).astype(np.int32), axis=1).reshape(-1)
        val = None
        if valued:      # (--valued: the same pattern with real values, for the HASVAL kernels; labels stay those of the binary model)
            val = rng.lognormal(0.0, 0.3, l * FIELDS).astype(np.float32)
        blocks.append(PartitionBlock(k, l, len(uniq) + 1, np.arange(0, (l + 1) * FIELDS, FIELDS, dtype=np.int64), ci, val, y,
                                     np.ones(l, np.float32), np.zeros(l, np.float32),
                                     np.concatenate([uniq.astype(np.int32), [ng - 1]]).astype(np.int32)))
    return blocks, ng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--partitions", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lambdas", type=str, default="1.0")
    ap.add_argument("--cpu-sample", type=int, default=0)
    ap.add_argument("--no-profile", action="store_true", help="no per-launch-class HIP events (needed for MLX_STREAMS=2)")
    ap.add_argument("--valued", action="store_true", help="real-valued entries (the HASVAL kernels) instead of binary.feature")
    ap.add_argument("--check", type=int, default=0, help="verify the first N partitions' models of iteration 1 against the oracle")
    args = ap.parse_args()

    import mlease_amd  # noqa: F401
    from mlease_amd import admm
    from mlease_amd.hip_engine import HipAdmmEngine

    rng = np.random.default_rng(SEED)
    t0 = time.time()
    blocks, ng = gen(args.rows, args.partitions, rng, args.valued)
    tgen = time.time() - t0
    lam = sorted(float(x) for x in args.lambdas.split(","))
    rho = [1.0 if l <= 100 else 10.0 for l in lam]
    eng = HipAdmmEngine(ng, lam, rho, args.partitions, profiling=not args.no_profile)
    t0 = time.time()
    eng.add_partitions(blocks)
    eng.finalize()
    tup = time.time() - t0
    nnz = sum(b.nnz for b in blocks)
    nloc = np.array([b.n_local for b in blocks])

    e = np.float32(0.01)
    mindiff = 99999999.0
    acc = dict(solves=0, cg=0, newton=0, pref=0, pdev=0, ticks=0, alg=0.0, xms=0.0, tms=0.0, rms=0.0, cms=0.0, sms=0.0, rbusy=0.0, cbusy=0.0, sbusy=0.0, xbusy=0.0)
    recs = []
    for it in range(1, args.warmup + args.steps + 1):
        if it > 1 and mindiff < 0.001:
            e = np.float32(e / np.float32(10))
        if it == args.warmup + 1:
            tstart = time.perf_counter()
        st = eng.iterate(admm.float_string_roundtrip(e))
        mindiff = st.mindiff
        recs.append((it, st.ticks, st.cg_iters, st.total_ms, st.xpass_ms, st.maxdiff))
        if it > args.warmup:
            acc["solves"] += st.solves; acc["cg"] += st.cg_iters; acc["newton"] += st.newton_iters
            acc["pref"] += st.x_passes_ref; acc["pdev"] += st.x_passes_dev; acc["ticks"] += st.ticks
            acc["alg"] += st.alg_bytes_dev; acc["xms"] += st.xpass_ms; acc["tms"] += st.total_ms
            acc["rms"] += st.rowpass_ms; acc["cms"] += st.colpass_ms; acc["sms"] += st.step_ms
            acc["rbusy"] += st.rowpass_busy_ms; acc["cbusy"] += st.colpass_busy_ms; acc["sbusy"] += st.step_busy_ms; acc["xbusy"] += st.xpass_busy_ms
        if it == 1 and args.check:
            import oracle_lib as ol
            oc = ol.OracleAdmm(blocks[:args.check], ng, lam, rho, num_blocks=args.partitions)
            oc.solve_local(0.01, 1.0, nthreads=min(args.check, os.cpu_count() or 1))
            cnt = np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in oc.stats()])
            gcnt = eng.solve_counters()[:args.check * len(lam)]
            worst = 0.0
            for k in range(args.check):
                for li in range(len(lam)):
                    b, _, _ = eng.partition_model(k, li)
                    bo, _, _ = oc.partition_model(k, li)
                    fl = 1e-2 * np.max(np.abs(bo))
                    worst = max(worst, float(np.max(np.abs(b.astype(np.float64) - bo) / np.maximum(np.abs(bo), fl))))
            print("check: counters equal = %s, worst rel err = %.3e" % (bool(np.array_equal(cnt, gcnt)), worst), file=sys.stderr)
    dt = time.perf_counter() - tstart
    out = {"workload": "one-hot %d rows x %d features, %d partitions, lambdas %s" % (args.rows, ng - 1, args.partitions, lam),
           "nnz": int(nnz), "n_local_mean": float(nloc.mean()), "n_local_max": int(nloc.max()), "gen_s": round(tgen, 1), "upload_s": round(tup, 1),
           "solves_per_s": round(acc["solves"] / dt, 1), "ms_per_step": round(dt * 1e3 / args.steps, 3),
           "x_passes_ref_per_s": round(acc["pref"] / dt, 1), "x_passes_dev_per_s": round(acc["pdev"] / dt, 1),
           "ticks_per_step": acc["ticks"] / args.steps, "cg_per_solve": acc["cg"] / max(1, acc["solves"]),
           # two tick streams (default): a class's launches overlap the other half's, *_ms are SUMS of launch durations, *_busy_ms the time
           # with at least one launch of the class running; MLX_PROFILE_ONE_STREAM=1: every launch alone on the chip, both are the same
           "xpass_GBps_alg": round(acc["alg"] / max(1e-9, acc["xbusy"] * 1e-3) / 1e9, 1), "xpass_share": round(acc["xbusy"] / (dt * 1e3), 3),
           "device_ms_share": round(acc["tms"] / (dt * 1e3), 3),
           "us_per_tick": {"row": round(1e3 * acc["rbusy"] / max(1, acc["ticks"]), 1), "col": round(1e3 * acc["cbusy"] / max(1, acc["ticks"]), 1),
                           "step": round(1e3 * acc["sbusy"] / max(1, acc["ticks"]), 1), "definition": "busy time of the class (union of its launch intervals) per tick"},
           "us_per_tick_sum_of_launch_durations": {"row": round(1e3 * acc["rms"] / max(1, acc["ticks"]), 1), "col": round(1e3 * acc["cms"] / max(1, acc["ticks"]), 1),
                                                   "step": round(1e3 * acc["sms"] / max(1, acc["ticks"]), 1)},
           "iters": recs}
    try:                                         # timing-experiment builds only (tools/ablate_build.sh -DMLX_PHASE_TIMING)
        import ctypes
        pt = (ctypes.c_double * 16)()
        if eng.L.mlx_debug_phase_times(pt) == 0:
            out["phase_us_sum_over_workgroups"] = [round(v, 1) for v in pt]
    except AttributeError:
        pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
