"""Reference-order numerics on the tick kernels, measured (GPU): BASELINE configs[2]-shaped one-hot partitions.

    python tools/ro_probe.py [partitions=256] [iterations=4] [check_partitions=8] [rows_per_partition=39062]

1. the product path (fast numerics) for `iterations` ADMM iterations from z = 0: solves/s per iteration;
2. the same job with mlx_set_numerics(REFERENCE_ORDER): solves/s per iteration, kernel-class times (HIP events, replay);
3. bit-identity of the reference-order handle against the oracle twin (liboracle_pm.so) on the first `check_partitions` partitions,
   every solve started from the reference-order handle's own state at that iteration.
Prints one JSON record."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import mlease_amd  # noqa: F401,E402
from mlease_amd.hip_engine import HipAdmmEngine  # noqa: E402
from mlease_amd.dataset import PartitionBlock  # noqa: E402
from mlease_amd import admm  # noqa: E402
import synth_data as sd  # noqa: E402
import oracle_lib as ol  # noqa: E402


LIBS = []


def cnts(o):
    return np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in o.stats()], np.int32)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    nv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rows = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000 // 256
    lam = [float(x) for x in os.environ.get("RO_LAMBDAS", "1.0").split(",")]
    rho = [1.0 if l <= 100 else 10.0 for l in lam]
    nl = len(lam)
    blocks, ng = [], None
    for k in range(P):
        rp, ci, y, l2g, ng = sd.onehot_partition(k, rows)
        blocks.append(PartitionBlock(k, rows, len(l2g), rp, ci, None, y, np.ones(rows, np.float32), np.zeros(rows, np.float32), l2g))
    out = {"partitions": P, "rows": rows, "n_local_mean": float(np.mean([b.n_local for b in blocks])), "lambdas": lam}

    def run(numerics, record_states=False, profile=False):
        t0 = time.perf_counter()
        opts = {"tick_streams": int(os.environ["RO_STREAMS"])} if (numerics == "reference_order" and os.environ.get("RO_STREAMS")) else None      # (A/B: tick streams of the reference-order handle)
        eng = HipAdmmEngine(ng, lam, rho, P, numerics=numerics, options=opts)
        eng.add_partitions(blocks)
        eng.finalize()
        prep = time.perf_counter() - t0
        kern = eng.get_option("numerics_kernels")
        LIBS.append(eng.L)
        sys.stderr.write("[ro_probe] %s: tick streams %s (probe rejects %s)\n" % (numerics, eng.get_option("tick_streams"), eng.get_option("stream_probe_rejects")))
        eps = 0.01
        per, states = [], []
        for it in range(iters):
            if record_states:
                Z = eng.z()[0].copy()
                u = np.stack([np.stack([eng.partition_model(k, li)[2] for li in range(nl)]) for k in range(nv)]) if (it and nv) else np.zeros((nv, nl, ng), np.float32)
            t0 = time.perf_counter()
            st = eng.solve_local(eps, 1.0)
            dt = time.perf_counter() - t0
            if record_states:
                pm = [eng.partition_model(k, li) for k in range(nv) for li in range(nl)]
                if nv:
                    states.append((Z, u, eps, np.stack([m[0] for m in pm]), np.stack([m[1] for m in pm]), eng.solve_counters()[:nv * nl].copy()))
            fin = eng.consensus_finish()
            per.append({"it": it + 1, "s": round(dt, 4), "solves_per_s": round(st.solves / dt, 1), "ticks": int(st.ticks), "cg": int(st.cg_iters),
                        "newton": int(st.newton_iters), "maxdiff": fin.maxdiff})
        prof = None
        if profile:
            eng.set_profiling(True, one_stream=True)
            st = eng.solve_local(eps, 1.0)
            eng.consensus_finish()
            prof = {"ticks": int(st.ticks), "row_ms": round(st.rowpass_ms, 2), "col_ms": round(st.colpass_ms, 2), "step_ms": round(st.step_ms, 2),
                    "total_ms": round(st.total_ms, 2), "us_per_tick": {"row": round(1e3 * st.rowpass_ms / st.ticks, 1), "col": round(1e3 * st.colpass_ms / st.ticks, 1),
                                                                        "step": round(1e3 * st.step_ms / st.ticks, 1)}}
            eng.set_profiling(False)
        eng.close()
        return {"kernels": kern, "prep_s": round(prep, 1), "per_iteration": per, "one_stream_profile_of_next_iteration": prof}, states

    if os.environ.get("RO_ONLY") == "1":      # (timing-experiment builds: only the reference-order handle's workgroups in the phase sums)
        out["reference_order"], states = run("reference_order", record_states=True, profile=False)
        out["fast"] = out["reference_order"]
    else:
        out["fast"], _ = run("fast", profile=True)
        out["reference_order"], states = run("reference_order", record_states=True, profile=True)
    try:                                         # timing-experiment builds only (tools/ablate_build.sh -DMLX_PHASE_TIMING)
        import ctypes
        pt = (ctypes.c_double * 16)()
        if LIBS[-1].mlx_debug_phase_times(pt) == 0:
            out["phase_us_sum_over_workgroups"] = [round(v, 1) for v in pt]
    except Exception:
        pass
    try:                                         # timing-experiment builds only (tools/ablate_build.sh -DMLX_RO_PASS_TIMING)
        import ctypes
        pt = (ctypes.c_double * 16)()
        if LIBS[-1].mlx_debug_ropass_times(pt) == 0:
            v = list(pt)
            cg, ev = max(v[8], 1.0), max(v[9], 1.0)
            out["ro_step_pass_us_per_tick"] = {"cg_ticks": v[8], "eval_ticks": v[9], "cg_pass_A": round(v[0] * 0.01 / cg, 1), "cg_pass_B": round(v[1] * 0.01 / cg, 1),
                                               "cg_boundary_pass": round(v[2] * 0.01 / cg, 1), "cg_pass_C": round(v[3] * 0.01 / cg, 1),
                                               "eval_row_folds": round(v[4] * 0.01 / ev, 1), "eval_pass_n": round(v[5] * 0.01 / ev, 1), "eval_copy": round(v[6] * 0.01 / ev, 1)}
    except Exception:
        pass
    f = np.mean([x["solves_per_s"] for x in out["fast"]["per_iteration"][1:]] or [0])
    r = np.mean([x["solves_per_s"] for x in out["reference_order"]["per_iteration"][1:]] or [0])
    out["solves_per_s_after_first_iteration"] = {"fast": round(float(f), 1), "reference_order": round(float(r), 1), "ratio": round(float(f / max(r, 1e-9)), 2)}
    if nv > 0:
        oc = ol.OracleAdmm(blocks[:nv], ng, lam, rho, num_blocks=P, pm=True)
        chk = {"partitions": nv, "solves": 0, "equal_counters": 0, "bit_identical_beta_and_uplusx": 0, "first_mismatch": None}
        for i, (Z, u, e, gb, gu, gc) in enumerate(states):
            oc.set_state(Z, u)
            oc.solve_local(e, 1.0, nthreads=min(16, nv * nl))
            cc = cnts(oc)
            for q in range(nv * nl):
                k, li = divmod(q, nl)
                ob, ou, _ = oc.partition_model(k, li)
                chk["solves"] += 1
                eqc = bool(np.array_equal(gc[q], cc[q]))
                eqb = bool(np.array_equal(gb[q], ob) and np.array_equal(gu[q], ou))
                chk["equal_counters"] += int(eqc)
                chk["bit_identical_beta_and_uplusx"] += int(eqb)
                if not (eqc and eqb) and chk["first_mismatch"] is None:
                    d = np.abs(gb[q].astype(np.float64) - ob.astype(np.float64))
                    chk["first_mismatch"] = {"iteration": i + 1, "partition": k, "lambda": li, "gpu_counters": gc[q].tolist(), "oracle_counters": cc[q].tolist(),
                                             "max_abs_diff_beta": float(d.max()), "differing_coefficients": int((gb[q] != ob).sum())}
        out["vs_oracle_twin"] = chk
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
