"""Reference-order numerics on DENSE tiles (csrc/mlx_ro_dense.h), measured on the GPU at the BASELINE configs[1] shape.

    python tools/ro_dense_probe.py [partitions=64] [iterations=8] [check_partitions=4] [rows_per_partition=15625] [features=1000]

Environment: RO_STREAMS=n (tick streams of the reference-order handle), RO_FAST=1 (also run the fast contract beside it).
1. mlx_set_numerics(REFERENCE_ORDER) on `partitions` tiles: solves/s per ADMM iteration under the driver's epsilon schedule;
2. a one-stream replay of the last iteration with events: microseconds per tick of the row pass, the column pass and the step, and the
   passes' fraction of the HBM peak (algorithmic bytes 4 l n + 8 l + 8 n per pass);
3. bit-identity against the oracle twin (liboracle_pm.so) on the first `check_partitions` partitions for the first 3 iterations.
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

import torch  # noqa: E402
import mlease_amd  # noqa: F401,E402
from mlease_amd.hip_engine import HipAdmmEngine  # noqa: E402
from mlease_amd.dataset import PartitionBlock  # noqa: E402
from mlease_amd import admm  # noqa: E402
import synth_data as sd  # noqa: E402


class Eps:
    def __init__(self):
        self.e, self.mindiff, self.it = np.float32(0.01), 99999999.0, 0

    def next(self):
        self.it += 1
        if self.it > 1 and self.mindiff < 0.001:
            self.e = np.float32(self.e / np.float32(10))
        return admm.float_string_roundtrip(self.e)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nv = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rows = int(sys.argv[4]) if len(sys.argv) > 4 else 15625
    nf = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
    dev = torch.device("cuda:0")
    out = {"partitions": P, "rows": rows, "features": nf}
    sample = []

    def build(numerics, streams=None):
        opts = {"tick_streams": int(streams)} if streams else None
        eng = HipAdmmEngine(nf + 1, [1.0], [1.0], P, numerics=numerics, options=opts)
        for k in range(P):
            X, y = sd.dense_rows_torch(torch, dev, k, rows, nf, stride=P)
            torch.cuda.synchronize()
            eng.add_partition_dense_device(k, X.data_ptr(), rows, nf, nf, y.data_ptr())
            if k < nv and len(sample) < nv:
                sample.append((X.cpu().numpy(), y.cpu().numpy()))
            del X, y
        eng.finalize()
        return eng

    def run(eng, label):
        sched = Eps()
        per, states = [], []
        for it in range(iters):
            eps = sched.next()
            if nv and it < 3 and label == "reference_order":
                states.append((eng.z()[0].copy(), np.stack([eng.partition_model(k, 0)[2] for k in range(nv)])[:, None, :].copy() if it else
                               np.zeros((nv, 1, nf + 1), np.float32), eps))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = eng.solve_local(eps, 1.0)
            dt = time.perf_counter() - t0
            if nv and it < 3 and label == "reference_order":
                states[-1] = states[-1] + ([eng.partition_model(k, 0) for k in range(nv)], eng.solve_counters()[:nv].copy())
            fin = eng.consensus_finish()
            sched.mindiff = fin.mindiff
            per.append({"it": it + 1, "eps": eps, "ms": round(dt * 1e3, 2), "solves_per_s": round(st.solves / dt, 1), "ticks": int(st.ticks), "cg": int(st.cg_iters),
                        "alg_GBps": round(st.alg_bytes_dev / dt / 1e9, 1)})
        # one-stream replay of one more iteration with events
        eng.set_profiling(True, one_stream=True)
        eps = sched.next()
        st = eng.solve_local(eps, 1.0)
        eng.consensus_finish()
        eng.set_profiling(False)
        pass_bytes = P * (4.0 * rows * nf + 8.0 * rows + 8.0 * (nf + 1))
        prof = {"ticks": int(st.ticks), "xpass_ms": round(st.xpass_ms, 3), "row_ms": round(st.rowpass_ms, 3), "col_ms": round(st.colpass_ms, 3), "step_ms": round(st.step_ms, 3),
                "total_ms": round(st.total_ms, 3), "alg_bytes": st.alg_bytes_dev}
        if st.rowpass_ms > 0:
            # (every problem of the list takes part in a launch until it is done: bytes of the ACTIVE problems / the class's time)
            half = st.alg_bytes_dev / 2.0
            prof["row_frac_of_hbm_peak"] = round(half / (st.rowpass_ms * 1e-3) / 8e12, 4)
            prof["col_frac_of_hbm_peak"] = round(half / (st.colpass_ms * 1e-3) / 8e12, 4)
            prof["us_per_tick"] = {"row": round(1e3 * st.rowpass_ms / st.ticks, 1), "col": round(1e3 * st.colpass_ms / st.ticks, 1), "step": round(1e3 * st.step_ms / st.ticks, 1)}
            prof["full_launch_bytes"] = pass_bytes
        elif st.xpass_ms > 0:
            prof["xpass_frac_of_hbm_peak"] = round(st.alg_bytes_dev / (st.xpass_ms * 1e-3) / 8e12, 4)
        return {"kernels": eng.get_option("numerics_kernels"), "tick_streams": eng.get_option("tick_streams"), "dense_tiles": eng.get_option("dense_tiles"),
                "per_iteration": per, "one_stream_profile": prof,
                "solves_per_s_last_half": round(float(np.mean([x["solves_per_s"] for x in per[len(per) // 2:]])), 1)}, states

    if os.environ.get("RO_FAST") == "1":
        eng = build("fast")
        out["fast"], _ = run(eng, "fast")
        eng.close()
    eng = build("reference_order", os.environ.get("RO_STREAMS"))
    out["reference_order"], states = run(eng, "reference_order")
    eng.close()
    if nv and states:
        import oracle_lib as ol
        blocks = []
        for k, (Xh, yh) in enumerate(sample):
            l = Xh.shape[0]
            blocks.append(PartitionBlock(k, l, nf + 1, np.arange(0, (l + 1) * nf, nf, dtype=np.int64), np.tile(np.arange(nf, dtype=np.int32), l),
                                         Xh.reshape(-1), yh, np.ones(l, np.float32), np.zeros(l, np.float32), np.arange(nf + 1, dtype=np.int32)))
        oc = ol.OracleAdmm(blocks, nf + 1, [1.0], [1.0], num_blocks=P, pm=True)
        chk = {"partitions": nv, "solves": 0, "equal_counters": 0, "bit_identical_beta_and_uplusx": 0}
        for Z, u, e, models, gc in states:
            oc.set_state(Z, u)
            oc.solve_local(e, 1.0, nthreads=min(8, nv))
            cc = np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in oc.stats()], np.int32)
            for k in range(nv):
                ob, ou, _ = oc.partition_model(k, 0)
                chk["solves"] += 1
                chk["equal_counters"] += int(np.array_equal(gc[k], cc[k]))
                chk["bit_identical_beta_and_uplusx"] += int(np.array_equal(models[k][0], ob) and np.array_equal(models[k][1], ou))
        out["vs_oracle_twin"] = chk
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
