#!/usr/bin/env python3
"""bench_gram.py -- the fp64-MFMA Gram kernel (full posterior covariance, mlx_posterior_variance) on one
config-#2 partition (15 625 x 1000 dense): achieved TFLOP/s against the MI355X fp64 matrix peak.

    python tools/bench_gram.py [--rows 15625] [--features 1000] [--reps 5]
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FP64_MATRIX_PEAK_TFLOPS = 78.6      # AMD MI355X datasheet (fp64 matrix); not listed in the in-image guide


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=15625)
    ap.add_argument("--features", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import mlease_amd  # noqa: F401
    from mlease_amd.hip_engine import HipAdmmEngine
    rng = np.random.default_rng(1)
    X = rng.normal(0, 1, (a.rows, a.features)).astype(np.float32)
    y = np.where(rng.random(a.rows) < 0.3, 1, -1).astype(np.int8)
    eng = HipAdmmEngine(a.features + 1, [1.0], [1.0], 1)
    eng.add_partition_dense(0, X, y)
    eng.finalize()
    w = rng.normal(0, 0.1, a.features + 1)
    pv = np.ones(a.features + 1)
    ms, wall = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        _, _, m = eng.posterior_variance(0, w, pv, True)
        wall.append(time.perf_counter() - t0)
        ms.append(m)
    nb = (a.features + 127) // 128
    blocks = nb * (nb + 1) // 2
    flops_exec = 2.0 * blocks * 128 * 128 * a.rows          # what the kernel executes (lower-triangle 128x128 blocks)
    flops_alg = 1.0 * a.rows * (a.features + 1) * (a.features + 2)   # 2 * l * n(n+1)/2: the reference's triangle loop
    best = min(ms)
    print(json.dumps({"workload": "X'DX fp64 Gram, %d x %d dense partition" % (a.rows, a.features), "gram_ms": ms,
                      "tflops_executed": round(flops_exec / best / 1e9, 2), "tflops_algorithmic": round(flops_alg / best / 1e9, 2),
                      "peak_tflops_fp64_matrix": FP64_MATRIX_PEAK_TFLOPS,
                      "frac_of_peak_executed": round(flops_exec / best / 1e9 / FP64_MATRIX_PEAK_TFLOPS, 3),
                      "wall_s_incl_host_cholesky": [round(x, 3) for x in wall]}))


if __name__ == "__main__":
    main()
