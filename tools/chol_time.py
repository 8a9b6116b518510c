#!/usr/bin/env python3
"""Times the library's host-side Cholesky + inverse (mlx_debug_cholesky_inverse) for several MLX_CHOL_THREADS on this host."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlease_amd  # noqa: F401
from mlease_amd import hip_engine
lib = hip_engine.load_library()
f = lib.mlx_debug_cholesky_inverse
f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
f.restype = ctypes.c_int
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1001
rng = np.random.default_rng(1)
B = rng.normal(size=(2 * n, n))
H = B.T @ B + np.eye(n)
H = (H + H.T) / 2
X = np.empty((n, n))
for th in ("1", "2", "4", "8", "16", "32", "64"):
    os.environ["MLX_CHOL_THREADS"] = th
    ts = []
    for _ in range(3):
        t = time.perf_counter(); rc = f(n, H.ctypes.data, X.ctypes.data); ts.append(time.perf_counter() - t)
    print(th, rc, ["%.4f" % x for x in ts])
print("cores", os.cpu_count())
