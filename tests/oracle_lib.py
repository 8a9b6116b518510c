"""ctypes wrapper around oracle/liboracle.so (the C restatement of the reference path).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_ASAN = os.environ.get("MLX_ASAN", "0") not in ("", "0")          # tools/run_asan.sh: the ASan + UBSan builds (make -C oracle asan)
_LIB_PATH = os.path.join(_ORACLE_DIR, "liboracle_asan.so" if _ASAN else "liboracle.so")
_LIB_PM_PATH = os.path.join(_ORACLE_DIR, "liboracle_pm_asan.so" if _ASAN else "liboracle_pm.so")     # verification twin: portable exp/log1p (oracle/Makefile)


class TronStats(C.Structure):
    _fields_ = [("newton_iters", C.c_int), ("accepted", C.c_int), ("cg_iters", C.c_int),
                ("x_passes", C.c_int), ("fun_evals", C.c_int), ("grad_evals", C.c_int),
                ("hv_evals", C.c_int), ("f", C.c_double), ("gnorm", C.c_double), ("gnorm1", C.c_double)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("admm_oracle.c", "synth.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"] + (["asan"] if _ASAN else ["liboracle.so", "liboracle_pm.so"]))
    return _LIB_PATH


_libs = {}


def lib(pm: bool = False):
    """pm=True: the verification twin whose exp/log1p are ml-ease_amd/csrc/portable_math.h (same restatement otherwise)."""
    if pm not in _libs:
        path = _LIB_PM_PATH if pm else _LIB_PATH
        if not os.path.exists(path):
            build(force=True)
        L = C.CDLL(path)
        vp, i32, f64, f32 = C.c_void_p, C.c_int, C.c_double, C.c_float
        L.orc_math_kind.restype = C.c_char_p
        kind = L.orc_math_kind().decode()
        if kind != ("portable" if pm else "libm"):     # a twin whose -DORC_PORTABLE_MATH got lost must not pass for the other
            raise RuntimeError("%s evaluates %s exp/log1p" % (path, kind))
        L.orc_dataset_create.restype = vp
        L.orc_dataset_create.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp]
        L.orc_dataset_destroy.argtypes = [vp]
        L.orc_eval.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_train.argtypes = [vp, vp, vp, vp, f64, i32, vp]
        L.orc_posterior_variance.argtypes = [vp, vp, vp, i32, vp, vp, vp]
        L.orc_posterior_variance.restype = i32
        L.orc_admm_create.restype = vp
        L.orc_admm_create.argtypes = [i32, i32, i32, i32, vp, vp, i32]
        L.orc_admm_destroy.argtypes = [vp]
        L.orc_admm_set_partition.argtypes = [vp, i32, vp, vp]
        L.orc_admm_set_options.argtypes = [vp, i32, vp]
        L.orc_admm_solve_local.argtypes = [vp, f64, f32, i32]
        L.orc_admm_naive_solve_local.argtypes = [vp, f64, f64, i32]
        L.orc_admm_naive_finish.argtypes = [vp]
        L.orc_admm_xbar.restype = C.POINTER(C.c_double)
        L.orc_admm_xbar.argtypes = [vp]
        L.orc_admm_ubar.restype = C.POINTER(C.c_double)
        L.orc_admm_ubar.argtypes = [vp]
        L.orc_admm_finish.argtypes = [vp, vp, vp]
        L.orc_admm_iterate.argtypes = [vp, f64, f32, i32, vp, vp]
        L.orc_admm_get_z.argtypes = [vp, vp, vp]
        L.orc_admm_set_state.argtypes = [vp, vp, vp]
        L.orc_admm_get_partition_model.argtypes = [vp, i32, i32, vp, vp, vp]
        L.orc_admm_get_stats.argtypes = [vp, vp]
        L.orc_float_to_string_to_double.restype = f64
        L.orc_float_to_string_to_double.argtypes = [f32]
        L.orc_test_loglik_sum.restype = f64
        L.orc_test_loglik_sum.argtypes = [i32, vp, i32, vp, vp, vp, vp, vp, vp]
        L.orc_score_rows.argtypes = [i32, vp, i32, vp, vp, vp, vp, vp]
        L.orc_synth_dense.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, i32, i32, f64, vp, vp, vp]
        L.orc_admm_run.restype = i32
        L.orc_admm_run.argtypes = [vp, i32, f64, i32, i32, vp, vp]
        _libs[pm] = L
    return _libs[pm]


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleDataset:
    """One partition in the reference's row-sparse layout (built from the C-ABI CSR block)."""

    def __init__(self, l, n_local, row_ptr, col_idx, val, y, weight, offset, pm: bool = False):
        self.l, self.n = int(l), int(n_local)
        self.L = lib(pm)
        self._keep = [np.ascontiguousarray(row_ptr, np.int64), np.ascontiguousarray(col_idx, np.int32),
                      None if val is None else np.ascontiguousarray(val, np.float32),
                      np.ascontiguousarray(y, np.int8), np.ascontiguousarray(weight, np.float32),
                      np.ascontiguousarray(offset, np.float32)]
        k = self._keep
        self.h = self.L.orc_dataset_create(self.l, self.n, _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), _p(k[4]), _p(k[5]))
        self._keep = None            # orc_dataset_create copies everything into its own row-sparse layout

    @classmethod
    def from_block(cls, b, pm: bool = False) -> "OracleDataset":
        return cls(b.l, b.n_local, b.row_ptr, b.col_idx, b.val, b.y, b.weight, b.offset, pm=pm)

    def eval(self, w, prior_mean, prior_var, s=None):
        w = np.ascontiguousarray(w, np.float64)
        pm = np.ascontiguousarray(prior_mean, np.float64)
        pv = np.ascontiguousarray(prior_var, np.float64)
        f = C.c_double()
        g = np.empty(self.n)
        Hs = np.empty(self.n) if s is not None else None
        s_ = None if s is None else np.ascontiguousarray(s, np.float64)
        self.L.orc_eval(self.h, _p(w), _p(pm), _p(pv), _p(s_), C.byref(f), _p(g), _p(Hs))
        return f.value, g, Hs

    def train(self, init, prior_mean, prior_var, epsilon, max_iter=10000):
        w = np.array(init, dtype=np.float64, copy=True)
        pm = np.ascontiguousarray(prior_mean, np.float64)
        pv = np.ascontiguousarray(prior_var, np.float64)
        st = TronStats()
        self.L.orc_train(self.h, _p(w), _p(pm), _p(pv), float(epsilon), int(max_iter), C.byref(st))
        return w, st

    def posterior_variance(self, w, prior_var, full):
        """LibLinear.train's computePosteriorVar tail (llf/LibLinear.java:314-337): -> (post_var, matrix or None, H or None)."""
        w = np.ascontiguousarray(w, np.float64)
        pv = np.ascontiguousarray(prior_var, np.float64)
        out = np.empty(self.n)
        M = np.empty((self.n, self.n)) if full else None
        H = np.empty((self.n, self.n)) if full else None
        rc = self.L.orc_posterior_variance(self.h, _p(w), _p(pv), int(bool(full)), _p(out), _p(M), _p(H))
        if rc != 0:
            raise ArithmeticError("CholeskyDecomposition failed (%d)" % rc)
        return out, M, H

    def __del__(self):
        try:
            self.L.orc_dataset_destroy(self.h)
        except Exception:
            pass


class OracleAdmm:
    """The reference ADMM loop over in-memory partition blocks (CPU oracle)."""

    def __init__(self, blocks: Sequence, n_global: int, lambdas: Sequence[float], rhos: Sequence[float],
                 num_blocks: Optional[int] = None, penalize_intercept: bool = False, regularizer: int = 2,
                 lambda_map: Optional[np.ndarray] = None, pm: bool = False):
        self.L = lib(pm)
        order = np.argsort(np.asarray(lambdas, dtype=np.float32), kind="stable")
        self.lambdas = np.ascontiguousarray(np.asarray(lambdas, dtype=np.float32)[order])
        self.rhos = np.ascontiguousarray(np.asarray(rhos, dtype=np.float32)[order])
        self.nlocal = len(blocks)
        self.N = int(num_blocks if num_blocks is not None else len(blocks))
        self.ng, self.nl = int(n_global), len(lambdas)
        self.h = self.L.orc_admm_create(self.N, self.nlocal, self.ng, self.nl, _p(self.lambdas), _p(self.rhos),
                                       int(penalize_intercept))
        if regularizer != 2 or lambda_map is not None:
            lm = None if lambda_map is None else np.ascontiguousarray(lambda_map, np.float32)
            self.L.orc_admm_set_options(self.h, int(regularizer), _p(lm))
        self.ds: List[OracleDataset] = []
        for k, b in enumerate(blocks):
            d = OracleDataset.from_block(b, pm=pm)
            self.ds.append(d)
            l2g = np.ascontiguousarray(b.local_to_global, np.int32)
            self.L.orc_admm_set_partition(self.h, k, d.h, _p(l2g))

    def solve_local(self, epsilon, rho_adapt_rate=1.0, nthreads=1):
        self.L.orc_admm_solve_local(self.h, float(epsilon), float(rho_adapt_rate), int(nthreads))

    def naive_solve_local(self, epsilon, prior_mean=0.0, nthreads=1):
        self.L.orc_admm_naive_solve_local(self.h, float(epsilon), float(prior_mean), int(nthreads))

    def naive_finish(self):
        self.L.orc_admm_naive_finish(self.h)

    def partial_means(self):
        n = self.nl * self.ng
        xb = np.ctypeslib.as_array(self.L.orc_admm_xbar(self.h), shape=(n,))
        ub = np.ctypeslib.as_array(self.L.orc_admm_ubar(self.h), shape=(n,))
        return xb, ub        # views into the oracle's buffers (writable: all-reduce in place)

    def finish(self):
        mx, mn = C.c_double(), C.c_double()
        self.L.orc_admm_finish(self.h, C.byref(mx), C.byref(mn))
        return mx.value, mn.value

    def iterate(self, epsilon, rho_adapt_rate=1.0, nthreads=1):
        mx, mn = C.c_double(), C.c_double()
        self.L.orc_admm_iterate(self.h, float(epsilon), float(rho_adapt_rate), int(nthreads), C.byref(mx), C.byref(mn))
        return mx.value, mn.value

    def run(self, niter, epsilon_stop=1e-4, aggressive=False, nthreads=1):
        diffs = np.zeros(2 * niter)
        eps = np.zeros(niter)
        done = self.L.orc_admm_run(self.h, int(niter), float(epsilon_stop), int(aggressive), int(nthreads),
                                  _p(diffs), _p(eps))
        return done, diffs.reshape(-1, 2)[:done], eps[:done]

    def z(self):
        Z = np.empty((self.nl, self.ng))
        z32 = np.empty((self.nl, self.ng), np.float32)
        self.L.orc_admm_get_z(self.h, _p(Z), _p(z32))
        return Z, z32

    def set_state(self, Z=None, u=None):
        Z_ = None if Z is None else np.ascontiguousarray(Z, np.float64)
        u_ = None if u is None else np.ascontiguousarray(u, np.float32)
        self.L.orc_admm_set_state(self.h, _p(Z_), _p(u_))

    def partition_model(self, k, li):
        b = np.empty(self.ng, np.float32)
        upx = np.empty(self.ng, np.float32)
        un = np.empty(self.ng, np.float32)
        self.L.orc_admm_get_partition_model(self.h, int(k), int(li), _p(b), _p(upx), _p(un))
        return b, upx, un

    def stats(self):
        arr = (TronStats * (self.nlocal * self.nl))()
        self.L.orc_admm_get_stats(self.h, arr)
        return list(arr)

    def __del__(self):
        try:
            self.L.orc_admm_destroy(self.h)
        except Exception:
            pass


def test_loglik_sum(z, row_ptr, gidx, val, response, weight=None, offset=None) -> float:
    z = np.ascontiguousarray(z, np.float64)
    rp = np.ascontiguousarray(row_ptr, np.int64)
    gi = np.ascontiguousarray(gidx, np.int32)
    v = None if val is None else np.ascontiguousarray(val, np.float64)
    y = np.ascontiguousarray(response, np.int8)
    w = None if weight is None else np.ascontiguousarray(weight, np.float64)
    o = None if offset is None else np.ascontiguousarray(offset, np.float64)
    return lib().orc_test_loglik_sum(len(z), _p(z), len(rp) - 1, _p(rp), _p(gi), _p(v), _p(y), _p(w), _p(o))


test_loglik_sum.__test__ = False


def float_to_string_to_double(e) -> float:
    return lib().orc_float_to_string_to_double(float(np.float32(e)))


def score_rows(model32, row_ptr, gidx, val, offset=None) -> np.ndarray:
    """RegressionTest mapper (jobs/RegressionTest.java:147-175) on CSR rows with global ids; model as float32."""
    z = np.ascontiguousarray(np.asarray(model32, np.float32).astype(np.float64))
    rp = np.ascontiguousarray(row_ptr, np.int64)
    gi = np.ascontiguousarray(gidx, np.int32)
    v = None if val is None else np.ascontiguousarray(val, np.float64)
    o = None if offset is None else np.ascontiguousarray(offset, np.float64)
    out = np.empty(len(rp) - 1, np.float32)
    lib().orc_score_rows(len(z), _p(z), len(rp) - 1, _p(rp), _p(gi), _p(v), _p(o), _p(out))
    return out


def synth_dense(row0: int, rows: int, nfeat: int, beta: np.ndarray, seed: int, stream: int = 0, bias: float = -1.0, stride: int = 1):
    """C twin of tools/synth_data.dense_rows_np (oracle/synth.c): (X float32 [rows, nfeat], y int8)."""
    X = np.empty((rows, nfeat), np.float32)
    y = np.empty(rows, np.int8)
    b = np.ascontiguousarray(beta, np.float64)
    lib().orc_synth_dense(int(seed), int(stream), int(row0), int(stride), int(rows), int(nfeat), float(bias), _p(b), _p(X), _p(y))
    return X, y
