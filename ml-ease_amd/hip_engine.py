"""ctypes binding of the C-ABI (include/mlease_admm.h) exported by csrc/libmlease_hip.so.

This is the ONLY compute engine of the package: there is no CPU fallback. Constructing a
:class:`HipAdmmEngine` without the built library or without an MI355X raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .dataset import ModelFittingError, PartitionBlock

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLX_LIB_PATH") or os.path.join(_HERE, "csrc", "libmlease_hip.so")   # MLX_LIB_PATH: A/B builds (tools/ablate.sh)
UNIQUE_ID_BYTES = 128

# every entry point include/mlease_admm.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "mlx_create", "mlx_destroy", "mlx_last_error", "mlx_set_stream", "mlx_set_profiling", "mlx_set_numerics", "mlx_set_option", "mlx_get_option", "mlx_set_problem",
    "mlx_set_regularizer", "mlx_add_partition_csr", "mlx_add_partitions_csr", "mlx_add_partition_dense", "mlx_finalize", "mlx_set_state",
    "mlx_admm_iterate", "mlx_admm_solve_local", "mlx_naive_init", "mlx_naive_solve_local", "mlx_naive_finish", "mlx_consensus_buffer", "mlx_admm_consensus_finish", "mlx_get_z",
    "mlx_get_partition_model", "mlx_get_solve_counters", "mlx_get_dims", "mlx_set_test_data", "mlx_test_loglik", "mlx_solve_one", "mlx_posterior_variance", "mlx_score_rows", "mlx_comm_get_unique_id", "mlx_comm_init",
    "mlx_version",
]


class MlxStats(C.Structure):
    _fields_ = [("maxdiff", C.c_double), ("mindiff", C.c_double), ("solves", C.c_int64),
                ("newton_iters", C.c_int64), ("accepted", C.c_int64), ("cg_iters", C.c_int64),
                ("x_passes_ref", C.c_int64), ("x_passes_dev", C.c_int64), ("ticks", C.c_int64),
                ("alg_bytes_dev", C.c_double), ("xpass_ms", C.c_double), ("total_ms", C.c_double),
                ("xpass_launches", C.c_int64), ("rowpass_ms", C.c_double), ("colpass_ms", C.c_double),
                ("step_ms", C.c_double), ("xpass_busy_ms", C.c_double), ("rowpass_busy_ms", C.c_double),
                ("colpass_busy_ms", C.c_double), ("step_busy_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


EXP_LIB_PATH = os.path.join(_HERE, "csrc", "libmlease_hip_exp.so")    # the same sources + -DMLX_EXPERIMENTAL (csrc/Makefile): tests only
_libs = {}


def load_library(experimental: Optional[bool] = None):
    """dlopen the in-tree HIP library; raises (never falls back) when it is missing. experimental=True (or MLX_EXPERIMENTAL=1
    in the environment when an engine is constructed) loads libmlease_hip_exp.so instead: the product code plus the
    measured-slower / test-only pieces kept as evidence (fused step, shared-X lambda-sweep passes, in-process communicator)."""
    if experimental is None:
        experimental = os.environ.get("MLX_EXPERIMENTAL", "0") not in ("", "0")
    path = EXP_LIB_PATH if experimental else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "the MI355X path has no CPU fallback" % path)
    L = C.CDLL(path)
    vp, i32, i64, f64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_float
    L.mlx_version.restype = C.c_char_p
    L.mlx_last_error.restype = C.c_char_p
    L.mlx_last_error.argtypes = [vp]
    L.mlx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mlx_destroy.argtypes = [vp]
    L.mlx_set_stream.argtypes = [vp, vp]
    L.mlx_set_profiling.argtypes = [vp, C.c_int]
    L.mlx_set_numerics.argtypes = [vp, i32]
    L.mlx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.mlx_get_option.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
    L.mlx_set_problem.argtypes = [vp, i32, i32, vp, vp, i32, i32, vp]
    L.mlx_set_regularizer.argtypes = [vp, i32]
    L.mlx_add_partition_csr.argtypes = [vp, i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp]
    L.mlx_add_partitions_csr.argtypes = [vp, i32] + [vp] * 11
    L.mlx_add_partition_dense.argtypes = [vp, i32, i32, i32, i64, vp, vp, vp, vp, vp, i32]
    L.mlx_finalize.argtypes = [vp]
    L.mlx_set_state.argtypes = [vp, vp, vp]
    L.mlx_admm_iterate.argtypes = [vp, f64, f32, C.POINTER(MlxStats)]
    L.mlx_admm_solve_local.argtypes = [vp, f64, f32, C.POINTER(MlxStats)]
    L.mlx_naive_init.argtypes = [vp, f64, f64, C.POINTER(MlxStats)]
    L.mlx_naive_solve_local.argtypes = [vp, f64, f64, C.POINTER(MlxStats)]
    L.mlx_naive_finish.argtypes = [vp]
    L.mlx_consensus_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.mlx_admm_consensus_finish.argtypes = [vp, C.POINTER(MlxStats)]
    L.mlx_get_z.argtypes = [vp, vp, vp]
    L.mlx_get_partition_model.argtypes = [vp, i32, i32, vp, vp, vp]
    L.mlx_get_solve_counters.argtypes = [vp, vp]
    L.mlx_get_dims.argtypes = [vp, i32, vp]
    L.mlx_set_test_data.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, vp]
    L.mlx_test_loglik.argtypes = [vp, vp]
    L.mlx_solve_one.argtypes = [vp, i32, vp, vp, vp, f64, i32, vp, vp, vp, vp]
    L.mlx_posterior_variance.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp]
    L.mlx_score_rows.argtypes = [vp, i32, vp, i32, i64, vp, vp, vp, vp, vp]
    L.mlx_comm_get_unique_id.argtypes = [vp]
    L.mlx_comm_init.argtypes = [vp, vp, i32, i32]
    _libs[path] = L
    return L


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipAdmmEngine:
    """One handle = one MI355X. Mirrors the reducer + consensus half of RegressionAdmmTrain.run."""

    def __init__(self, n_global: int, lambdas: Sequence[float], rhos: Sequence[float], num_blocks: int,
                 penalize_intercept: bool = False, device: int = 0, regularizer: int = 2,
                 lambda_map: Optional[np.ndarray] = None, stream: Optional[int] = None, profiling: bool = False,
                 numerics: Optional[str] = None, options: Optional[dict] = None):
        """numerics: None (the library's default: fast, or what MLX_FAITHFUL seeds), "fast", "reference_order" or
        "reference_order_one_launch" (include/mlease_admm.h: mlx_set_numerics); options: mlx_set_option key -> value."""
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.mlx_create(int(device), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("mlx_create failed (%d): %s" % (rc, self.L.mlx_last_error(None).decode()))
        order = np.argsort(np.asarray(lambdas, dtype=np.float32), kind="stable")    # ascending, jobs/...:636-638
        self.lambdas = np.ascontiguousarray(np.asarray(lambdas, np.float32)[order])
        self.rhos = np.ascontiguousarray(np.asarray(rhos, np.float32)[order])
        self.n_global, self.n_lambda, self.num_blocks = int(n_global), len(self.lambdas), int(num_blocks)
        self.nlocal = 0
        self.device = int(device)
        lm = None if lambda_map is None else np.ascontiguousarray(lambda_map, np.float32)
        if numerics is not None:
            self.set_option("numerics", numerics)
        for k, v in (options or {}).items():
            self.set_option(k, v)
        if stream is not None:
            self._ck(self.L.mlx_set_stream(self.h, C.c_void_p(stream)))
        self._ck(self.L.mlx_set_problem(self.h, self.n_global, self.n_lambda, _p(self.lambdas), _p(self.rhos),
                                        self.num_blocks, int(penalize_intercept), _p(lm)))
        if regularizer != 2:
            self._ck(self.L.mlx_set_regularizer(self.h, int(regularizer)))
        if profiling:
            self._ck(self.L.mlx_set_profiling(self.h, 1))

    def _ck(self, rc: int):
        if rc != 0:
            msg = self.L.mlx_last_error(self.h).decode()
            if rc in (-4, -5):
                raise ModelFittingError("Model fitting error! " + msg)      # jobs/RegressionAdmmTrain.java:713-716
            raise RuntimeError("mlease_hip error %d: %s" % (rc, msg))

    # -- data -------------------------------------------------------------------------------------
    def add_partition(self, b: PartitionBlock):
        rp = np.ascontiguousarray(b.row_ptr, np.int64)
        ci = np.ascontiguousarray(b.col_idx, np.int32)
        val = None if b.val is None else np.ascontiguousarray(b.val, np.float32)
        y = np.ascontiguousarray(b.y, np.int8)
        wt = np.ascontiguousarray(b.weight, np.float32)
        off = np.ascontiguousarray(b.offset, np.float32)
        l2g = np.ascontiguousarray(b.local_to_global, np.int32)
        self._ck(self.L.mlx_add_partition_csr(self.h, int(b.partition_id), int(b.l), int(b.n_local), int(rp[-1]),
                                              _p(rp), _p(ci), _p(val), _p(y), _p(wt), _p(off), _p(l2g)))
        self.nlocal += 1

    def add_partitions(self, blocks: Sequence[PartitionBlock]):
        """Several CSR partitions in one call (mlx_add_partitions_csr): host preparation on a thread pool."""
        if not blocks:
            return
        keep = []                                   # the arrays must outlive the call

        def arr(a, dt):
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.ctypes.data

        n = len(blocks)
        has_val = any(b.val is not None for b in blocks)       # mixed calls are fine: per-entry NULL = binary.feature
        pid = np.array([b.partition_id for b in blocks], np.int32)
        ls = np.array([b.l for b in blocks], np.int32)
        nl = np.array([b.n_local for b in blocks], np.int32)
        nnz = np.array([len(b.col_idx) for b in blocks], np.int64)
        PtrArr = C.c_void_p * n
        rp = PtrArr(*[arr(b.row_ptr, np.int64) for b in blocks])
        ci = PtrArr(*[arr(b.col_idx, np.int32) for b in blocks])
        vv = PtrArr(*[(arr(b.val, np.float32) if b.val is not None else None) for b in blocks]) if has_val else None
        yy = PtrArr(*[arr(b.y, np.int8) for b in blocks])
        ww = PtrArr(*[arr(b.weight, np.float32) for b in blocks])
        oo = PtrArr(*[arr(b.offset, np.float32) for b in blocks])
        lg = PtrArr(*[arr(b.local_to_global, np.int32) for b in blocks])
        self._ck(self.L.mlx_add_partitions_csr(self.h, n, _p(pid), _p(ls), _p(nl), _p(nnz), C.cast(rp, C.c_void_p), C.cast(ci, C.c_void_p),
                                               None if vv is None else C.cast(vv, C.c_void_p), C.cast(yy, C.c_void_p),
                                               C.cast(ww, C.c_void_p), C.cast(oo, C.c_void_p), C.cast(lg, C.c_void_p)))
        self.nlocal += n

    def add_partition_dense(self, partition_id: int, X: np.ndarray, y: np.ndarray, weight=None, offset=None,
                            local_to_global=None):
        X = np.ascontiguousarray(X, np.float32)
        l, nf = X.shape
        y = np.ascontiguousarray(y, np.int8)
        wt = None if weight is None else np.ascontiguousarray(weight, np.float32)
        off = None if offset is None else np.ascontiguousarray(offset, np.float32)
        l2g = np.arange(nf + 1, dtype=np.int32) if local_to_global is None else np.ascontiguousarray(local_to_global, np.int32)
        self._ck(self.L.mlx_add_partition_dense(self.h, int(partition_id), l, nf, nf, _p(X), _p(y), _p(wt), _p(off),
                                                _p(l2g), 0))
        self.nlocal += 1

    def add_partition_dense_device(self, partition_id: int, x_ptr: int, l: int, n_feat: int, ld: int, y_ptr: int,
                                   weight_ptr: Optional[int] = None, offset_ptr: Optional[int] = None):
        """X / y / weight / offset are device pointers (e.g. torch tensors' data_ptr()); copied by the library."""
        l2g = np.arange(n_feat + 1, dtype=np.int32)
        self._ck(self.L.mlx_add_partition_dense(self.h, int(partition_id), int(l), int(n_feat), int(ld),
                                                C.c_void_p(x_ptr), C.c_void_p(y_ptr),
                                                None if weight_ptr is None else C.c_void_p(weight_ptr),
                                                None if offset_ptr is None else C.c_void_p(offset_ptr), _p(l2g), 1))
        self.nlocal += 1

    def set_option(self, key: str, value) -> None:
        """mlx_set_option: per-handle behaviour (numerics, tick_streams, grid_rounded_dots, ...)."""
        if isinstance(value, bool):
            value = int(value)
        self._ck(self.L.mlx_set_option(self.h, str(key).encode(), str(value).encode()))

    def get_option(self, key: str, size: int = 64) -> str:
        buf = C.create_string_buffer(size)
        self._ck(self.L.mlx_get_option(self.h, str(key).encode(), buf, size))
        return buf.value.decode()

    def tick_log(self):
        """The last solve_local's batches of 4 ticks: [(ticks queued, problems done, us since the solve's first launch), ...] -- the
        done counts are read one batch late, the time differences are the batches' durations on the GPU."""
        s = self.get_option("tick_log", 1 << 17)
        return [tuple(float(x) for x in rec.split(":")) for rec in s.split(";") if rec]

    def set_profiling(self, enable, one_stream: bool = False):
        """Per-launch-class HIP events on / off (every tick stream carries its own chain of marks). one_stream=True also keeps all ticks
        on ONE stream, so that a launch's duration is the kernel's alone (measurement only)."""
        self._ck(self.L.mlx_set_profiling(self.h, (2 if one_stream else 1) if enable else 0))

    def finalize(self):
        self._ck(self.L.mlx_finalize(self.h))

    def set_state(self, Z=None, u=None):
        Z_ = None if Z is None else np.ascontiguousarray(Z, np.float64)
        u_ = None if u is None else np.ascontiguousarray(u, np.float32)
        self._ck(self.L.mlx_set_state(self.h, _p(Z_), _p(u_)))

    # -- one ADMM iteration -------------------------------------------------------------------------
    def iterate(self, epsilon: float, rho_adapt_rate: float = 1.0) -> MlxStats:
        st = MlxStats()
        self._ck(self.L.mlx_admm_iterate(self.h, float(epsilon), float(rho_adapt_rate), C.byref(st)))
        return st

    def solve_local(self, epsilon: float, rho_adapt_rate: float = 1.0) -> MlxStats:
        st = MlxStats()
        self._ck(self.L.mlx_admm_solve_local(self.h, float(epsilon), float(rho_adapt_rate), C.byref(st)))
        return st

    # -- mean-model warm start (initialize.boost.rate; jobs/RegressionAdmmTrain.java:236-276) ---------
    def naive_init(self, epsilon: float, prior_mean: float = 0.0) -> MlxStats:
        st = MlxStats()
        self._ck(self.L.mlx_naive_init(self.h, float(epsilon), float(prior_mean), C.byref(st)))
        return st

    def naive_solve_local(self, epsilon: float, prior_mean: float = 0.0) -> MlxStats:
        st = MlxStats()
        self._ck(self.L.mlx_naive_solve_local(self.h, float(epsilon), float(prior_mean), C.byref(st)))
        return st

    def naive_finish(self) -> None:
        self._ck(self.L.mlx_naive_finish(self.h))

    def consensus_buffer(self) -> Tuple[int, int]:
        ptr, cnt = C.c_void_p(), C.c_size_t()
        self._ck(self.L.mlx_consensus_buffer(self.h, C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value

    def consensus_tensor(self):
        """torch CUDA tensor aliasing the device buffer [xbar | ubar] (for torch.distributed.all_reduce)."""
        import torch

        ptr, cnt = self.consensus_buffer()

        class _Alias:
            __cuda_array_interface__ = {"shape": (cnt,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Alias(), device="cuda:%d" % self.device)

    def consensus_finish(self) -> MlxStats:
        st = MlxStats()
        self._ck(self.L.mlx_admm_consensus_finish(self.h, C.byref(st)))
        return st

    # -- results ----------------------------------------------------------------------------------
    def z(self) -> Tuple[np.ndarray, np.ndarray]:
        Z = np.empty((self.n_lambda, self.n_global), np.float64)
        z32 = np.empty((self.n_lambda, self.n_global), np.float32)
        self._ck(self.L.mlx_get_z(self.h, _p(Z), _p(z32)))
        return Z, z32

    def partition_model(self, local_index: int, lambda_index: int):
        b = np.empty(self.n_global, np.float32)
        upx = np.empty(self.n_global, np.float32)
        un = np.empty(self.n_global, np.float32)
        self._ck(self.L.mlx_get_partition_model(self.h, int(local_index), int(lambda_index), _p(b), _p(upx), _p(un)))
        return b, upx, un

    def set_test_data(self, row_ptr, global_idx, val, response, weight=None, offset=None):
        """Test rows in GLOBAL feature ids (-1 = not in the model); see mlx_set_test_data."""
        rp = np.ascontiguousarray(row_ptr, np.int64)
        gi = np.ascontiguousarray(global_idx, np.int32)
        v = None if val is None else np.ascontiguousarray(val, np.float64)
        y = np.ascontiguousarray(response, np.int8)
        w = None if weight is None else np.ascontiguousarray(weight, np.float64)
        o = None if offset is None else np.ascontiguousarray(offset, np.float64)
        self._ck(self.L.mlx_set_test_data(self.h, len(rp) - 1, int(rp[-1]), _p(rp), _p(gi), _p(v), _p(y), _p(w), _p(o)))

    def test_loglik_sums(self) -> np.ndarray:
        out = np.zeros(self.n_lambda, np.float64)
        self._ck(self.L.mlx_test_loglik(self.h, _p(out)))
        return out

    def dims(self, local_index: int = -1) -> dict:
        out = np.zeros(6, np.int32)
        self._ck(self.L.mlx_get_dims(self.h, int(local_index), _p(out)))
        return dict(zip(("n_global", "n_lambda", "partitions_local", "num_blocks", "n_local", "l"), (int(v) for v in out)))

    def solve_counters(self) -> np.ndarray:
        out = np.zeros((self.nlocal * self.n_lambda, 4), np.int32)
        self._ck(self.L.mlx_get_solve_counters(self.h, _p(out)))
        return out

    def solve_one(self, local_index: int, init: np.ndarray, prior_mean: np.ndarray, prior_var: np.ndarray,
                  epsilon: float, max_iter: int = 10000):
        """LibLinear.train seam (llf/LibLinear.java:200-228) on one uploaded partition."""
        w = np.array(init, dtype=np.float64, copy=True)
        pm = np.ascontiguousarray(prior_mean, np.float64)
        pv = np.ascontiguousarray(prior_var, np.float64)
        cnt = np.zeros(4, np.int32)
        f, gn, gn1 = C.c_double(), C.c_double(), C.c_double()
        self._ck(self.L.mlx_solve_one(self.h, int(local_index), _p(w), _p(pm), _p(pv), float(epsilon), int(max_iter),
                                      _p(cnt), C.byref(f), C.byref(gn), C.byref(gn1)))
        return w, cnt, (f.value, gn.value, gn1.value)

    def posterior_variance(self, local_index: int, w: np.ndarray, prior_var: np.ndarray, full: bool = False):
        """LibLinear.train's computePosteriorVar tail (llf/LibLinear.java:314-337) at the mode w:
        -> (post_var[n_local], post_var_matrix[n_local, n_local] or None, Gram-kernel ms or None)."""
        w_ = np.ascontiguousarray(w, np.float64)
        pv = np.ascontiguousarray(prior_var, np.float64)
        out = np.empty(len(w_))
        M = np.empty((len(w_), len(w_))) if full else None
        ms = C.c_double(0.0)
        self._ck(self.L.mlx_posterior_variance(self.h, int(local_index), _p(w_), _p(pv), int(bool(full)), _p(out), _p(M),
                                               C.byref(ms)))
        return out, M, (ms.value if full else None)

    # -- RCCL -------------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = load_library().mlx_comm_get_unique_id(buf)
        if rc != 0:
            raise RuntimeError("mlx_comm_get_unique_id failed (%d)" % rc)
        return buf.raw

    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        self._ck(self.L.mlx_comm_init(self.h, buf, int(nranks), int(rank)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.mlx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipScorer:
    """RegressionTest's mapper on the GPU (mlx_score_rows): a bare handle, no training partitions."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.mlx_create(int(device), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("mlx_create failed (%d): %s" % (rc, (self.L.mlx_last_error(None) or b"").decode()))

    def score_rows(self, model32: np.ndarray, row_ptr, global_idx, val, offset=None) -> np.ndarray:
        m = np.ascontiguousarray(model32, np.float32)
        rp = np.ascontiguousarray(row_ptr, np.int64)
        gi = np.ascontiguousarray(global_idx, np.int32)
        v = None if val is None else np.ascontiguousarray(val, np.float64)
        o = None if offset is None else np.ascontiguousarray(offset, np.float64)
        out = np.empty(len(rp) - 1, np.float32)
        rc = self.L.mlx_score_rows(self.h, len(m), _p(m), len(rp) - 1, int(rp[-1]), _p(rp), _p(gi), _p(v), _p(o), _p(out))
        if rc != 0:
            raise RuntimeError("mlx_score_rows failed (%d): %s" % (rc, (self.L.mlx_last_error(self.h) or b"").decode()))
        return out

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.mlx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
