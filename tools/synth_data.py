"""Synthetic benchmark data of BASELINE configs[1..4], reproducible bit for bit on any machine.

Dense (configs[1], "C2"): element (row r, column c) of the 1M x 1K matrix is an Irwin-Hall(12) variate built from
INTEGER arithmetic only -- twelve 16-bit chunks of three 64-bit counter hashes, summed, centred and divided by 2^16
(mean 0, variance 1 - 2^-32, |x| <= 6) -- so the torch-on-GPU generator bench.py uses, the NumPy generator the tests
use and the C generator behind the committed reference log-likelihood (tests/golden/make_ref_loglik.py) produce the same
float32 values. Labels: y = +1 iff u_r < sigmoid(x_r . beta* + b) with u_r a 53-bit hash uniform; beta* = 0.1 * IH12,
b = -1. (The fp64 dot product is summed in a different order by rocBLAS, OpenBLAS and the C loop: a label can differ
only when u_r is within ~1e-15 of the threshold.)

One-hot (configs[2..4], "C3"): 20 categorical fields x 5000 levels, level ~ Zipf(1.1) within the field, binary features,
beta* ~ N(0, 0.3^2), intercept -3; NumPy on the host (no committed constant depends on these bits).
"""
from __future__ import annotations

import numpy as np

SEED = 20260925
M64 = (1 << 64) - 1
GOLD = 0x9E3779B97F4A7C15
C1 = 0xBF58476D1CE4E5B9
C2 = 0x94D049BB133111EB
IH_MEAN = 6 * 65535           # 12 chunks, each uniform on 0..65535


def stream_key(seed: int, stream: int, k: int) -> int:
    """64-bit additive key of hash k of a stream (plain Python ints; the same constants go to every backend)."""
    return (seed * 0xD1342543DE82EF95 + stream * 0x2545F4914F6CDD1D + (k + 1) * 0x9E6C63D0676A9A99) & M64


# ---- NumPy ------------------------------------------------------------------------------------------------------
def _mix_np(x: np.ndarray) -> np.ndarray:
    x = (x ^ (x >> np.uint64(30))) * np.uint64(C1)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(C2)
    return x ^ (x >> np.uint64(31))


def _hash_np(counter: np.ndarray, key: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        return _mix_np((counter + np.uint64(1)) * np.uint64(GOLD) + np.uint64(key))


def ih12_np(counter: np.ndarray, seed: int, stream: int) -> np.ndarray:
    """Irwin-Hall(12) integer sum minus its mean, as int64 in (-393210, 393210]; x = value / 65536."""
    counter = counter.astype(np.uint64)
    s = np.zeros(counter.shape, np.int64)
    for k in range(3):
        h = _hash_np(counter, stream_key(seed, stream, k))
        for sh in (0, 16, 32, 48):
            s += ((h >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    return s - IH_MEAN


def uniform53_np(counter: np.ndarray, seed: int, stream: int) -> np.ndarray:
    h = _hash_np(counter.astype(np.uint64), stream_key(seed, stream, 7))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def dense_beta(nfeat: int, seed: int = SEED) -> np.ndarray:
    return 0.1 * (ih12_np(np.arange(nfeat), seed, 1).astype(np.float64) / 65536.0)


def dense_rows_np(row0: int, rows: int, nfeat: int, seed: int = SEED, stream: int = 0, bias: float = -1.0, stride: int = 1):
    """Rows row0, row0+stride, ... (`rows` of them) of the dense matrix: (X float32 [rows, nfeat], y int8 +1/-1).
    Partition k of P = (row0=k, stride=P): SURVEY 8d's `row % P` assignment."""
    r = (row0 + stride * np.arange(rows)).astype(np.uint64)
    counter = r[:, None] * np.uint64(nfeat) + np.arange(nfeat, dtype=np.uint64)[None, :]
    X = (ih12_np(counter, seed, stream).astype(np.float32)) / np.float32(65536.0)
    logit = X.astype(np.float64) @ dense_beta(nfeat, seed) + bias
    u = uniform53_np(r, seed, stream + 100)
    y = np.where(u < 1.0 / (1.0 + np.exp(-logit)), 1, -1).astype(np.int8)
    return X, y


# ---- torch (any device) ---------------------------------------------------------------------------------------------
def _s64(v: int) -> int:
    v &= M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr_t(torch, x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _hash_t(torch, counter, key: int):
    x = (counter + 1) * _s64(GOLD) + _s64(key)          # int64 arithmetic wraps (two's complement)
    x = (x ^ _lsr_t(torch, x, 30)) * _s64(C1)
    x = (x ^ _lsr_t(torch, x, 27)) * _s64(C2)
    return x ^ _lsr_t(torch, x, 31)


def dense_rows_torch(torch, device, row0: int, rows: int, nfeat: int, seed: int = SEED, stream: int = 0, bias: float = -1.0,
                     stride: int = 1):
    """Same values as dense_rows_np, generated on `device`: (X float32 [rows, nfeat], y int8)."""
    r = row0 + stride * torch.arange(rows, dtype=torch.int64, device=device)
    counter = r[:, None] * nfeat + torch.arange(nfeat, dtype=torch.int64, device=device)[None, :]
    s = torch.zeros(counter.shape, dtype=torch.int64, device=device)
    for k in range(3):
        h = _hash_t(torch, counter, stream_key(seed, stream, k))
        for sh in (0, 16, 32, 48):
            s += _lsr_t(torch, h, sh) & 0xFFFF if sh else h & 0xFFFF
        del h
    X = (s - IH_MEAN).to(torch.float32) / 65536.0
    del s, counter
    beta = torch.from_numpy(dense_beta(nfeat, seed)).to(device)
    logit = X.double() @ beta + bias
    hu = _hash_t(torch, r, stream_key(seed, stream + 100, 7))
    u = _lsr_t(torch, hu, 11).to(torch.float64) * (1.0 / 9007199254740992.0)
    y = torch.where(u < torch.sigmoid(logit), 1, -1).to(torch.int8)
    return X, y


# ---- one-hot (configs[2..4]) ----------------------------------------------------------------------------------------
FIELDS, LEVELS = 20, 5000


def onehot_partition(pid: int, rows: int, seed: int = SEED):
    """Partition `pid` of the one-hot data set as a binary CSR block in partition-local ids:
    (row_ptr int64, col_idx int32 sorted per row, y int8, local_to_global int32 incl. the intercept, n_global)."""
    rng = np.random.default_rng([seed, 3, pid])
    p = np.arange(1, LEVELS + 1, dtype=np.float64) ** -1.1
    cdf = np.cumsum(p / p.sum())
    beta = np.random.default_rng([seed, 2]).normal(0, 0.3, FIELDS * LEVELS)
    ng = FIELDS * LEVELS + 1
    lev = np.minimum(np.searchsorted(cdf, rng.random((rows, FIELDS))), LEVELS - 1).astype(np.int32)
    gid = lev + (np.arange(FIELDS, dtype=np.int32) * LEVELS)[None, :]
    logit = beta[gid].sum(axis=1) - 3.0
    y = np.where(rng.random(rows) < 1 / (1 + np.exp(-logit)), 1, -1).astype(np.int8)
    uniq, inv = np.unique(gid.reshape(-1), return_inverse=True)
    ci = np.sort(inv.reshape(rows, FIELDS).astype(np.int32), axis=1).reshape(-1)
    l2g = np.concatenate([uniq.astype(np.int32), [ng - 1]]).astype(np.int32)
    return np.arange(0, (rows + 1) * FIELDS, FIELDS, dtype=np.int64), ci, y, l2g, ng


def onehot_test_rows(rows: int, seed: int = SEED):
    """Held-out rows in GLOBAL ids: (row_ptr, global_idx, response 1/0)."""
    rp, ci, y, l2g, ng = onehot_partition(1_000_003, rows, seed)
    return rp, l2g[ci].astype(np.int32), np.where(y == 1, 1, 0).astype(np.int8), ng
