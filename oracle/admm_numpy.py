"""admm_numpy.py -- second, independent CPU ORACLE (test infrastructure, NOT product code).

NumPy/SciPy restatement of the same reference path as ``admm_oracle.c`` but written
separately, in a different language and with a different summation structure
(vectorised CSR mat-vecs, ``np.dot`` reductions), so that agreement between the two
(<=1e-12 relative before the float32 writes, tests/test_oracle.py) is evidence that
neither mis-states the reference.

PARITY UNPINNED: the reference has no tests/golden vectors for this path and cannot
run here (no JVM) -- see the header of admm_oracle.c.

Citations use the aliases of SURVEY.md (bw/ llf/ jobs/ models/ consumers/).
Only tests/ may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp


# ----------------------------------------------------------------------------- Tron
def euclidean_norm(v: np.ndarray) -> float:
    """bw/Tron.java:220-252 computes scale*sqrt(sum((v/scale)^2)) with a running scale;
    mathematically max|v| * sqrt(sum((v/max|v|)^2)); equal up to the last bits."""
    n = v.shape[0]
    if n < 1:
        return 0.0
    if n == 1:
        return float(abs(v[0]))
    m = float(np.max(np.abs(v)))
    if m == 0.0:
        return 0.0
    t = v / m
    return m * math.sqrt(float(np.dot(t, t)))


@dataclass
class TronStats:
    newton_iters: int = 0
    accepted: int = 0
    cg_iters: int = 0
    x_passes: int = 0
    f: float = 0.0
    gnorm: float = 0.0
    gnorm1: float = 0.0
    trace: List[Tuple[int, float, float, float, float, float, int]] = field(default_factory=list)


class LogisticL2:
    """llf/LogisticRegressionL2.java:72-248 on a scipy CSR matrix X (intercept column included)."""

    def __init__(self, X: sp.csr_matrix, y: np.ndarray, weight: np.ndarray, offset: np.ndarray,
                 prior_mean: np.ndarray, prior_var: np.ndarray, stats: TronStats):
        self.X = X
        self.XT = X.T.tocsr()
        self.y = y.astype(np.float64)
        self.weight = weight.astype(np.float64)
        self.offset = offset.astype(np.float64)
        self.pm = prior_mean
        self.pinv = 1.0 / prior_var                      # :107-109
        self.z = np.zeros(X.shape[0])
        self.D = np.zeros(X.shape[0])
        self.st = stats

    def fun(self, w: np.ndarray, count: bool = True) -> float:      # :156-193
        if count:
            self.st.x_passes += 1
        z = self.X @ w + self.offset
        self.z = z
        yz = self.y * z
        # two-branch stable form :174-178 == logaddexp(0, -yz)
        loss = np.where(yz >= 0, np.log1p(np.exp(-np.abs(yz))), -yz + np.log1p(np.exp(-np.abs(yz))))
        t = w - self.pm
        f = (2.0 * float(np.dot(self.weight, loss)) + float(np.dot(t * t, self.pinv))) / 2.0
        return f

    def grad(self, w: np.ndarray) -> np.ndarray:                    # :199-225
        self.st.x_passes += 1
        with np.errstate(over="ignore"):
            p = 1.0 / (1.0 + np.exp(-self.y * self.z))
        self.D = p * (1.0 - p)
        t = self.weight * (p - 1.0) * self.y
        return (w - self.pm) * self.pinv + self.XT @ t

    def Hv(self, s: np.ndarray) -> np.ndarray:                      # :231-248
        self.st.x_passes += 2
        wa = self.weight * self.D * (self.X @ s)
        return s * self.pinv + self.XT @ wa


def trcg(fo: LogisticL2, delta: float, g: np.ndarray) -> Tuple[np.ndarray, np.ndarray, int]:
    """bw/Tron.java:126-179"""
    s = np.zeros_like(g)
    r = -g
    d = r.copy()
    cgtol = 0.1 * euclidean_norm(g)
    cg_iter = 0
    rTr = float(np.dot(r, r))
    while True:
        if euclidean_norm(r) <= cgtol:
            break
        cg_iter += 1
        Hd = fo.Hv(d)
        alpha = rTr / float(np.dot(d, Hd))
        s = s + alpha * d
        if euclidean_norm(s) > delta:
            s = s + (-alpha) * d
            std = float(np.dot(s, d))
            sts = float(np.dot(s, s))
            dtd = float(np.dot(d, d))
            dsq = delta * delta
            rad = math.sqrt(std * std + dtd * (dsq - sts))
            if std >= 0:
                alpha = (dsq - sts) / (std + rad)
            else:
                alpha = (rad - std) / dtd
            s = s + alpha * d
            r = r + (-alpha) * Hd
            break
        r = r + (-alpha) * Hd
        rnew = float(np.dot(r, r))
        beta = rnew / rTr
        d = beta * d + r
        rTr = rnew
    return s, r, cg_iter


def tron(fo: LogisticL2, w: np.ndarray, eps: float, max_iter: int = 10000) -> np.ndarray:
    """bw/Tron.java:30-124"""
    eta0, eta1, eta2 = 1e-4, 0.25, 0.75
    sigma1, sigma2, sigma3 = 0.25, 0.5, 4.0
    st = fo.st
    zero = np.zeros_like(w)
    fo.fun(zero, count=False)
    g = fo.grad(zero)
    gnorm1 = euclidean_norm(g)
    f = fo.fun(w)
    g = fo.grad(w)
    delta = euclidean_norm(g)
    gnorm = delta
    search = not (gnorm <= eps * gnorm1)
    it = 1
    w = w.copy()
    while it <= max_iter and search:
        s, r, cg_iter = trcg(fo, delta, g)
        st.newton_iters += 1
        st.cg_iters += cg_iter
        w_new = w + s
        gs = float(np.dot(g, s))
        prered = -0.5 * (gs - float(np.dot(s, r)))
        fnew = fo.fun(w_new)
        actred = f - fnew
        snorm = euclidean_norm(s)
        if it == 1:
            delta = min(delta, snorm)
        if fnew - f - gs <= 0:
            alpha = sigma3
        else:
            alpha = max(sigma1, -0.5 * (gs / (fnew - f - gs)))
        if actred < eta0 * prered:
            delta = min(max(alpha, sigma1) * snorm, sigma2 * delta)
        elif actred < eta1 * prered:
            delta = max(sigma1 * delta, min(alpha * snorm, sigma2 * delta))
        elif actred < eta2 * prered:
            delta = max(sigma1 * delta, min(alpha * snorm, sigma3 * delta))
        else:
            delta = max(delta, min(alpha * snorm, sigma3 * delta))
        st.trace.append((it, actred, prered, delta, f, gnorm, cg_iter))
        if actred > eta0 * prered:
            it += 1
            w = w_new
            f = fnew
            g = fo.grad(w)
            st.accepted += 1
            gnorm = euclidean_norm(g)
            if gnorm <= eps * gnorm1:
                break
        if f < -1.0e32:
            break
        if abs(actred) <= 0 and prered <= 0:
            break
        if abs(actred) <= 1.0e-12 * abs(f) and abs(prered) <= 1.0e-12 * abs(f):
            break
    st.f, st.gnorm, st.gnorm1 = f, gnorm, gnorm1
    return w


# ----------------------------------------------------------------------------- partitions
def posterior_variance(X: sp.csr_matrix, y: np.ndarray, weight: np.ndarray, offset: np.ndarray, w: np.ndarray,
                       prior_var: np.ndarray, full: bool):
    """llf/LibLinear.java:314-337 on llf/LogisticRegressionL2.java:258-327: H = diag(1/priorVar) + X'DX with
    D_ii = weight_i p_i (1-p_i); full -> inverse(H) (dense solve instead of commons-math3's Cholesky), else 1/diag(H).
    X includes the intercept column."""
    score = X @ w + offset
    p = 1.0 / (1.0 + np.exp(-y * score))
    q = weight * p * (1 - p)
    if not full:
        return 1.0 / (1.0 / prior_var + np.asarray(X.multiply(X).T @ q).ravel()), None
    Xd = X.toarray()
    H = np.diag(1.0 / prior_var) + Xd.T @ (q[:, None] * Xd)
    V = np.linalg.inv(H)
    return np.diag(V).copy(), V


@dataclass
class Partition:
    X: sp.csr_matrix          # l x n_local (intercept column last, value 1.0)
    y: np.ndarray             # +1/-1
    weight: np.ndarray
    offset: np.ndarray
    l2g: np.ndarray           # local -> global index (intercept -> n_global-1)


def partition_from_csr(row_ptr, col_idx, val, y, weight, offset, n_local, l2g) -> Partition:
    """CSR without intercept (0-based local ids) -> scipy matrix with the intercept column appended
    (llf/LibLinearDataset.java:592-615). Duplicate (row, col) entries are summed by scipy, which is
    what Xv/XTv do with repeated FeatureNodes."""
    l = len(row_ptr) - 1
    data = np.ones(len(col_idx)) if val is None else np.asarray(val, dtype=np.float32).astype(np.float64)
    X = sp.csr_matrix((data, np.asarray(col_idx), np.asarray(row_ptr)), shape=(l, n_local - 1))
    X = sp.hstack([X, sp.csr_matrix(np.ones((l, 1)))], format="csr")
    X.sum_duplicates()
    yy = np.where(np.asarray(y) == 1, 1, -1)
    return Partition(X, yy, np.asarray(weight, dtype=np.float32).astype(np.float64),
                     np.asarray(offset, dtype=np.float32).astype(np.float64), np.asarray(l2g))


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def float_str_roundtrip(e: np.float32) -> float:
    """String.valueOf(float) -> Double.parseDouble (jobs/RegressionAdmmTrain.java:346,702; utils/Util.java:145-155)."""
    f = np.float32(e)
    s = np.format_float_scientific(f, unique=True, trim="0")
    if len(s.split("e")[0].replace(".", "").replace("-", "").rstrip("0")) <= 1:
        s = "%.1e" % float(f)          # Java prints >= 2 significant digits (1.4E-45, not 1E-45)
    return float(s)


class AdmmNumpy:
    """jobs/RegressionAdmmTrain.java:278-497 (L2 branch) over in-memory partitions."""

    def __init__(self, parts: Sequence[Partition], n_global: int, lambdas: Sequence[float],
                 rhos: Optional[Sequence[float]] = None, penalize_intercept: bool = False,
                 num_blocks: Optional[int] = None, regularizer: int = 2, lambda_map: Optional[np.ndarray] = None):
        self.parts = list(parts)
        self.N = num_blocks if num_blocks is not None else len(self.parts)
        self.ng = n_global
        lam = sorted(np.float32(x) for x in lambdas)                       # :636-638
        if rhos is None:
            rho = [np.float32(1.0) if l <= 100 else np.float32(10.0) for l in lam]   # :174-181
        else:
            m = {np.float32(l): np.float32(r) for l, r in zip(lambdas, rhos)}
            rho = [m[l] for l in lam]
        self.lam, self.rho = lam, rho
        self.pen = penalize_intercept
        self.reg = regularizer                                             # :143-147
        self.lambda_map = None if lambda_map is None else np.asarray(lambda_map, np.float32)   # per global feature, NaN = not listed
        nl = len(lam)
        self.Z = np.zeros((nl, n_global))
        self.u = np.zeros((len(self.parts), nl, n_global))                 # float32-valued doubles
        self.B = np.zeros_like(self.u)
        self.UPX = np.zeros_like(self.u)
        self.stats: Dict[Tuple[int, int], TronStats] = {}

    def solve(self, k: int, li: int, epsilon: float, rho_adapt_rate: float = 1.0) -> None:
        p = self.parts[k]
        zt = f32(self.Z[li])                                               # init-value file :330-331
        u = self.u[k, li]
        rho = float(self.rho[li])
        if np.float32(rho_adapt_rate) != np.float32(1.0):
            rho = rho * float(np.float32(rho_adapt_rate))                  # :652-658
        w0 = zt[p.l2g]
        pm = (zt - u)[p.l2g]                                               # :695-697
        pv = np.full(p.X.shape[1], 1.0 / rho)
        st = TronStats()
        fo = LogisticL2(p.X, p.y, p.weight, p.offset, pm, pv, st)
        pos = int(np.sum(p.y == 1))
        neg = p.X.shape[0] - pos
        w = tron(fo, w0, epsilon * min(pos, neg) / p.X.shape[0])           # llf/LibLinear.java:310-312
        beta = zt - u                                                      # absent features :373-383
        beta[p.l2g] = w
        self.B[k, li] = f32(beta)
        self.UPX[k, li] = f32(u + beta)                                    # :709-711
        self.stats[(k, li)] = st

    def mean_model_init(self, epsilon: float = 0.01, prior_mean: float = 0.0,
                        lambda_map: Optional[np.ndarray] = None) -> None:
        """initialize.boost.rate branch, jobs/RegressionAdmmTrain.java:236-276: one RegressionNaiveTrain reducer per
        (lambda, partition) (jobs/RegressionNaiveTrain.java:318-404) and z = meanModel (MeanLinearModelConsumer:61)."""
        b = 1.0 / self.N
        for li, l in enumerate(self.lam):
            zbar = np.zeros(self.ng)
            for k, p in enumerate(self.parts):
                n = p.X.shape[1]
                pv = np.full(n, 1.0 / float(l))                            # `1.0 / lambda`, :380
                if lambda_map is not None:                                 # 1/lambda.map[k], :311-316
                    lm = np.asarray(lambda_map, np.float32)[p.l2g].astype(np.float64)
                    pv = np.where(np.isnan(lm), pv, 1.0 / lm)
                if not self.pen:
                    pv[p.l2g == self.ng - 1] = 100000.0                    # :317-320
                st = TronStats()
                fo = LogisticL2(p.X, p.y, p.weight, p.offset, np.full(n, float(prior_mean)), pv, st)
                pos = int(np.sum(p.y == 1))
                neg = p.X.shape[0] - pos
                w = tron(fo, np.zeros(n), epsilon * min(pos, neg) / p.X.shape[0])
                model = np.zeros(self.ng)
                model[p.l2g] = f32(w)                                      # part file float32, LinearModel.java:703,716
                zbar = zbar + b * model
                self.stats[(k, li)] = st
            self.Z[li] = zbar
        self.u[:] = 0.0

    def iterate(self, epsilon: float, rho_adapt_rate: float = 1.0) -> Tuple[float, float]:
        nl = len(self.lam)
        for k in range(len(self.parts)):
            for li in range(nl):
                self.solve(k, li, epsilon, rho_adapt_rate)
        b = 1.0 / self.N
        mindiff, maxdiff = 99999999.0, 0.0
        for li in range(nl):
            xbar = np.zeros(self.ng)
            ubar = np.zeros(self.ng)
            for k in range(len(self.parts)):                               # consumers/MeanLinearModelConsumer.java:61
                xbar = xbar + b * self.B[k, li]
                ubar = ubar + b * self.u[k, li]
            l, r = self.lam[li], self.rho[li]
            nr = np.float32(self.N) * r                                    # int * float in float
            if self.reg == 2:
                weight = float(nr / (l + nr))                              # float arithmetic :381
                wvec = np.full(self.ng, weight)
                if self.lambda_map is not None:                            # weightmap :383-386: float sum, then + 0.0 in double
                    lm = self.lambda_map
                    listed = ~np.isnan(lm)
                    wvec[listed] = float(nr) / ((lm[listed] + nr).astype(np.float32).astype(np.float64) + 0.0)
                # thisz = 0; linearCombine(1, weight, xbar, weightmap); linearCombine(1, weight, ubar, weightmap) :387-391;
                # the intercept always takes the scalar weight (models/LinearModel.java:205)
                zn = (1.0 * np.zeros(self.ng) + wvec * xbar)
                zn = 1.0 * zn + wvec * ubar
                zn[-1] = (0.0 + weight * xbar[-1]) + weight * ubar[-1]
            else:                                                          # L1 :406-451
                weight = float(l) / (float(np.float32(r * np.float32(self.N))) + 0.0)
                zn = (np.zeros(self.ng) + 1.0 * xbar) + 1.0 * ubar
                coef = zn[:-1]
                coef[...] = np.where(coef > weight, coef - weight, np.where(coef < -weight, coef + weight, coef))  # :424-436 (the band is kept)
            if not self.pen:
                zn[-1] = xbar[-1] + ubar[-1]                               # :392-403, :438-449
            diff = float(np.max(np.abs(self.Z[li] - zn)))                  # :463-464
            self.Z[li] = zn
            mindiff = min(mindiff, diff)
            maxdiff = max(maxdiff, diff)
        for li in range(nl):
            self.u[:, li, :] = f32(self.UPX[:, li, :] - self.Z[li][None, :])   # computeU :752-757
        return maxdiff, mindiff

    def run(self, niter: int, epsilon_stop: float = 1e-4, aggressive: bool = False):
        mindiff = 99999999.0
        e = np.float32(0.01)                                               # :279
        hist = []
        for i in range(1, niter + 1):
            if i > 1 and mindiff < 0.001 and not aggressive:
                e = np.float32(e / np.float32(10))                         # :338-341
            elif aggressive and i > 5:
                e = np.float32(e / np.float32(10))
            eps = float_str_roundtrip(e)
            maxdiff, mindiff = self.iterate(eps)
            hist.append((eps, maxdiff, mindiff))
            if maxdiff < epsilon_stop and float(e) <= 0.00001:             # :493-496
                break
        return hist
