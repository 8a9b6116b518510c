"""Test-side adaptor: the CPU oracle behind the engine protocol of mlease_amd.admm.AdmmTrain.

Lives in tests/ on purpose: the product package never routes through the oracle.
"""
import numpy as np
import torch

import oracle_lib as ol


class _Fin:
    def __init__(self, mx, mn):
        self.maxdiff, self.mindiff = mx, mn


class OracleEngine:
    def __init__(self, blocks, n_global, lambdas, rhos, num_blocks, penalize_intercept=False, nthreads=2,
                 regularizer=2, lambda_map=None):
        self.o = ol.OracleAdmm(blocks, n_global, lambdas, rhos, num_blocks=num_blocks,
                               penalize_intercept=penalize_intercept, regularizer=regularizer, lambda_map=lambda_map)
        self.nthreads = nthreads
        n = self.o.nl * self.o.ng
        self._buf = np.zeros(2 * n)

    def solve_local(self, eps, rate=1.0):
        self.o.solve_local(eps, rate, self.nthreads)
        xb, ub = self.o.partial_means()
        n = len(xb)
        self._buf[:n] = xb
        self._buf[n:] = ub
        return None

    def naive_solve_local(self, eps, prior_mean=0.0):
        self.o.naive_solve_local(eps, prior_mean, self.nthreads)
        xb, ub = self.o.partial_means()
        n = len(xb)
        self._buf[:n] = xb
        self._buf[n:] = ub

    def naive_finish(self):
        xb, ub = self.o.partial_means()
        n = len(xb)
        xb[:] = self._buf[:n]
        ub[:] = self._buf[n:]
        self.o.naive_finish()

    def consensus_tensor(self):
        return torch.from_numpy(self._buf)

    def consensus_finish(self):
        xb, ub = self.o.partial_means()
        n = len(xb)
        xb[:] = self._buf[:n]
        ub[:] = self._buf[n:]
        return _Fin(*self.o.finish())

    def z(self):
        return self.o.z()

    def set_test_data(self, row_ptr, global_idx, val, response, weight=None, offset=None):
        self._test = (row_ptr, global_idx, val, response, weight, offset)

    def test_loglik_sums(self):
        Z, _ = self.o.z()
        return np.array([ol.test_loglik_sum(Z[li], *self._test) for li in range(Z.shape[0])])


class OracleScorer:
    """CPU stand-in of mlease_amd.hip_engine.HipScorer for the host-logic tests."""

    @staticmethod
    def score_rows(model32, row_ptr, global_idx, val, offset=None):
        return ol.score_rows(model32, row_ptr, global_idx, val, offset)
