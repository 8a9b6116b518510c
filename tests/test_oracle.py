"""CPU tests that pin the oracle as far as it can be pinned without a JVM.

The reference has no tests/goldens for this path ("parity unpinned"), so the C oracle is
held by: an independent NumPy restatement (<=1e-12 before the float32 writes), committed oracle
outputs (regression), and mathematical properties that depend on neither implementation.
"""
import os

import numpy as np
import pytest
import scipy.optimize as so

import admm_numpy as an
import oracle_lib as ol
import mlease_amd  # noqa: F401
from mlease_amd import dataset
from fixtures import load_c1, load_c1_golden, synth_sparse


@pytest.fixture(scope="module")
def c1():
    return load_c1()


@pytest.fixture(scope="module")
def gold():
    return load_c1_golden()


def np_parts(pd):
    return [an.partition_from_csr(b.row_ptr, b.col_idx, b.val, b.y, b.weight, b.offset, b.n_local,
                                  b.local_to_global) for b in pd.blocks]


def test_c1_fixture_shape(c1):
    # 1000 records, 200 features, 100 326 nnz, 299 positives
    assert c1.n_global == 201 and len(c1.blocks) == 8
    assert sum(b.l for b in c1.blocks) == 1000
    assert sum(b.nnz for b in c1.blocks) == 100326
    assert sum(int(np.sum(b.y == 1)) for b in c1.blocks) == 299
    for b in c1.blocks:
        assert b.local_to_global[-1] == c1.n_global - 1
        for i in range(b.l):
            seg = b.col_idx[b.row_ptr[i]:b.row_ptr[i + 1]]
            assert np.all(np.diff(seg) > 0)          # sorted by local id, llf/LibLinearDataset.java:481-482


def test_oracle_matches_committed_golden(c1, gold):
    oc = ol.OracleAdmm(c1.blocks, c1.n_global, [1.0], [1.0])
    done, diffs, eps = oc.run(20, nthreads=4)
    assert done == 20
    assert np.array_equal(diffs, gold["diffs"]) and np.array_equal(eps, gold["eps"])
    assert np.array_equal(oc.z()[0], gold["Z"][-1])
    for k in range(8):
        b, upx, un = oc.partition_model(k, 0)
        assert np.array_equal(b, gold["B_it20"][k, 0])
        assert np.array_equal(upx, gold["UPX_it20"][k, 0])
        assert np.array_equal(un, gold["Unext_it20"][k, 0])
    cnt = np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in oc.stats()])
    assert np.array_equal(cnt, gold["counters"][-1])


def test_c_vs_numpy_single_solve_1e12(c1):
    """Same solve, two implementations with different summation structure: raw doubles agree <=1e-12."""
    rng = np.random.default_rng(7)
    for k in (0, 3, 7):
        b = c1.blocks[k]
        d = ol.OracleDataset.from_block(b)
        init = rng.normal(0, 0.1, b.n_local)
        pm = rng.normal(0, 0.1, b.n_local)
        pv = np.full(b.n_local, 1.0 / 1.0)
        wc, st = d.train(init, pm, pv, 0.01)
        p = np_parts(type("X", (), {"blocks": [b]}))[0]
        stn = an.TronStats()
        fo = an.LogisticL2(p.X, p.y, p.weight, p.offset, pm, pv, stn)
        pos = int(np.sum(p.y == 1))
        wn = an.tron(fo, init, 0.01 * min(pos, b.l - pos) / b.l)
        assert (st.newton_iters, st.cg_iters, st.x_passes) == (stn.newton_iters, stn.cg_iters, stn.x_passes)
        assert np.max(np.abs(wc - wn)) <= 1e-12 * max(1.0, np.max(np.abs(wc)))


def test_c_vs_numpy_admm_float32_identical(c1, gold):
    na = an.AdmmNumpy(np_parts(c1), c1.n_global, [1.0])
    na.run(5)
    assert np.array_equal(na.Z.astype(np.float32), gold["Z"][4].astype(np.float32))


def test_multilambda_golden_and_sorting(c1, gold):
    # lambdas given unsorted: key % n_lambda indexes the ASCENDING list (jobs/RegressionAdmmTrain.java:636-638,648)
    lam = [100.0, 1.0, 1000.0, 10.0]
    rho = [1.0, 1.0, 10.0, 1.0]
    oc = ol.OracleAdmm(c1.blocks, c1.n_global, lam, rho)
    assert list(oc.lambdas) == [1.0, 10.0, 100.0, 1000.0] and list(oc.rhos) == [1.0, 1.0, 1.0, 10.0]
    done, diffs, eps = oc.run(6, nthreads=4)
    assert np.array_equal(oc.z()[0], gold["Zm"][-1])
    assert np.array_equal(diffs, gold["diffsm"])


def test_finite_difference_grad_and_hv(c1):
    b = c1.blocks[1]
    d = ol.OracleDataset.from_block(b)
    rng = np.random.default_rng(3)
    n = b.n_local
    w = rng.normal(0, 0.2, n)
    pm = rng.normal(0, 0.2, n)
    pv = rng.uniform(0.5, 2.0, n)
    s = rng.normal(0, 1, n)
    f, g, Hs = d.eval(w, pm, pv, s)
    h = 1e-6
    fp, gp, _ = d.eval(w + h * s, pm, pv, s)
    fm, gm, _ = d.eval(w - h * s, pm, pv, s)
    assert abs((fp - fm) / (2 * h) - g @ s) <= 1e-6 * max(1.0, abs(g @ s))
    assert np.max(np.abs((gp - gm) / (2 * h) - Hs)) <= 1e-5 * max(1.0, np.max(np.abs(Hs)))


def test_kkt_at_exit_and_eps_scaling(c1):
    """At exit ||grad|| <= eps_tron * ||grad(0)|| with eps_tron = epsilon*min(pos,neg)/l (llf/LibLinear.java:311)."""
    for k in range(8):
        b = c1.blocks[k]
        d = ol.OracleDataset.from_block(b)
        n = b.n_local
        pm = np.zeros(n)
        pv = np.ones(n)
        w, st = d.train(np.zeros(n), pm, pv, 0.01)
        pos = int(np.sum(b.y == 1))
        eps_tron = 0.01 * min(pos, b.l - pos) / b.l
        f, g, _ = d.eval(w, pm, pv)
        _, g0, _ = d.eval(np.zeros(n), pm, pv)
        assert np.linalg.norm(g) <= eps_tron * np.linalg.norm(g0) * (1 + 1e-12)
        assert abs(st.gnorm1 - np.linalg.norm(g0)) <= 1e-12 * st.gnorm1
        assert st.x_passes == 3 + 2 * st.cg_iters + st.newton_iters + st.accepted      # SURVEY 8d pass formula


def test_tight_solve_matches_scipy_optimum(c1):
    b = c1.blocks[2]
    d = ol.OracleDataset.from_block(b)
    n = b.n_local
    pm = np.full(n, 0.05)
    pv = np.full(n, 0.5)
    w, _ = d.train(np.zeros(n), pm, pv, 1e-10)
    res = so.minimize(lambda v: d.eval(v, pm, pv)[0], np.zeros(n), jac=lambda v: d.eval(v, pm, pv)[1],
                      method="L-BFGS-B", options={"maxiter": 5000, "ftol": 1e-15, "gtol": 1e-10})
    assert np.max(np.abs(res.x - w)) < 1e-5


def test_admm_approaches_centralised_optimum(c1):
    """ADMM iterates approach argmin sum loss + (lambda/2)||beta_{-0}||^2 (unpenalised intercept); slow but monotone."""
    rows = []
    for b in c1.blocks:
        import scipy.sparse as sp
        X = sp.csr_matrix((b.val.astype(np.float64), b.col_idx, b.row_ptr), shape=(b.l, b.n_local - 1))
        P = sp.csr_matrix((np.ones(b.n_local - 1), (np.arange(b.n_local - 1), b.local_to_global[:-1])),
                          shape=(b.n_local - 1, c1.n_global - 1))
        rows.append((X @ P, b.y.astype(np.float64)))
    import scipy.sparse as sp
    X = sp.vstack([r[0] for r in rows]).tocsr()
    y = np.concatenate([r[1] for r in rows])

    def obj(v):
        zz = X @ v[:-1] + v[-1]
        f = np.sum(np.logaddexp(0, -y * zz)) + 0.5 * 1.0 * np.dot(v[:-1], v[:-1])
        p = 1 / (1 + np.exp(y * zz))
        g = np.concatenate([X.T @ (-y * p) + v[:-1], [np.sum(-y * p)]])
        return f, g

    wstar = so.minimize(obj, np.zeros(c1.n_global), jac=True, method="L-BFGS-B",
                        options={"maxiter": 10000, "ftol": 1e-15, "gtol": 1e-9}).x
    oc = ol.OracleAdmm(c1.blocks, c1.n_global, [1.0], [1.0])
    dist = []
    e = np.float32(0.01)
    for i in range(1, 51):
        oc.iterate(ol.float_to_string_to_double(e), 1.0, nthreads=4)
        if i in (10, 20, 35, 50):
            dist.append(np.linalg.norm(oc.z()[0][0] - wstar) / np.linalg.norm(wstar))
    assert all(a > b for a, b in zip(dist, dist[1:])) and dist[-1] < 0.06


def test_eps_schedule_float_string_roundtrip():
    # 0.01f -> "0.01" -> 0.01 ; 0.01f/10 -> "9.999999E-4" -> 9.999999e-4
    e = np.float32(0.01)
    assert ol.float_to_string_to_double(e) == 0.01
    e = np.float32(e / np.float32(10))
    assert ol.float_to_string_to_double(e) == 9.999999e-4
    assert an.float_str_roundtrip(e) == 9.999999e-4
    for _ in range(60):                  # no floor: decays through float32 denormals to exactly 0
        e = np.float32(e / np.float32(10))
        assert ol.float_to_string_to_double(e) == an.float_str_roundtrip(e)
    assert float(e) == 0.0


def test_absent_features_and_partition_local_space():
    """Features absent from a partition are not optimised: beta_k[j] = z[j]-u_k[j] (llf/LibLinear.java:373-383)."""
    pd = synth_sparse(11, 600, 300, 4, 6)
    assert any(b.n_local < pd.n_global for b in pd.blocks)
    oc = ol.OracleAdmm(pd.blocks, pd.n_global, [1.0], [1.0])
    oc.iterate(0.01)
    Z1, z1_32 = oc.z()
    _, _, u2 = oc.partition_model(0, 0)
    oc.iterate(0.01)
    b2, upx2, _ = oc.partition_model(0, 0)
    absent = np.setdiff1d(np.arange(pd.n_global), pd.blocks[0].local_to_global)
    assert len(absent) > 0
    expect = (z1_32[0].astype(np.float64) - u2.astype(np.float64))
    assert np.array_equal(b2[absent], expect[absent].astype(np.float32))
    assert np.array_equal(upx2[absent], (u2.astype(np.float64) + expect)[absent].astype(np.float32))
    # and the numpy restatement agrees on such data, multi-lambda
    na = an.AdmmNumpy(np_parts(pd), pd.n_global, [0.5, 20.0])
    oc2 = ol.OracleAdmm(pd.blocks, pd.n_global, [0.5, 20.0], [1.0, 1.0])
    na.run(4)
    oc2.run(4)
    assert np.array_equal(na.Z.astype(np.float32), oc2.z()[1])


def test_binary_weighted_offset_variants_agree():
    for binary in (False, True):
        pd = synth_sparse(5 + binary, 500, 40, 5, 4, binary=binary, weights=True, offsets=True)
        na = an.AdmmNumpy(np_parts(pd), pd.n_global, [2.0])
        oc = ol.OracleAdmm(pd.blocks, pd.n_global, [2.0], [1.0])
        na.run(4)
        oc.run(4)
        assert np.array_equal(na.Z.astype(np.float32), oc.z()[1])


def test_sharded_partial_means_equal_single_instance(c1):
    """Two shards (partitions 0,2,4,6 / 1,3,5,7) + summed partial means == one instance, to <=1 ulp(double)."""
    full = ol.OracleAdmm(c1.blocks, c1.n_global, [1.0], [1.0])
    sh = [ol.OracleAdmm(c1.blocks[r::2], c1.n_global, [1.0], [1.0], num_blocks=8) for r in range(2)]
    for it in range(3):
        full.iterate(0.01)
        for s in sh:
            s.solve_local(0.01)
        xs = sum(s.partial_means()[0].copy() for s in sh)
        us = sum(s.partial_means()[1].copy() for s in sh)
        for s in sh:
            xb, ub = s.partial_means()
            xb[:] = xs
            ub[:] = us
            s.finish()
        assert np.array_equal(sh[0].z()[1], full.z()[1])
        assert np.array_equal(sh[0].z()[0], sh[1].z()[0])


def test_reference_algorithm_is_order_sensitive_on_onehot_data():
    """On rare-feature one-hot data (BASELINE configs[2] shape) the reference algorithm itself is not reproducible
    to 1e-5 under a mere row permutation: boundary-constrained TRON steps amplify last-bit differences ~100x per
    Newton iteration. Hadoop does not fix the row order inside a reducer key, so this is the
    reference's own spread; the GPU parity test on such data is stated relative to it."""
    from fixtures import onehot_blocks, permute_rows
    pd = onehot_blocks(80000, 2)
    b = pd.blocks[0]
    n = b.n_local
    z = np.zeros(n)
    one = np.ones(n)
    w_a, st_a = ol.OracleDataset.from_block(b).train(z, z, one, 0.01)
    w_b, st_b = ol.OracleDataset.from_block(permute_rows(b)).train(z, z, one, 0.01)
    spread = np.max(np.abs(w_a - w_b)) / np.max(np.abs(w_a))
    assert 1e-5 < spread < 1e-2
    # ... while the first iterations agree to rounding and both orders reach the same optimum when solved tightly
    w_c, _ = ol.OracleDataset.from_block(b).train(z, z, one, 0.2)
    w_d, _ = ol.OracleDataset.from_block(permute_rows(b)).train(z, z, one, 0.2)
    assert np.max(np.abs(w_c - w_d)) < 1e-9
    w_e, _ = ol.OracleDataset.from_block(b).train(z, z, one, 1e-9)
    w_f, _ = ol.OracleDataset.from_block(permute_rows(b)).train(z, z, one, 1e-9)
    assert np.max(np.abs(w_e - w_f)) < 1e-6


def test_mean_model_warm_start_c_vs_numpy_and_seam(c1):
    """initialize.boost.rate (jobs/RegressionAdmmTrain.java:236-276): the C restatement of the NaiveTrain + meanModel
    step equals (a) its own LibLinear.train seam composed by hand and (b) the independent numpy restatement."""
    pd = synth_sparse(5, 900, 350, 5, 4, weights=True)
    assert any(b.n_local < pd.n_global for b in pd.blocks)
    lm = np.full(pd.n_global, np.nan, np.float32)
    lm[::9] = 12.0
    lam = [0.5, 8.0]
    for kw, pmean in ((dict(), 0.0), (dict(lambda_map=lm), 0.0), (dict(penalize_intercept=True), 0.25)):
        oc = ol.OracleAdmm(pd.blocks, pd.n_global, lam, [1.0, 1.0], **kw)
        oc.naive_solve_local(0.01, pmean, nthreads=2)
        oc.naive_finish()
        Z = oc.z()[0]
        # (a) composition through the S2 seam
        for li, l in enumerate(lam):
            want = np.zeros(pd.n_global)
            for b in pd.blocks:
                pv = np.full(b.n_local, 1.0 / float(np.float32(l)))
                if "lambda_map" in kw:
                    m = lm[b.local_to_global].astype(np.float64)
                    pv = np.where(np.isnan(m), pv, 1.0 / m)
                if not kw.get("penalize_intercept"):
                    pv[-1] = 100000.0
                w, _ = ol.OracleDataset.from_block(b).train(np.zeros(b.n_local), np.full(b.n_local, pmean), pv, 0.01)
                model = np.zeros(pd.n_global)
                model[b.local_to_global] = w.astype(np.float32)
                want = want + (1.0 / len(pd.blocks)) * model
            assert np.array_equal(Z[li], want)
        # (b) numpy restatement: same float32 mean model
        na = an.AdmmNumpy(np_parts(pd), pd.n_global, lam, penalize_intercept=bool(kw.get("penalize_intercept")))
        na.mean_model_init(0.01, pmean, kw.get("lambda_map"))
        assert np.array_equal(na.Z.astype(np.float32), Z.astype(np.float32))
        assert np.max(np.abs(na.Z - Z)) <= 1e-12 * np.max(np.abs(Z))
    # the warm start is a sensible start: mean of the per-partition fits is far closer to the consensus than 0
    oc0 = ol.OracleAdmm(pd.blocks, pd.n_global, lam, [1.0, 1.0])
    oc0.run(30)
    zfin = oc0.z()[0]
    oc = ol.OracleAdmm(pd.blocks, pd.n_global, lam, [1.0, 1.0])
    oc.naive_solve_local(0.01, 0.0)
    oc.naive_finish()
    assert np.linalg.norm(oc.z()[0] - zfin) < 0.8 * np.linalg.norm(zfin)


def test_posterior_variance_c_vs_numpy(c1):
    """LibLinear.train's computePosteriorVar tail (llf/LibLinear.java:314-337; hessian / hessianDiagonal
    llf/LogisticRegressionL2.java:258-327; commons-math3 3.2 CholeskyDecomposition + inverse restated): the C oracle vs
    the numpy restatement (dense inverse), on valued and binary rows."""
    rng = np.random.default_rng(3)
    for pd_, binary in ((c1, False), (synth_sparse(8, 500, 60, 5, 2, binary=True, weights=True, offsets=True), True)):
        b = pd_.blocks[0]
        p = np_parts(pd_)[0]
        od = ol.OracleDataset.from_block(b)
        pv = rng.uniform(0.3, 3.0, b.n_local)
        w, _ = od.train(np.zeros(b.n_local), np.zeros(b.n_local), pv, 1e-3)
        dv, _, _ = od.posterior_variance(w, pv, False)
        dn, _ = an.posterior_variance(p.X, p.y, p.weight, p.offset, w, pv, False)
        assert np.max(np.abs(dv - dn) / dn) < 1e-13
        fv, V, H = od.posterior_variance(w, pv, True)
        fn, Vn = an.posterior_variance(p.X, p.y, p.weight, p.offset, w, pv, True)
        assert np.array_equal(H, H.T) and np.all(np.diag(H) > 1.0 / pv - 1e-15)
        assert np.max(np.abs(V - Vn)) <= 1e-10 * np.max(np.abs(Vn))
        assert np.array_equal(fv, np.diag(V)) and np.max(np.abs(V @ H - np.eye(b.n_local))) < 1e-9
        assert np.all(fv >= dv * (1 - 1e-12))        # (H^-1)_kk >= 1/H_kk for SPD H


def test_l1_and_lambda_map_c_vs_numpy(c1):
    """The remaining consensus branches (jobs/RegressionAdmmTrain.java:383-386 lambda.map weights, :406-451 L1 iterative
    thresholding -- including its untouched band |z| <= weight) in the C oracle and in the independent numpy restatement."""
    lm = np.full(c1.n_global, np.nan, np.float32)
    lm[::6] = 35.0
    lm[4] = 0.25
    lm[-1] = 7.0                                   # an intercept entry in the map is ignored by linearCombine
    for kw in (dict(lambda_map=lm), dict(regularizer=1), dict(regularizer=1, penalize_intercept=True),
               dict(lambda_map=lm, penalize_intercept=True)):
        oc = ol.OracleAdmm(c1.blocks, c1.n_global, [0.5, 20.0], [1.0, 1.0], **kw)
        na = an.AdmmNumpy(np_parts(c1), c1.n_global, [0.5, 20.0], [1.0, 1.0], **kw)
        for it in range(4):
            mo = oc.iterate(0.01, 1.0, nthreads=4)
            mn = na.iterate(0.01)
            assert np.array_equal(na.Z.astype(np.float32), oc.z()[1]), (list(kw), it)
            assert abs(mo[0] - mn[0]) <= 1e-10 * mo[0] and abs(mo[1] - mn[1]) <= 1e-10 * mo[1]
    # L1: coefficients inside the band are kept, not zeroed (:424-436)
    oc = ol.OracleAdmm(c1.blocks, c1.n_global, [400.0], [1.0], regularizer=1)
    oc.iterate(0.01, 1.0, nthreads=4)
    z = oc.z()[0][0][:-1]
    w = 400.0 / 8.0
    assert np.any((np.abs(z) > 0) & (np.abs(z) <= w))


def test_benchmark_data_generators_agree_bit_for_bit():
    """tools/synth_data.py: the NumPy generator, its torch twin (what bench.py runs on the GPU; here on the CPU device) and the
    C twin behind tests/golden/make_ref_loglik.py (oracle/synth.c) build the dense benchmark rows from integer arithmetic only and
    must produce identical float32 values and labels -- the committed oracle log-likelihood of BASELINE configs[1]
    (tests/golden/c2_ref_loglik.json) is only meaningful for bench.py if they do."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_data as sd
    for row0, rows, nf, stride in ((0, 257, 1000, 1), (5, 300, 1000, 64), (1_000_000, 100, 333, 1)):
        X, y = sd.dense_rows_np(row0, rows, nf, stride=stride)
        Xc, yc = ol.synth_dense(row0, rows, nf, sd.dense_beta(nf), sd.SEED, stride=stride)
        Xt, yt = sd.dense_rows_torch(torch, "cpu", row0, rows, nf, stride=stride)
        assert np.array_equal(X, Xc) and np.array_equal(y, yc)
        assert np.array_equal(X, Xt.numpy()) and np.array_equal(y, yt.numpy())
    x = sd.dense_rows_np(0, 2000, 500)[0].astype(np.float64).ravel()
    assert abs(x.mean()) < 5e-3 and abs(x.std() - 1.0) < 5e-3 and np.max(np.abs(x)) <= 6.0


def test_reference_loglik_golden_is_consistent():
    """tests/golden/c2_ref_loglik.json (tests/golden/make_ref_loglik.py): 20 iterations, epsilon schedule of the driver loop, and a
    spot check that the generator still produces the job it was computed on (first rows of partition 0)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_data as sd
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_ref_loglik.json")))
    assert g["iterations"] == 20 and len(g["loglik_by_iteration"]) == 20 and g["ref_loglik"] == g["loglik_by_iteration"][-1]
    assert g["epsilon_by_iteration"][0] == 0.01 and g["epsilon_by_iteration"][3] == ol.float_to_string_to_double(np.float32(0.01) / np.float32(10))
    assert all(-0.36 < v < -0.33 for v in g["loglik_by_iteration"]) and "seed %d" % sd.SEED in g["generator"]


def test_portable_math_and_the_verification_twin():
    """ml-ease_amd/csrc/portable_math.h through the oracle twin liboracle_pm.so: same restatement, exp/log1p from +,-,*,/ only.
    On config #1 the twin follows the plain oracle's trajectory and agrees to 1e-12 on the double z (the functions differ
    from libm by <= 1 ulp); on the solve level it is the bit-exact partner of the HIP library's MLX_FAITHFUL mode (-m gpu)."""
    from fixtures import load_c1
    c1 = load_c1()
    a = ol.OracleAdmm(c1.blocks, c1.n_global, [1.0], [1.0])
    b = ol.OracleAdmm(c1.blocks, c1.n_global, [1.0], [1.0], pm=True)
    for _ in range(3):
        a.iterate(0.01, 1.0, nthreads=2)
        b.iterate(0.01, 1.0, nthreads=2)
        ca = [(s.newton_iters, s.cg_iters) for s in a.stats()]
        cb = [(s.newton_iters, s.cg_iters) for s in b.stats()]
        assert ca == cb
        assert np.max(np.abs(a.z()[0] - b.z()[0])) <= 1e-12 * np.max(np.abs(a.z()[0]))


def test_verification_twin_keeps_its_math_without_openmp(tmp_path):
    """The -DORC_PORTABLE_MATH block must not depend on -fopenmp (round-2 finding: it sat inside `#ifdef _OPENMP`, so a twin
    built without OpenMP silently evaluated libm). Build both variants without OpenMP and ask them."""
    import ctypes
    import subprocess
    odir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    for flag, want in (([], b"libm"), (["-DORC_PORTABLE_MATH"], b"portable")):
        so = str(tmp_path / ("o_%s.so" % want.decode()))
        subprocess.check_call(["gcc", "-O0", "-fPIC", "-std=c11", "-ffp-contract=off", "-shared", "-I" + os.path.join(odir, "..", "ml-ease_amd", "csrc"),
                               *flag, "-o", so, os.path.join(odir, "admm_oracle.c"), os.path.join(odir, "synth.c"), "-lm"])
        L = ctypes.CDLL(so)
        L.orc_math_kind.restype = ctypes.c_char_p
        assert L.orc_math_kind() == want
    assert ol.lib(False).orc_math_kind() == b"libm" and ol.lib(True).orc_math_kind() == b"portable"


def test_oracle_tron_follows_c_liblinear_through_scikit_learn():
    """An independent pin of the TRON core (R3-R9 of SURVEY 8a), short of running the Java: scikit-learn's `liblinear` solver IS the C++
    liblinear whose Java port the reference vendors (de.bwaldvogel.liblinear), primal L2-regularised logistic regression (solver 0).
    With prior mean 0, prior variance 1 (= C 1), the bias feature appended last with value 1 and penalised like the rest
    (intercept_scaling 1) the two objectives are the same function, so the whole trust-region trajectory must coincide: equal Newton
    iteration counts and coefficients to ~1e-15 at EVERY tolerance -- including 0.01, where the result is whatever the trajectory has
    reached. Also with per-row weights (scikit-learn's liblinear multiplies the loss by sample_weight, as llf/LogisticRegressionL2.java
    :166-182 does by weight[i]) and on binary one-hot rows; and at another C the minimisers agree at a tight tolerance."""
    sk = pytest.importorskip("sklearn.linear_model")
    import scipy.sparse as sps
    from fixtures import load_c1, onehot_blocks
    c1 = load_c1()
    rng = np.random.default_rng(3)
    # (tolerance per case: on the sample data the trajectories agree to the last bits; on one-hot rows the C library's unrolled BLAS
    # dot products -- another summation order than the Java port's plain loops -- are amplified to ~1e-6 at the loose tolerance,
    # the order sensitivity DESIGN section 5 is about, here between liblinear's own two language versions)
    cases = [(c1.blocks[0], None, 1e-12), (c1.blocks[3], None, 1e-12),
             (c1.blocks[5], rng.uniform(0.5, 2.0, c1.blocks[5].l).astype(np.float32), 1e-12),
             (onehot_blocks(3000, 1, levels=40).blocks[0], None, 1e-5)]
    for b, wt, rtol in cases:
        nf = b.n_local - 1
        vals = np.ones(len(b.col_idx)) if b.val is None else b.val.astype(np.float64)
        X = sps.csr_matrix((vals, b.col_idx, b.row_ptr), shape=(b.l, nf))
        od = ol.OracleDataset(b.l, b.n_local, b.row_ptr, b.col_idx, b.val, b.y, b.weight if wt is None else wt, b.offset)
        for tol in (1e-2, 1e-4, 1e-8):
            clf = sk.LogisticRegression(penalty="l2", C=1.0, solver="liblinear", tol=tol, fit_intercept=True, intercept_scaling=1.0, max_iter=10000)
            clf.fit(X, b.y, sample_weight=None if wt is None else wt.astype(np.float64))
            w_sk = np.concatenate([clf.coef_[0], clf.intercept_])
            w, st = od.train(np.zeros(b.n_local), np.zeros(b.n_local), np.ones(b.n_local), tol)
            assert int(clf.n_iter_[0]) == st.newton_iters == st.accepted, (tol, clf.n_iter_, st.newton_iters, st.accepted)
            assert np.max(np.abs(w - w_sk)) <= rtol * max(1.0, np.max(np.abs(w))), (tol, float(np.max(np.abs(w - w_sk))))
    # another regularisation strength: same minimiser (the objectives differ by the factor C, so the trajectories do)
    b = c1.blocks[1]
    X = sps.csr_matrix((b.val.astype(np.float64), b.col_idx, b.row_ptr), shape=(b.l, b.n_local - 1))
    clf = sk.LogisticRegression(penalty="l2", C=0.25, solver="liblinear", tol=1e-10, fit_intercept=True, intercept_scaling=1.0, max_iter=10000).fit(X, b.y)
    w, _ = ol.OracleDataset.from_block(b).train(np.zeros(b.n_local), np.zeros(b.n_local), np.full(b.n_local, 0.25), 1e-10)
    assert np.max(np.abs(w - np.concatenate([clf.coef_[0], clf.intercept_]))) <= 1e-7


def _closed_form(b, w, pm, pv, s=None):
    """F, grad F, (hess F) s of the local objective (Appendix A5 of SURVEY.md) as three lines of scipy.sparse algebra -- shares no
    loop, no order of operations and no code with oracle/admm_oracle.c or oracle/admm_numpy.py."""
    import scipy.sparse as sps
    from scipy.special import expit, log1p
    nf = b.n_local - 1
    vals = np.ones(len(b.col_idx)) if b.val is None else b.val.astype(np.float64)
    X = sps.hstack([sps.csr_matrix((vals, b.col_idx, b.row_ptr), shape=(b.l, nf)), sps.csr_matrix(np.ones((b.l, 1)))]).tocsr()
    y, wt = b.y.astype(np.float64), b.weight.astype(np.float64)
    yz = y * (X @ w + b.offset.astype(np.float64))
    f = float(np.sum(wt * (np.maximum(-yz, 0) + log1p(np.exp(-np.abs(yz))))) + 0.5 * np.sum((w - pm) ** 2 / pv))
    p = expit(yz)
    g = X.T @ (wt * (p - 1) * y) + (w - pm) / pv
    Hs = None if s is None else X.T @ (wt * p * (1 - p) * (X @ s)) + s / pv
    return f, g, Hs


def test_linkedin_additions_against_closed_forms_and_scipy_trust_ncg():
    """Pins what the scikit-learn pin cannot reach -- LinkedIn's additions to liblinear's L2-LR: prior MEAN, per-coordinate prior
    VARIANCE, per-row weights, OFFSETS, a warm start != 0, gnorm1 taken at w = 0 with the prior term (llf/LogisticRegressionL2.java
    :156-248, llf/LibLinear.java:221-312, bw/Tron.java:47-62) -- against (a) the objective, gradient and Hessian action written as
    closed-form sparse algebra, (b) scipy's own trust-region Newton-CG (`trust-ncg`, a third-party optimiser) run on those closed
    forms, and (c) the exit rule evaluated by the closed forms at every tolerance of the driver's schedule."""
    from fixtures import load_c1, onehot_blocks
    c1 = load_c1()
    rng = np.random.default_rng(11)
    cases = [c1.blocks[4], onehot_blocks(4000, 1, levels=60).blocks[0]]
    for ci, b0 in enumerate(cases):
        wt = rng.uniform(0.25, 3.0, b0.l).astype(np.float32)
        off = rng.normal(0, 0.5, b0.l).astype(np.float32)
        b = dataset.PartitionBlock(b0.partition_id, b0.l, b0.n_local, b0.row_ptr, b0.col_idx, b0.val, b0.y, wt, off, b0.local_to_global)
        n = b.n_local
        pm = rng.normal(0, 0.3, n)
        pv = rng.uniform(0.2, 5.0, n)
        w0 = rng.normal(0, 0.2, n)
        s = rng.normal(0, 1, n)
        d = ol.OracleDataset.from_block(b)
        # (a) closed forms
        f, g, Hs = d.eval(w0, pm, pv, s)
        fc, gc, Hc = _closed_form(b, w0, pm, pv, s)
        assert abs(f - fc) <= 1e-12 * abs(fc)
        assert np.max(np.abs(g - gc)) <= 1e-11 * np.max(np.abs(gc))
        assert np.max(np.abs(Hs - Hc)) <= 1e-11 * np.max(np.abs(Hc))
        # (b) a third-party trust-region Newton-CG on the closed forms finds the minimiser TRON finds from the warm start
        res = so.minimize(lambda v: _closed_form(b, v, pm, pv)[0], w0, jac=lambda v: _closed_form(b, v, pm, pv)[1],
                          hessp=lambda v, q: _closed_form(b, v, pm, pv, q)[2], method="trust-ncg", options={"gtol": 1e-9, "maxiter": 500})
        # (at the optimum scipy may stop on "failure to predict improvement", i.e. rounding noise; the bound below uses the gradient it reached)
        w, st = d.train(w0, pm, pv, 1e-12)
        # F is strongly convex with modulus 1 / max(priorVar): ||a - b|| <= max(pv) (||grad F(a)|| + ||grad F(b)||) for any two points
        bound = float(np.max(pv)) * (np.linalg.norm(_closed_form(b, res.x, pm, pv)[1]) + np.linalg.norm(_closed_form(b, w, pm, pv)[1]))
        assert bound <= 1e-4 and np.linalg.norm(w - res.x) <= bound * (1 + 1e-6) + 1e-13, (ci, float(np.linalg.norm(w - res.x)), bound)
        # (c) exit rule at every tolerance: ||grad F(w)|| <= eps_tron ||grad F(0)|| (gnorm1 WITH the prior term, at w = 0),
        # eps_tron = epsilon min(pos, neg) / l; and one accepted step fewer would not have satisfied it
        pos = int(np.sum(b.y == 1))
        g0 = np.linalg.norm(_closed_form(b, np.zeros(n), pm, pv)[1])
        last = None
        for eps in (1e-2, 1e-3, 1e-4, 1e-6, 1e-8):
            w, st = d.train(w0, pm, pv, eps)
            eps_tron = eps * min(pos, b.l - pos) / b.l
            assert np.linalg.norm(_closed_form(b, w, pm, pv)[1]) <= eps_tron * g0 * (1 + 1e-9), (ci, eps)
            assert abs(st.gnorm1 - g0) <= 1e-12 * g0
            assert last is None or st.newton_iters >= last          # a tighter tolerance never needs fewer Newton iterations
            last = st.newton_iters


def test_prior_mean_is_a_shift_of_offsets_and_start():
    """TRON on F(w) with prior mean m, from w0, is TRON on G(v) = F(m + v): prior mean 0, offsets + X m, from w0 - m -- the same
    iterates shifted by m, provided the stopping threshold is the same number (LinkedIn's gnorm1 is taken at w = 0, :50-53, which is
    v = -m for G: epsilon is rescaled by the ratio of the two gnorm1). Ties the prior-mean code path to the offset / warm-start path:
    equal Newton, CG and pass counts, coefficients equal to ~1e-8. X m must survive the float32 offset column: binary rows and a
    dyadic m make it exact."""
    from fixtures import onehot_blocks
    b = onehot_blocks(6000, 1, levels=6).blocks[0]          # (few levels: every feature is frequent, the problem well conditioned)
    n = b.n_local
    rng = np.random.default_rng(5)
    m = rng.integers(-64, 64, n) / 256.0
    pv = np.full(n, 0.5)
    w0 = rng.integers(-32, 32, n) / 128.0
    import scipy.sparse as sps
    X = sps.hstack([sps.csr_matrix((np.ones(len(b.col_idx)), b.col_idx, b.row_ptr), shape=(b.l, n - 1)), sps.csr_matrix(np.ones((b.l, 1)))]).tocsr()
    xm = X @ m
    assert np.array_equal(xm.astype(np.float32).astype(np.float64), xm)
    bs = dataset.PartitionBlock(b.partition_id, b.l, n, b.row_ptr, b.col_idx, b.val, b.y, b.weight, xm.astype(np.float32), b.local_to_global)
    d, ds = ol.OracleDataset.from_block(b), ol.OracleDataset.from_block(bs)
    g1_f = np.linalg.norm(d.eval(np.zeros(n), m, pv)[1])
    g1_g = np.linalg.norm(ds.eval(np.zeros(n), np.zeros(n), pv)[1])
    for eps in (1e-2, 1e-4, 1e-7):
        w, st = d.train(w0, m, pv, eps)
        v, sv = ds.train(w0 - m, np.zeros(n), pv, eps * g1_f / g1_g)
        assert (st.newton_iters, st.accepted, st.cg_iters, st.x_passes) == (sv.newton_iters, sv.accepted, sv.cg_iters, sv.x_passes), eps
        assert np.max(np.abs(w - (v + m))) <= 1e-6 * max(1.0, np.max(np.abs(w))), eps      # (measured 1e-8: two roundings of the same iterates)


def test_driver_block_against_its_closed_form(c1):
    """One pass of the driver block (jobs/RegressionAdmmTrain.java:362-405, 736-765) recomputed from the reducers' float32 outputs
    in three numpy lines: xbar / ubar with the fixed divisor num.blocks, z = c xbar + c ubar with the FLOAT weight c and the
    intercept rule, u_k = f32(f32(u_k + beta_k) - z)."""
    lam, rho, N = 3.0, 1.0, len(c1.blocks)
    oc = ol.OracleAdmm(c1.blocks, c1.n_global, [lam], [rho])
    for it in range(3):
        u_prev = np.stack([oc.partition_model(k, 0)[2] for k in range(N)]) if it else np.zeros((N, c1.n_global), np.float32)
        oc.solve_local(0.01, 1.0, nthreads=2)
        oc.finish()
        B = np.stack([oc.partition_model(k, 0)[0] for k in range(N)]).astype(np.float64)
        UPX = np.stack([oc.partition_model(k, 0)[1] for k in range(N)])
        xbar = np.zeros(c1.n_global)
        ubar = np.zeros(c1.n_global)
        for k in range(N):                      # sequential, file order; weight 1/N (utils/LinearModelUtils.java:77-84)
            xbar += (1.0 / N) * B[k]
            ubar += (1.0 / N) * u_prev[k].astype(np.float64)
        c = float(np.float32(N * np.float32(rho) / (np.float32(lam) + N * np.float32(rho))))
        z = c * xbar + c * ubar
        z[-1] = xbar[-1] + ubar[-1]             # the intercept is not shrunk (penalize.intercept = false)
        Z = oc.z()[0][0]
        assert np.array_equal(Z, z), float(np.max(np.abs(Z - z)))
        u_next = np.stack([oc.partition_model(k, 0)[2] for k in range(N)])
        assert np.array_equal(u_next, (UPX.astype(np.float64) - z).astype(np.float32))


def test_sequential_sum_is_the_exact_sum_of_grid_rounded_terms_within_a_binade():
    """The mechanism behind the step kernels' grid-rounded dots (mlx_kernels.hip: round_to_grid): while a running
    sum stays inside one binade, `s = fl(s + x)` equals `s + rnd_u(x)` EXACTLY, where rnd_u rounds x to the nearest multiple of u = ulp(s)
    -- the rounding depends on (x, binade of s), not on the order of the terms -- and `(x + 1.5 * 2^52 u) - 1.5 * 2^52 u` computes
    rnd_u(x). So the sequential loop of bw/Tron.java:204-213 and ANY parallel sum of the rounded terms give the same bits (sums of
    multiples of u are exact), while the plain parallel (or exact) sum of the unrounded terms lands tens of ulp away."""
    import math
    rng = np.random.default_rng(7)
    off = []
    for trial in range(20):
        s0 = float(rng.uniform(1.0, 1.2)) * 2.0 ** int(rng.integers(-20, 20))           # a running sum low in its binade
        u = np.spacing(s0)
        x = rng.lognormal(-14.0, 3.0, 20000) * s0 / 20000.0                                 # small positive terms: total growth < 2x
        x = x[np.cumsum(x) < 0.7 * s0]                                                       # (stay inside the binade)
        seq = s0
        for v in x:                                                                          # the reference's loop
            seq = seq + float(v)
        magic = 1.5 * 2.0 ** 52 * u
        xr = (x + magic) - magic                                                             # the kernels' round_to_grid
        assert np.all(np.abs(xr - x) <= 0.5 * u) and np.all(np.round(xr / u) == xr / u)      # nearest multiples of u
        for order in (np.arange(len(x)), rng.permutation(len(x)), np.argsort(x)):            # any order, any association
            tree = xr[order].copy()
            while len(tree) > 1:
                if len(tree) % 2:
                    tree = np.append(tree, 0.0)
                tree = tree[0::2] + tree[1::2]
            assert s0 + float(tree[0]) == seq, trial
        off.append(abs(s0 + math.fsum(float(v) for v in x) - seq) / u)
    # the exact sum of the UNROUNDED terms is somewhere else (ulps of s): the sequential loop's rounding errors are what it lacks
    assert np.median(off) >= 3.0, off


def test_experiment_switches_live_in_their_own_translation_unit(c1):
    """the summation-order experiments (tools/sum_order_experiment.py) sat inside the checker's hot functions. They are
    oracle/experiments.c now; admm_oracle.c keeps one `if (orc_hooks.x)` line per function, NULL by default. The hooks still work
    (a compensated Tron.dot changes last bits, not the solution) and switching them off restores the reference's bits."""
    import ctypes
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(ROOT, "oracle", "admm_oracle.c")).read()
    for word in ("acc2", "g_sum_mode", "magic", "frexp"):
        assert word not in src, word
    assert src.count("orc_hooks.") == 10 and "orc_set_sum_mode" in open(os.path.join(ROOT, "oracle", "experiments.c")).read()
    L = ol.lib()
    L.orc_set_sum_mode.argtypes = [ctypes.c_int]
    L.orc_get_sum_mode.restype = ctypes.c_int
    assert L.orc_get_sum_mode() == 0
    b = c1.blocks[0]
    ds = ol.OracleDataset.from_block(b)
    init, pm, pv = np.zeros(b.n_local), np.zeros(b.n_local), np.ones(b.n_local)
    w0, st0 = ds.train(init, pm, pv, 1e-6)
    try:
        L.orc_set_sum_mode(1 | 2 | 4 | 8)                 # compensated dots, pass sums, norms, loss
        w1, st1 = ds.train(init, pm, pv, 1e-6)
    finally:
        L.orc_set_sum_mode(0)
    w2, st2 = ds.train(init, pm, pv, 1e-6)
    assert np.array_equal(w0, w2) and st0.cg_iters == st2.cg_iters
    assert not np.array_equal(w0, w1) and np.max(np.abs(w1 - w0)) <= 1e-9 * max(1.0, np.max(np.abs(w0)))


def test_portable_exp_and_log1p_against_libm_in_ulps(tmp_path):
    """the reference-order contract evaluates csrc/portable_math.h on BOTH sides (kernels and oracle twin), so
    bit-identity to the twin pins the summation order but says nothing about these two functions. Direct sweep against glibc
    (tests/native/pm_ulp_sweep.c): pm_exp over [-745, 709] with dense bands around 0 and the logistic range, pm_log1p over (-1, 1e308)
    on both signs, around 0, near -1 and on u = exp(-z) -- what row_eval feeds it. Bounds: exp 1 ulp, log1p 3 ulp (Java specifies
    Math.exp / Math.log1p to 1 ulp); the special values (0, +-inf, NaN, the overflow / underflow thresholds, log1p(-1), log1p(< -1))
    agree."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pm_ulp_sweep")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "ml-ease_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "pm_ulp_sweep.c"), "-o", exe, "-lm"], check=True, timeout=120)
    out = subprocess.run([exe, "400000"], capture_output=True, text=True, check=True, timeout=300).stdout
    vals = {ln.split()[0]: float(ln.split()[1]) for ln in out.splitlines() if ln.split() and ln.split()[0] in ("exp_max_ulp", "log1p_max_ulp", "specials_bad")}
    assert vals["exp_max_ulp"] <= 1.0, out
    assert vals["log1p_max_ulp"] <= 3.0, out
    assert vals["specials_bad"] == 0, out
