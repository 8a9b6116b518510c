#!/usr/bin/env python3
"""CPU experiment (no GPU): how far does a SUMMATION ORDER move the reference's TRON trajectory on one-hot data, and what would
an order-independent (compensated) implementation gain?

Per ADMM iteration of a closed P-block job on full-size configs[2] partitions, every solve starts from the BASE oracle's state
(z, u_k) at that iteration, so nothing compounds across iterations. Compared with the base oracle's solve (oracle/admm_oracle.c,
sequential Java loop order, caller's row / column order):

  perm[k]      the oracle on the partition with rows permuted and features renumbered in first-seen order (what the reference's
               own indexing does when Hadoop hands it the rows in another order: llf/LibLinearDataset.java:467-482), NPERM seeds
  scalars      orc_set_sum_mode(13): the n- and l-long scalar reductions (dot, norm, loss, prior) as compensated sums
  all          orc_set_sum_mode(15): ... and the row / column sums of Xv / XTv too  (= what an order-independent kernel computes)
  only_*       single switches (oracle/admm_oracle.c: orc_set_sum_mode bits)
  all+perm     mode 15 on a permuted partition: if the sums are order-independent this equals `all`

Reported per iteration: solves with TRON counters equal to the base solve, median / max relative error of beta (floor 1e-4 max).

    python tools/sum_order_experiment.py [--partitions 8] [--rows 39063] [--iters 6] [--perms 8] [--threads 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import mlease_amd  # noqa: F401,E402
from mlease_amd import admm  # noqa: E402
from mlease_amd.dataset import PartitionBlock  # noqa: E402
import oracle_lib as ol  # noqa: E402
import synth_data as sd  # noqa: E402
from fixtures import permute_rows  # noqa: E402


def rel_err(a, ref):
    a, ref = a.astype(np.float64), ref.astype(np.float64)
    fl = 1e-4 * np.max(np.abs(ref), axis=1, keepdims=True)
    return np.max(np.abs(a - ref) / np.maximum(np.abs(ref), np.maximum(fl, 1e-300)), axis=1)


def counters(o):
    return np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in o.stats()], np.int32)


def betas(o, P):
    return np.stack([o.partition_model(k, 0)[0] for k in range(P)])


VB_FIELDS, VB_LEVELS = 12, 3000


def valued_b_partition(pid, rows, seed=777):
    """The second data set (--dataset valued_b): flatter level frequencies, fewer and valued entries per row, weights, offsets."""
    rng = np.random.default_rng([seed, 3, pid])
    p = np.arange(1, VB_LEVELS + 1, dtype=np.float64) ** -0.8
    cdf = np.cumsum(p / p.sum())
    beta = np.random.default_rng([seed, 2]).normal(0, 0.4, VB_FIELDS * VB_LEVELS)
    ng = VB_FIELDS * VB_LEVELS + 1
    lev = np.minimum(np.searchsorted(cdf, rng.random((rows, VB_FIELDS))), VB_LEVELS - 1).astype(np.int32)
    gid = lev + (np.arange(VB_FIELDS, dtype=np.int32) * VB_LEVELS)[None, :]
    val = rng.lognormal(0.0, 0.35, (rows, VB_FIELDS)).astype(np.float32)
    logit = (beta[gid] * val).sum(axis=1) - 1.0
    y = np.where(rng.random(rows) < 1 / (1 + np.exp(-logit)), 1, -1).astype(np.int8)
    uniq, inv = np.unique(gid.reshape(-1), return_inverse=True)
    loc = inv.reshape(rows, VB_FIELDS).astype(np.int32)
    order = np.argsort(loc, axis=1, kind="stable")
    ci = np.take_along_axis(loc, order, axis=1).reshape(-1)
    vv = np.take_along_axis(val, order, axis=1).reshape(-1)
    l2g = np.concatenate([uniq.astype(np.int32), [ng - 1]]).astype(np.int32)
    wt = rng.uniform(0.5, 2.0, rows).astype(np.float32)
    off = rng.normal(0, 0.2, rows).astype(np.float32)
    return PartitionBlock(pid, rows, len(l2g), np.arange(0, (rows + 1) * VB_FIELDS, VB_FIELDS, dtype=np.int64), ci, vv, y, wt, off, l2g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partitions", type=int, default=8)
    ap.add_argument("--rows", type=int, default=39063)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--perms", type=int, default=8)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--json", default="")
    ap.add_argument("--minimal", action="store_true", help="only the base oracle, the permuted oracles and (--gpu) the HIP path")
    ap.add_argument("--gpu", action="store_true", help="add the HIP product path (needs an MI355X) as one more variant")
    ap.add_argument("--dataset", choices=["onehot", "valued_b"], default="onehot",
                    help="onehot = tools/synth_data.onehot_partition (configs[2]); valued_b = a SECOND data set for the head-column choice of the "
                         "grid-rounded dots: 12 fields x 3000 levels, Zipf exponent 0.8, 27 %% positives, real values, row weights and offsets")
    ap.add_argument("--heads-only", action="store_true", help="only the permuted oracles and the grid-rounded-dot variants (head 64 / 256 / running prefix, trees)")
    a = ap.parse_args()
    P = a.partitions
    blocks, ng = [], None
    for k in range(P):
        if a.dataset == "onehot":
            rp, ci, y, l2g, ng = sd.onehot_partition(k, a.rows)
            blocks.append(PartitionBlock(k, a.rows, len(l2g), rp, ci, None, y, np.ones(a.rows, np.float32), np.zeros(a.rows, np.float32), l2g))
        else:
            blocks.append(valued_b_partition(k, a.rows))
            ng = VB_FIELDS * VB_LEVELS + 1
    L = ol.lib()
    L.orc_set_sum_mode.argtypes = [__import__("ctypes").c_int]
    base = ol.OracleAdmm(blocks, ng, [1.0], [1.0])
    variants = {"perm%d" % i: (0, ol.OracleAdmm([permute_rows(b, 100 * i + 7 + j, relabel=True) for j, b in enumerate(blocks)], ng, [1.0], [1.0]))
                for i in range(a.perms)}
    if not a.minimal:
        variants["scalars"] = (13, ol.OracleAdmm(blocks, ng, [1.0], [1.0]))
        variants["all"] = (15, ol.OracleAdmm(blocks, ng, [1.0], [1.0]))
        variants["all+perm"] = (15, ol.OracleAdmm([permute_rows(b, 5000 + j, relabel=True) for j, b in enumerate(blocks)], ng, [1.0], [1.0]))
        variants["scalars+perm"] = (13, ol.OracleAdmm([permute_rows(b, 5000 + j, relabel=True) for j, b in enumerate(blocks)], ng, [1.0], [1.0]))
        def freq_order(b):
            """columns renumbered most-frequent-first (the library's own order of the n-vectors), rows as they are"""
            nf = b.n_local - 1
            cnt = np.bincount(b.col_idx, minlength=nf)
            order = np.argsort(-cnt, kind="stable")
            newid = np.empty(nf, np.int64)
            newid[order] = np.arange(nf)
            cols = newid[b.col_idx]
            rowid = np.repeat(np.arange(b.l, dtype=np.int64), np.diff(b.row_ptr))
            srt = np.lexsort((cols, rowid))
            l2g = np.concatenate([b.local_to_global[:nf][order], b.local_to_global[nf:]]).astype(np.int32)
            return PartitionBlock(b.partition_id, b.l, b.n_local, b.row_ptr, cols[srt].astype(np.int32), None if b.val is None else b.val[srt], b.y, b.weight, b.offset, l2g)
        fb = [freq_order(b) for b in blocks]
        for name, mode in (("freq_order", 0), ("freq_order+grid2048", 128), ("freq_order+grid2048+passes", 130), ("freq_order+tree+passes", 66)):
            variants[name] = (mode, ol.OracleAdmm(fb, ng, [1.0], [1.0]))
        variants["perm+grid2048+passes"] = (130, ol.OracleAdmm([permute_rows(b, 5000 + j, relabel=True) for j, b in enumerate(blocks)], ng, [1.0], [1.0]))
        for name, mode in (("dot", 1), ("passes", 2), ("norm", 4), ("fun", 8), ("plainnorm", 16), ("dot+fun", 9), ("gridrounded_dot", 32), ("tree_dot", 64), ("tree_dot+passes", 66), ("gridrounded_dot+passes", 34), ("grid2048_dot", 128), ("grid64_dot", 256), ("grid2048_dot+passes", 130)):
            variants["only_" + name] = (mode, ol.OracleAdmm(blocks, ng, [1.0], [1.0]))

    eng = None
    if a.gpu:
        from mlease_amd.hip_engine import HipAdmmEngine
        eng = HipAdmmEngine(ng, [1.0], [1.0], P)
        eng.add_partitions(blocks)
        eng.finalize()
    # per-call-site mixes on the frequency-ordered partitions (the library's column order): which dots must be grid-rounded?
    L.orc_set_dot_site_mode.argtypes = [__import__("ctypes").c_int, __import__("ctypes").c_int]
    site_mixes = {"fo:sites012=grid,345=tree,+passes": ({0: 128, 1: 128, 2: 128, 3: 64, 4: 64, 5: 64}, 2),
                  "fo:sites012=gridK1,345=tree,+passes": ({0: 512, 1: 512, 2: 512, 3: 64, 4: 64, 5: 64}, 2),
                  "fo:sites012=gridK4,345=tree,+passes": ({0: 1024, 1: 1024, 2: 1024, 3: 64, 4: 64, 5: 64}, 2),
                  "fo:sites012=grid_first256,345=tree,+passes": ({0: 2048, 1: 2048, 2: 2048, 3: 64, 4: 64, 5: 64}, 2),
                  "fo:sites012=grid_first64,345=tree,+passes": ({0: 4096, 1: 4096, 2: 4096, 3: 64, 4: 64, 5: 64}, 2),
                  "fo:sites012=grid_first256,345=tree (tree passes)": ({0: 2048, 1: 2048, 2: 2048, 3: 64, 4: 64, 5: 64}, 0),
                  "fo:sites12=grid,0345=tree,+passes": ({0: 64, 1: 128, 2: 128, 3: 64, 4: 64, 5: 64}, 2),
                  "fo:sites01245=grid,3=tree,+passes": ({0: 128, 1: 128, 2: 128, 3: 64, 4: 128, 5: 128}, 2),
                  "fo:all sites grid,+passes": ({}, 130)}
    for name in ([] if a.minimal else site_mixes):
        variants[name] = (("mix", name), ol.OracleAdmm(fb, ng, [1.0], [1.0]))
    if a.heads_only:
        keep = ("perm", "freq_order", "fo:sites012=grid,345=tree,+passes", "fo:sites012=grid_first256,345=tree,+passes",
                "fo:sites012=grid_first64,345=tree,+passes", "freq_order+tree+passes")
        variants = {k: v for k, v in variants.items() if k.startswith("perm") and not k.startswith("perm+") or k in keep}
    e, mind = np.float32(0.01), 99999999.0
    out = []
    t0 = time.time()
    for it in range(1, a.iters + 1):
        if it > 1 and mind < 0.001:
            e = np.float32(e / np.float32(10))
        eps = admm.float_string_roundtrip(e)
        Z = base.z()[0].copy()
        U = np.stack([base.partition_model(k, 0)[2] for k in range(P)])[:, None, :].copy() if it > 1 else np.zeros((P, 1, ng), np.float32)
        L.orc_set_sum_mode(0)
        base.set_state(Z, U)
        base.solve_local(eps, 1.0, nthreads=a.threads)
        cb, bb = counters(base), betas(base, P)
        rec = {"iteration": it, "epsilon": eps, "cg_per_solve": float(cb[:, 2].mean()), "newton_per_solve": float(cb[:, 0].mean())}
        for name, (mode, o) in variants.items():
            for st_ in range(6):
                L.orc_set_dot_site_mode(st_, -1)
            if isinstance(mode, tuple):
                sites, gm = site_mixes[mode[1]]
                for st_, m_ in sites.items():
                    L.orc_set_dot_site_mode(st_, m_)
                mode = gm
            L.orc_set_sum_mode(mode)
            o.set_state(Z, U)
            o.solve_local(eps, 1.0, nthreads=a.threads)
            er = rel_err(betas(o, P), bb)
            eqv = np.all(counters(o) == cb, axis=1)
            rec[name] = {"equal": int(eqv.sum()), "median": float(np.median(er)), "max": float(er.max()),
                         "equal_by_solve": [int(x) for x in eqv], "err_by_solve": [float(x) for x in er]}
        L.orc_set_sum_mode(0)
        for st_ in range(6):
            L.orc_set_dot_site_mode(st_, -1)
        if eng is not None:
            eng.set_state(Z, U)
            eng.solve_local(eps, 1.0)
            gb = np.stack([eng.partition_model(k, 0)[0] for k in range(P)])
            er = rel_err(gb, bb)
            eqv = np.all(eng.solve_counters() == cb, axis=1)
            rec["gpu"] = {"equal": int(eqv.sum()), "median": float(np.median(er)), "max": float(er.max()),
                          "equal_by_solve": [int(x) for x in eqv], "err_by_solve": [float(x) for x in er]}
            easy = np.ones(P, bool)
            for i in range(a.perms):
                easy &= np.array(rec["perm%d" % i]["equal_by_solve"], bool)
            rec["easy_solves"] = int(easy.sum())
            rec["gpu_equal_on_easy_solves"] = int((eqv & easy).sum())
            pe_ = [rec["perm%d" % i] for i in range(a.perms)]
            print("   easy solves (every permuted oracle keeps the trajectory): %d, the GPU keeps it on %d of them" % (rec["easy_solves"], rec["gpu_equal_on_easy_solves"]))
            print("   gpu: equal %d (perms %d..%d)  median %.2e (perms %.2e..%.2e)  max %.2e (perms %.2e..%.2e)" % (
                rec["gpu"]["equal"], min(p["equal"] for p in pe_), max(p["equal"] for p in pe_), rec["gpu"]["median"],
                min(p["median"] for p in pe_), max(p["median"] for p in pe_), rec["gpu"]["max"], min(p["max"] for p in pe_), max(p["max"] for p in pe_)), flush=True)
        if a.minimal or a.heads_only:
            mind = base.finish()[1]
            out.append(rec)
            print("it %d eps %.3g cg/solve %.1f | %s  (%.0f s)" % (it, eps, rec["cg_per_solve"], {k: rec[k]["equal"] for k in variants}, time.time() - t0), flush=True)
            continue
        # `all` against `all+perm`: order independence of the compensated arithmetic itself
        ea = rel_err(betas(variants["all+perm"][1], P), betas(variants["all"][1], P))
        rec["all+perm_vs_all"] = {"equal": int(np.all(counters(variants["all+perm"][1]) == counters(variants["all"][1]), axis=1).sum()),
                                  "median": float(np.median(ea)), "max": float(ea.max()),
                                  "bit_identical_f32": float(np.mean(betas(variants["all+perm"][1], P) == betas(variants["all"][1], P)))}
        mind = base.finish()[1]            # the base job moves on with its own consensus (driver's mindiff)
        out.append(rec)
        pe = [rec["perm%d" % i] for i in range(a.perms)]
        print("it %d eps %.3g cg/solve %.1f | perm equal %s median %.2e..%.2e | scalars %d %.2e | all %d %.2e | scalars+perm %d %.2e | all+perm %d %.2e | all+perm vs all: %d %.2e ident %.3f  (%.0f s)" % (
            it, eps, rec["cg_per_solve"], sorted(p["equal"] for p in pe), min(p["median"] for p in pe), max(p["median"] for p in pe),
            rec["scalars"]["equal"], rec["scalars"]["median"], rec["all"]["equal"], rec["all"]["median"],
            rec["scalars+perm"]["equal"], rec["scalars+perm"]["median"], rec["all+perm"]["equal"], rec["all+perm"]["median"],
            rec["all+perm_vs_all"]["equal"], rec["all+perm_vs_all"]["median"], rec["all+perm_vs_all"]["bit_identical_f32"], time.time() - t0), flush=True)
    print("totals over iterations (solves with counters equal to the base solve):")
    for name in variants:
        print("  %-22s %s  sum %d" % (name, [r[name]["equal"] for r in out], sum(r[name]["equal"] for r in out)))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump({"partitions": P, "rows": a.rows, "perms": a.perms, "dataset": a.dataset, "per_iteration": out}, fh, indent=1)


if __name__ == "__main__":
    main()
