"""Loaders for the committed fixtures under tests/golden/ (see make_golden_c1.py)."""
import os

import numpy as np

import mlease_amd  # noqa: F401
from mlease_amd.dataset import PartitionBlock, PartitionedData

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_c1() -> PartitionedData:
    z = np.load(os.path.join(GOLDEN, "c1_partitions.npz"))
    nb = int(z["num_blocks"])
    blocks = []
    for k in range(nb):
        p = "p%d_" % k
        rp = z[p + "row_ptr"].astype(np.int64)
        l2g = z[p + "l2g"].astype(np.int32)
        blocks.append(PartitionBlock(k, len(rp) - 1, len(l2g), rp, z[p + "col_idx"].astype(np.int32),
                                     z[p + "val"].astype(np.float32), z[p + "y"].astype(np.int8),
                                     z[p + "weight"].astype(np.float32), z[p + "offset"].astype(np.float32), l2g))
    return PartitionedData(blocks, [str(s) for s in z["feature_names"]], nb)


def load_c1_golden():
    return np.load(os.path.join(GOLDEN, "c1_golden.npz"))


def synth_sparse(seed, nrows, nfeat, nnz_per_row, num_blocks, binary=False, weights=False, offsets=False):
    """Small seeded sparse problem with partition-local feature spaces (some features absent per partition)."""
    rng = np.random.default_rng(seed)
    beta = rng.normal(0, 0.5, nfeat)
    blocks_rows = [[] for _ in range(num_blocks)]
    for i in range(nrows):
        m = max(1, int(rng.poisson(nnz_per_row)))
        cols = rng.choice(nfeat, size=min(m, nfeat), replace=False)
        vals = np.ones(len(cols), np.float32) if binary else rng.normal(0, 1, len(cols)).astype(np.float32)
        s = float(np.dot(beta[cols], vals)) - 0.5
        y = 1 if rng.random() < 1 / (1 + np.exp(-s)) else 0
        w = np.float32(rng.uniform(0.5, 2.0)) if weights else np.float32(1)
        o = np.float32(rng.normal(0, 0.3)) if offsets else np.float32(0)
        blocks_rows[i % num_blocks].append((cols, vals, y, w, o))
    gseen = {}
    blocks = []
    for k, rows in enumerate(blocks_rows):
        lidx = {}
        rp, ci, vv, ys, ws, os_ = [0], [], [], [], [], []
        for cols, vals, y, w, o in rows:
            ent = []
            for c, v in zip(cols, vals):
                if c not in lidx:
                    lidx[c] = len(lidx)
                    gseen.setdefault(int(c), len(gseen))
                ent.append((lidx[c], v))
            ent.sort(key=lambda e: e[0])
            ci += [e[0] for e in ent]
            vv += [e[1] for e in ent]
            rp.append(len(ci))
            ys.append(1 if y == 1 else -1)
            ws.append(w)
            os_.append(o)
        inv = sorted(lidx.items(), key=lambda kv: kv[1])
        blocks.append((k, rp, ci, vv, ys, ws, os_, [c for c, _ in inv]))
    ng = len(gseen) + 1
    out = []
    for k, rp, ci, vv, ys, ws, os_, lcols in blocks:
        l2g = np.asarray([gseen[int(c)] for c in lcols] + [ng - 1], np.int32)
        out.append(PartitionBlock(k, len(ys), len(l2g), np.asarray(rp, np.int64), np.asarray(ci, np.int32),
                                  None if binary else np.asarray(vv, np.float32), np.asarray(ys, np.int8),
                                  np.asarray(ws, np.float32), np.asarray(os_, np.float32), l2g))
    names = [None] * (ng - 1)
    for c, g in gseen.items():
        names[g] = "f%d" % c
    return PartitionedData(out, names, num_blocks)


def onehot_blocks(rows, partitions, seed=5, fields=20, levels=5000):
    """BASELINE configs[2]-style data at any scale: `fields` categorical fields x `levels` one-hot binary features,
    Zipf(1.1) levels, rare positives (intercept -3); partition-local compaction (sorted local ids)."""
    rng = np.random.default_rng(seed)
    p = np.arange(1, levels + 1, dtype=np.float64) ** -1.1
    cdf = np.cumsum(p / p.sum())
    beta = rng.normal(0, 0.3, fields * levels)
    ng = fields * levels + 1
    blocks = []
    per = (rows + partitions - 1) // partitions
    for k in range(partitions):
        l = min(per, rows - k * per)
        lev = np.minimum(np.searchsorted(cdf, rng.random((l, fields))).astype(np.int32), levels - 1)
        gid = lev + (np.arange(fields, dtype=np.int32) * levels)[None, :]
        logit = beta[gid].sum(axis=1) - 3.0
        y = np.where(rng.random(l) < 1 / (1 + np.exp(-logit)), 1, -1).astype(np.int8)
        uniq, inv = np.unique(gid.reshape(-1), return_inverse=True)
        ci = np.sort(inv.reshape(l, fields).astype(np.int32), axis=1).reshape(-1)
        blocks.append(PartitionBlock(k, l, len(uniq) + 1, np.arange(0, (l + 1) * fields, fields, dtype=np.int64), ci, None, y,
                                     np.ones(l, np.float32), np.zeros(l, np.float32),
                                     np.concatenate([uniq.astype(np.int32), [ng - 1]]).astype(np.int32)))
    return PartitionedData(blocks, [str(i) for i in range(ng - 1)], partitions)


def dense_blocks(rows, nfeat, partitions, seed=9):
    """BASELINE configs[1]-style data in the small: dense N(0,1) float32 features handed over as CSR rows with every column present
    (the library turns such a partition into a dense tile; the verification mode keeps it CSR)."""
    rng = np.random.default_rng(seed)
    beta = rng.normal(0, 0.1, nfeat)
    blocks = []
    per = rows // partitions
    for k in range(partitions):
        X = rng.normal(0, 1, (per, nfeat)).astype(np.float32)
        y = np.where(rng.random(per) < 1 / (1 + np.exp(-(X.astype(np.float64) @ beta - 1.0))), 1, -1).astype(np.int8)
        blocks.append(PartitionBlock(k, per, nfeat + 1, np.arange(0, (per + 1) * nfeat, nfeat, dtype=np.int64),
                                     np.tile(np.arange(nfeat, dtype=np.int32), per), X.reshape(-1), y,
                                     np.ones(per, np.float32), np.zeros(per, np.float32), np.arange(nfeat + 1, dtype=np.int32)))
    return PartitionedData(blocks, [str(i + 1) for i in range(nfeat)], partitions)


def permute_rows(b, seed=0, relabel=False):
    """Same partition, rows in another order (the reference's row order within a reducer key is unspecified).
    relabel=True also renumbers the partition-local features the way the reference's indexing would for that row order:
    ids in first-seen order (llf/LibLinearDataset.java:467-478), every row re-sorted by id (:481-482) -- so the oracle's
    n-long dots and norms run in another order too, not only its row-order sums."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(b.l)
    lens = np.diff(b.row_ptr)
    rp = np.concatenate([[0], np.cumsum(lens[perm])]).astype(np.int64)
    starts = np.repeat(b.row_ptr[:-1][perm] - rp[:-1], lens[perm])
    idx = np.arange(rp[-1], dtype=np.int64) + starts
    cols = b.col_idx[idx]
    vals = None if b.val is None else b.val[idx]
    l2g = b.local_to_global
    if relabel:
        nf = b.n_local - 1
        uniq, first = np.unique(cols, return_index=True)
        seen = uniq[np.argsort(first, kind="stable")]                    # features in first-seen order
        rest = np.setdiff1d(np.arange(nf, dtype=cols.dtype), uniq)       # (never present: keep them behind)
        order = np.concatenate([seen, rest]).astype(np.int64)
        newid = np.empty(nf, np.int64)
        newid[order] = np.arange(nf)
        cols = newid[cols]
        rowid = np.repeat(np.arange(b.l, dtype=np.int64), lens[perm])
        srt = np.lexsort((cols, rowid))                                  # stable: duplicates keep their order
        cols = cols[srt].astype(np.int32)
        vals = None if vals is None else vals[srt]
        l2g = np.concatenate([b.local_to_global[:nf][order], b.local_to_global[nf:]]).astype(np.int32)
    return PartitionBlock(b.partition_id, b.l, b.n_local, rp, cols, vals, b.y[perm], b.weight[perm], b.offset[perm], l2g)


def c1_raw_records(c1, with_key=True):
    """The C1 fixture back as raw Pig-style records, rows interleaved across partitions like the original file."""
    recs = []
    maxl = max(b.l for b in c1.blocks)
    for i in range(maxl):
        for b in c1.blocks:
            if i >= b.l:
                continue
            sl = slice(b.row_ptr[i], b.row_ptr[i + 1])
            feats = [{"name": c1.feature_names[b.local_to_global[c]], "term": "", "value": float(v)}
                     for c, v in zip(b.col_idx[sl], b.val[sl])]
            recs.append({"features": feats, "offset": 0, "response": 1 if b.y[i] == 1 else 0, "weight": 1,
                         "pkey": b.partition_id if with_key else None, "unused": None})
    return recs
