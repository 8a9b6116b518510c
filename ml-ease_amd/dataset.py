"""Host-side row preparation and partition building for the ADMM hot path.

Mirrors, for the formats that feed the path,
  * ``RegressionPrepare.RegressionPrepareMapper.map`` (jobs/RegressionPrepare.java:96-191):
    response lookup, float casts, weight / num.click.replicates, partition key;
  * ``LibLinearDataset.addInstanceAvro`` + ``finish`` (llf/LibLinearDataset.java:413-484,586-658;
    binary: llf/LibLinearBinaryDataset.java:426-515,541-620): partition-local feature ids in
    first-seen order, name key = ``name + "\\u0001" + term`` when term is non-empty, y in {-1,+1},
    default weight 1 / offset 0, rejection rules.

The output is the CSR block layout the C-ABI takes (include/mlease_admm.h): 0-based
partition-local column ids WITHOUT the intercept entry; the library appends the intercept as local
index ``n_local-1`` with value 1.0 exactly as ``finish()`` does.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

INTERCEPT_NAME = "(INTERCEPT)"          # llf/LibLinearDataset.java:92
TERM_SEP = "\u0001"                     # llf/LibLinearDataset.java:458-459


class ModelFittingError(IOError):
    """Raised where the reference throws IOException (jobs/RegressionAdmmTrain.java:713-716)."""


def feature_key(name: str, term: str) -> str:
    return name if term == "" else name + TERM_SEP + term


def get_response(record: Dict[str, Any]) -> int:
    """utils/Util.java:309-337: last non-null of click / response / label; Boolean or Integer."""
    resp = None
    for k in ("click", "response", "label"):
        if record.get(k) is not None:
            resp = record[k]
    if resp is None:
        raise IOError("Data should contain one field of the three: response, click or label!")
    if isinstance(resp, bool):
        return 1 if resp else 0
    if not isinstance(resp, (int, np.integer)):
        raise IOError("Response/Click/Label column should be either boolean or int32!")
    return int(resp)


@dataclass
class PreparedRow:                       # avro/RegressionPrepareOutput.avsc:16-34
    key: str
    response: int
    features: List[tuple]                # (name, term, float32 value)
    weight: np.float32
    offset: np.float32

    def to_avro(self) -> Dict[str, Any]:
        return {"key": self.key, "response": self.response,
                "features": [{"name": n, "term": t, "value": float(v)} for n, t, v in self.features],
                "weight": float(self.weight), "offset": float(self.offset)}

    @staticmethod
    def from_avro(rec: Dict[str, Any]) -> "PreparedRow":
        return PreparedRow(str(rec["key"]), int(rec["response"]),
                           [(f["name"], f["term"], np.float32(f["value"])) for f in rec["features"]],
                           np.float32(rec["weight"]), np.float32(rec["offset"]))


def prepare_rows(records: Iterable[Dict[str, Any]], num_blocks: int, map_key: str = "",
                 binary_feature: bool = False, num_click_replicates: int = 1,
                 key_fn: Optional[Callable[[int, Dict[str, Any]], int]] = None,
                 rng: Optional[random.Random] = None) -> List[PreparedRow]:
    """jobs/RegressionPrepare.java:96-191.

    ``key_fn(row_index, record)`` replaces the reference's ``Math.random()`` key (:112), which is
    non-deterministic; when neither ``map_key`` nor ``key_fn`` is given a seeded ``rng`` is used.
    """
    out: List[PreparedRow] = []
    rng = rng or random.Random(0)
    for idx, data in enumerate(records):
        if map_key != "":
            if data.get(map_key) is None:
                raise IOError("map.key is wrongly specified! No such key exists in some lines of the data!")
            mapkey = str(data[map_key])
        elif key_fn is not None:
            mapkey = str(int(key_fn(idx, data)))
        else:
            mapkey = str(int(math.floor(rng.random() * num_blocks)))      # :112
        response = get_response(data)
        feats = data.get("features")
        if feats is None:
            raise IOError("features is null")
        if not isinstance(feats, list):
            raise IOError("features is not a list")
        newf = []
        for i, f in enumerate(feats):
            if not isinstance(f, dict):
                raise IOError("features[%d] is not a record" % i)
            if f.get("name") is None:
                raise IOError("name is null")
            name = str(f["name"])
            term = "" if f.get("term") is None else str(f["term"])
            value = np.float32(1.0)
            if not binary_feature:
                if f.get("value") is None:
                    raise IOError("value is null")
                value = np.float32(float(f["value"]))                    # :145
            newf.append((name, term, value))
        weight = 1.0
        if data.get("weight") is not None:
            weight = float(data["weight"])
        r = data.get("response")
        if r is None or isinstance(r, bool) or not isinstance(r, (int, np.integer)):
            raise IOError("response is not an integer")                  # Util.getIntAvro :159
        if int(r) == 1:
            weight = weight / num_click_replicates                        # :159-162
        offset = 0.0
        if data.get("offset") is not None:
            offset = float(data["offset"])
        row = PreparedRow(mapkey, response, newf, np.float32(weight), np.float32(offset))
        if map_key == "" and response == 1:                               # :172-186
            pid = int(mapkey)
            for _ in range(num_click_replicates):
                if pid >= num_blocks:
                    pid -= num_blocks
                out.append(PreparedRow(str(pid), row.response, row.features, row.weight, row.offset))
                pid += 1
        else:
            out.append(row)
    return out


@dataclass
class PartitionBlock:
    """One partition's rows in the C-ABI layout (see include/mlease_admm.h, mlx_add_partition_csr)."""
    partition_id: int
    l: int
    n_local: int                         # incl. intercept (last local index)
    row_ptr: np.ndarray                  # int64 [l+1]
    col_idx: np.ndarray                  # int32 [nnz], 0-based local ids, intercept NOT stored
    val: Optional[np.ndarray]            # float32 [nnz] or None (binary.feature)
    y: np.ndarray                        # int8 [l] in {-1,+1}
    weight: np.ndarray                   # float32 [l]
    offset: np.ndarray                   # float32 [l]
    local_to_global: np.ndarray          # int32 [n_local]; intercept -> n_global-1

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])


@dataclass
class PartitionedData:
    blocks: List[PartitionBlock]
    feature_names: List[str]             # global index -> name key (without the intercept)
    num_blocks: int

    @property
    def n_global(self) -> int:
        return len(self.feature_names) + 1


def build_partitions(rows: Sequence[PreparedRow], num_blocks: int, binary_feature: bool = False,
                     short_feature_index: bool = False,
                     global_names: Optional[List[str]] = None) -> PartitionedData:
    """Group prepared rows by key (AdmmPartitioner, jobs/RegressionAdmmTrain.java:579-590) and
    index each partition like LibLinear[Binary]Dataset.addInstanceAvro/finish."""
    gindex: Dict[str, int] = {}
    gnames: List[str] = []
    if global_names is not None:
        for nme in global_names:
            gindex[nme] = len(gnames)
            gnames.append(nme)
    per: List[List[PreparedRow]] = [[] for _ in range(num_blocks)]
    for r in rows:
        k = int(r.key)
        if k < 0 or k >= num_blocks:
            raise RuntimeError("Map key is wrong! key has to be in the range of [0,numPartitions-1].")
        per[k].append(r)
        for name, term, _ in r.features:          # global ids in input-stream order (same as the native host)
            key = feature_key(name, term)
            if key not in gindex and key != INTERCEPT_NAME:
                gindex[key] = len(gnames)
                gnames.append(key)
    blocks: List[PartitionBlock] = []
    for k, prow in enumerate(per):
        findex: Dict[str, int] = {}
        lnames: List[str] = []
        row_ptr = [0]
        cols: List[int] = []
        vals: List[np.float32] = []
        y = np.empty(len(prow), dtype=np.int8)
        wt = np.empty(len(prow), dtype=np.float32)
        off = np.empty(len(prow), dtype=np.float32)
        for i, r in enumerate(prow):
            if r.response not in (1, 0, -1):
                raise ModelFittingError("response = %d (only 1, 0, -1 are allowed)" % r.response)
            y[i] = 1 if r.response == 1 else -1                           # 0 -> -1, :421-422
            if float(r.weight) < 0:
                raise ModelFittingError("weight = %s (weight cannot < 0)" % r.weight)
            wt[i] = r.weight
            off[i] = r.offset
            ent = []
            for name, term, value in r.features:
                key = feature_key(name, term)
                if binary_feature and float(value) != 1.0:
                    raise ModelFittingError("Cannot handle non-binary feature value")      # binary :475-477
                j = findex.get(key)
                if j is None:
                    if key == INTERCEPT_NAME:
                        raise ModelFittingError("feature name cannot be " + INTERCEPT_NAME)  # :470-471
                    j = len(lnames)
                    findex[key] = j
                    lnames.append(key)
                    if short_feature_index and j + 1 >= 32767:
                        raise ModelFittingError("short feature index overflow")            # binary :503-505
                    if key not in gindex:
                        gindex[key] = len(gnames)
                        gnames.append(key)
                ent.append((j, value))
            ent.sort(key=lambda e: e[0])                                  # per-row sort by id, :481-482 (stable)
            cols.extend(e[0] for e in ent)
            vals.extend(e[1] for e in ent)
            row_ptr.append(len(cols))
        blocks.append(PartitionBlock(
            partition_id=k, l=len(prow), n_local=len(lnames) + 1,
            row_ptr=np.asarray(row_ptr, dtype=np.int64), col_idx=np.asarray(cols, dtype=np.int32),
            val=None if binary_feature else np.asarray(vals, dtype=np.float32),
            y=y, weight=wt, offset=off,
            local_to_global=np.asarray([gindex[nme] for nme in lnames] + [-1], dtype=np.int32)))
    ng = len(gnames) + 1
    for b in blocks:
        b.local_to_global[-1] = ng - 1                                    # intercept is the last global index
    return PartitionedData(blocks, gnames, num_blocks)


def dense_partitions(X: np.ndarray, y01: np.ndarray, num_blocks: int, weight: Optional[np.ndarray] = None,
                     offset: Optional[np.ndarray] = None) -> PartitionedData:
    """Row i -> partition i % num_blocks (the deterministic stand-in of SURVEY 8d); every feature
    present in every row, so local ids == global ids. Returned as CSR blocks."""
    nrow, nf = X.shape
    blocks = []
    for k in range(num_blocks):
        sel = np.arange(k, nrow, num_blocks)
        l = len(sel)
        blocks.append(PartitionBlock(
            k, l, nf + 1, np.arange(0, (l + 1) * nf, nf, dtype=np.int64),
            np.tile(np.arange(nf, dtype=np.int32), l), np.ascontiguousarray(X[sel], dtype=np.float32).reshape(-1),
            np.where(y01[sel] == 1, 1, -1).astype(np.int8),
            (np.ones(l, np.float32) if weight is None else weight[sel].astype(np.float32)),
            (np.zeros(l, np.float32) if offset is None else offset[sel].astype(np.float32)),
            np.arange(nf + 1, dtype=np.int32)))
    return PartitionedData(blocks, [str(j + 1) for j in range(nf)], num_blocks)


@dataclass
class TestRows:
    """Test rows of the per-iteration test-loglik (jobs/RegressionAdmmTrain.java:766-811) in GLOBAL feature ids."""
    __test__ = False
    row_ptr: np.ndarray          # int64 [l+1]
    global_idx: np.ndarray       # int32 [nnz]; -1 = feature name not in the training dictionary (skipped by eval)
    val: Optional[np.ndarray]    # float64 [nnz] as Util.getDoubleAvro yields (NOT cast to float: models/LinearModel.java:530-534) or None (binary.feature -> 1.0)
    response: np.ndarray         # int8 [l] as read (1 / 0 / -1)
    weight: np.ndarray           # float64 [l]  Util.getDoubleAvro(record, "weight"), default 1
    offset: np.ndarray           # float64 [l]
    n: float                     # sum of Double.parseDouble(record.get("weight").toString()) (:792-798)


def build_test_rows(records: Iterable[Dict[str, Any]], feature_names: Sequence[str], binary_feature: bool = False,
                    max_rows: int = 1000000) -> TestRows:
    """RAW test records (same format as the training input) -> arrays; at most MAX_NTEST_EVENTS rows (:122,:799)."""
    from .admm import java_float_to_string
    index = {k: j for j, k in enumerate(feature_names)}
    rp, gi, vv, ys, ws, os_ = [0], [], [], [], [], []
    n = 0.0
    for rec in records:
        y = get_response(rec)
        if y not in (1, 0, -1):
            raise IOError("response = %d" % y)
        feats = rec.get("features")
        if feats is None:
            raise IOError("features is null")
        for f in feats:
            name = str(f["name"])
            term = "" if f.get("term") is None else str(f["term"])
            gi.append(index.get(feature_key(name, term), -1))
            if not binary_feature:
                vv.append(float(f["value"]))                 # evalInstanceAvro keeps the double (a float field widens exactly)
        rp.append(len(gi))
        ys.append(y)
        w = rec.get("weight")
        ws.append(1.0 if w is None else float(w))
        if w is None:
            n += 1.0
        elif isinstance(w, (int, np.integer)):
            n += float(int(w))
        else:       # a Float prints as the shortest float32 string, a Double as itself
            n += float(java_float_to_string(w)) if float(np.float32(w)) == float(w) else float(w)
        o = rec.get("offset")
        os_.append(0.0 if o is None else float(o))
        if len(ys) >= max_rows:
            break
    return TestRows(np.asarray(rp, np.int64), np.asarray(gi, np.int32), None if binary_feature else np.asarray(vv, np.float64),
                    np.asarray(ys, np.int8), np.asarray(ws, np.float64), np.asarray(os_, np.float64), n)
