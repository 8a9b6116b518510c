"""Minimal Avro object-container reader/writer (pure Python, no avro library).

Host-side plumbing for the drop-in formats of the ADMM path: the prepared
rows (``RegressionPrepareOutput.avsc``), raw Pig-style input rows with nullable
unions (``examples/sample-data.avro``) and the model files (``LinearModelAvro.avsc``,
``RegressionTrainOutput.avsc``, ``SampleTestLoglik.avsc``) written by
``utils/LinearModelUtils.java:39-66`` / ``models/LinearModel.java:697-720``.

Codecs: ``null`` and ``deflate`` (the reference writes deflate level 9,
``mapred/AbstractAvroJob.java:253``; the sample file is ``null``).
"""
from __future__ import annotations

import io
import json
import os
import struct
import zlib
from typing import Any, BinaryIO, Dict, Iterator, List, Tuple

MAGIC = b"Obj\x01"

# --------------------------------------------------------------------------- schemas
FEATURE_ITEMS = {
    "type": "record", "name": "feature", "fields": [
        {"name": "name", "type": "string"},
        {"name": "term", "type": "string"},
        {"name": "value", "type": "float"}]}

LINEAR_MODEL_SCHEMA = {  # avro/LinearModelAvro.avsc:16-31
    "type": "record", "doc": "Linear Model in Avro format", "name": "LinearModelAvro",
    "namespace": "com.linkedin.mlease.avro",
    "fields": [{"name": "key", "type": "string"},
               {"name": "model", "type": {"type": "array", "items": FEATURE_ITEMS}}]}

PREPARE_OUTPUT_SCHEMA = {  # avro/RegressionPrepareOutput.avsc:16-34
    "type": "record", "name": "RegressionPrepareOutput",
    "doc": "Output for RegressionPrepare job before running Regression train jobs",
    "namespace": "com.linkedin.mlease.regression.avro",
    "fields": [{"name": "key", "type": "string"},
               {"name": "response", "type": "int"},
               {"name": "features", "type": {"type": "array", "items": FEATURE_ITEMS}},
               {"name": "weight", "type": "float"},
               {"name": "offset", "type": "float"}]}

TRAIN_OUTPUT_SCHEMA = {  # avro/RegressionTrainOutput.avsc:17-39
    "type": "record", "name": "RegressionTrainOutput", "doc": "Model output from AdmmTrain",
    "namespace": "com.linkedin.mlease.regression.avro",
    "fields": [{"name": "key", "type": "string"},
               {"name": "model", "type": {"type": "array", "items": FEATURE_ITEMS}},
               {"name": "uplusx", "type": {"type": "array", "items": {
                   "type": "record", "name": "feature1", "fields": [
                       {"name": "name", "type": "string"},
                       {"name": "term", "type": "string"},
                       {"name": "value", "type": "float"}]}}}]}

SAMPLE_TEST_LOGLIK_SCHEMA = {  # avro/SampleTestLoglik.avsc:16-26
    "type": "record", "name": "SampleTestLoglik", "doc": "Sample test loglik over iterations",
    "namespace": "com.linkedin.mlease.regression.avro",
    "fields": [{"name": "lambda", "type": "string"},
               {"name": "iter", "type": "int"},
               {"name": "testLoglik", "type": "float"}]}

LAMBDA_RHO_SCHEMA = {  # avro/LambdaRhoMap.avsc:16-25
    "type": "record", "name": "LambdaRhoMap", "doc": "The map of lambda values to rho values",
    "namespace": "com.linkedin.mlease.regression.avro",
    "fields": [{"name": "lambda", "type": "float"}, {"name": "rho", "type": "float"}]}


# --------------------------------------------------------------------------- decoding
class _Reader:
    __slots__ = ("buf", "pos")

    def __init__(self, buf: bytes, pos: int = 0):
        self.buf = buf
        self.pos = pos

    def read(self, n: int) -> bytes:
        b = self.buf[self.pos:self.pos + n]
        if len(b) != n:
            raise EOFError("truncated avro data")
        self.pos += n
        return b

    def long(self) -> int:
        shift = 0
        acc = 0
        buf = self.buf
        pos = self.pos
        while True:
            b = buf[pos]
            pos += 1
            acc |= (b & 0x7F) << shift
            if not (b & 0x80):
                break
            shift += 7
        self.pos = pos
        return (acc >> 1) ^ -(acc & 1)


def _named_types(schema: Any, table: Dict[str, Any]) -> None:
    if isinstance(schema, dict):
        t = schema.get("type")
        if t in ("record", "enum", "fixed"):
            table[schema["name"]] = schema
            ns = schema.get("namespace")
            if ns:
                table[ns + "." + schema["name"]] = schema
        if t == "record":
            for f in schema["fields"]:
                _named_types(f["type"], table)
        elif t == "array":
            _named_types(schema["items"], table)
        elif t == "map":
            _named_types(schema["values"], table)
        elif isinstance(t, (dict, list)):
            _named_types(t, table)
    elif isinstance(schema, list):
        for s in schema:
            _named_types(s, table)


def _decode(r: _Reader, schema: Any, named: Dict[str, Any]) -> Any:
    if isinstance(schema, str):
        t = schema
        if t == "null":
            return None
        if t == "boolean":
            return r.read(1) != b"\x00"
        if t in ("int", "long"):
            return r.long()
        if t == "float":
            return struct.unpack("<f", r.read(4))[0]
        if t == "double":
            return struct.unpack("<d", r.read(8))[0]
        if t == "bytes":
            return r.read(r.long())
        if t == "string":
            return r.read(r.long()).decode("utf-8")
        return _decode(r, named[t], named)
    if isinstance(schema, list):                       # union
        return _decode(r, schema[r.long()], named)
    t = schema["type"]
    if t == "record":
        return {f["name"]: _decode(r, f["type"], named) for f in schema["fields"]}
    if t == "array":
        out: List[Any] = []
        items = schema["items"]
        while True:
            n = r.long()
            if n == 0:
                break
            if n < 0:
                n = -n
                r.long()                               # block byte size
            for _ in range(n):
                out.append(_decode(r, items, named))
        return out
    if t == "map":
        m: Dict[str, Any] = {}
        while True:
            n = r.long()
            if n == 0:
                break
            if n < 0:
                n = -n
                r.long()
            for _ in range(n):
                k = r.read(r.long()).decode("utf-8")
                m[k] = _decode(r, schema["values"], named)
        return m
    if t == "enum":
        return schema["symbols"][r.long()]
    if t == "fixed":
        return r.read(schema["size"])
    return _decode(r, t, named)                        # {"type": "string"} etc.


def read_container(path: str) -> Tuple[Any, Iterator[Dict[str, Any]]]:
    """Return (writer schema, iterator over records) of an avro object container file."""
    with open(path, "rb") as fh:
        data = fh.read()
    r = _Reader(data)
    if r.read(4) != MAGIC:
        raise IOError("%s is not an avro object container file" % path)
    meta = _decode(r, {"type": "map", "values": "bytes"}, {})
    sync = r.read(16)
    schema = json.loads(meta["avro.schema"].decode("utf-8"))
    codec = meta.get("avro.codec", b"null").decode("ascii")
    if codec not in ("null", "deflate"):
        raise IOError("unsupported avro codec %r" % codec)
    named: Dict[str, Any] = {}
    _named_types(schema, named)

    def records() -> Iterator[Dict[str, Any]]:
        while r.pos < len(data):
            count = r.long()
            size = r.long()
            block = r.read(size)
            if codec == "deflate":
                block = zlib.decompress(block, -15)
            br = _Reader(block)
            for _ in range(count):
                yield _decode(br, schema, named)
            if r.read(16) != sync:
                raise IOError("avro sync marker mismatch in %s" % path)

    return schema, records()


def read_records(path: str) -> List[Dict[str, Any]]:
    """All records of a file, or of every ``*.avro`` part file of a directory (sorted by name)."""
    if os.path.isdir(path):
        out: List[Dict[str, Any]] = []
        for name in sorted(os.listdir(path)):
            if name.endswith(".avro"):
                out.extend(read_container(os.path.join(path, name))[1])
        return out
    return list(read_container(path)[1])


# --------------------------------------------------------------------------- encoding
def _zigzag(n: int) -> bytes:
    n = (n << 1) ^ (n >> 63)
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _encode(w: BinaryIO, schema: Any, v: Any, named: Dict[str, Any]) -> None:
    if isinstance(schema, str):
        t = schema
        if t == "null":
            return
        if t == "boolean":
            w.write(b"\x01" if v else b"\x00")
        elif t in ("int", "long"):
            w.write(_zigzag(int(v)))
        elif t == "float":
            w.write(struct.pack("<f", v))
        elif t == "double":
            w.write(struct.pack("<d", v))
        elif t == "bytes":
            w.write(_zigzag(len(v)))
            w.write(v)
        elif t == "string":
            b = v.encode("utf-8")
            w.write(_zigzag(len(b)))
            w.write(b)
        else:
            _encode(w, named[t], v, named)
        return
    if isinstance(schema, list):
        for i, s in enumerate(schema):
            if (v is None) == (s == "null"):
                w.write(_zigzag(i))
                _encode(w, s, v, named)
                return
        raise ValueError("no union branch for %r" % (v,))
    t = schema["type"]
    if t == "record":
        for f in schema["fields"]:
            _encode(w, f["type"], v[f["name"]], named)
    elif t == "array":
        if len(v):
            w.write(_zigzag(len(v)))
            for x in v:
                _encode(w, schema["items"], x, named)
        w.write(b"\x00")
    elif t == "map":
        if len(v):
            w.write(_zigzag(len(v)))
            for k, x in v.items():
                _encode(w, "string", k, named)
                _encode(w, schema["values"], x, named)
        w.write(b"\x00")
    else:
        _encode(w, t, v, named)


def write_container(path: str, schema: Any, records: List[Dict[str, Any]], codec: str = "deflate",
                    block_records: int = 4096, sync: bytes = b"mlease-amd-sync!") -> None:
    """Write an avro object container file (deflate level 9 like mapred/AbstractAvroJob.java:253)."""
    named: Dict[str, Any] = {}
    _named_types(schema, named)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        meta = {"avro.schema": json.dumps(schema).encode("utf-8"), "avro.codec": codec.encode("ascii")}
        _encode(fh, {"type": "map", "values": "bytes"}, meta, {})
        fh.write(sync)
        for b in range(0, len(records), block_records):
            chunk = records[b:b + block_records]
            bio = io.BytesIO()
            for rec in chunk:
                _encode(bio, schema, rec, named)
            raw = bio.getvalue()
            if codec == "deflate":
                c = zlib.compressobj(9, zlib.DEFLATED, -15)
                raw = c.compress(raw) + c.flush()
            fh.write(_zigzag(len(chunk)))
            fh.write(_zigzag(len(raw)))
            fh.write(raw)
            fh.write(sync)
