"""Native host (ml-ease_amd/host: C++ avro reader, RegressionPrepare + LibLinearDataset indexing, .job parsing, CLI)
checked against the Python mirror and, end to end on the GPU, against the committed C1 golden."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import mlease_amd  # noqa: F401
from mlease_amd import admm, avro_io, dataset
from fixtures import load_c1, load_c1_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ml-ease_amd", "host")

PIG_SCHEMA = {  # the writer schema shape of examples/sample-data.avro (Pig: everything nullable) + a partition key field
    "type": "record", "name": "TUPLE_0", "fields": [
        {"name": "features", "type": ["null", {"type": "array", "items": ["null", {
            "type": "record", "name": "TUPLE_1", "fields": [
                {"name": "name", "type": ["null", "string"]}, {"name": "term", "type": ["null", "string"]},
                {"name": "value", "type": ["null", "float"]}]}]}]},
        {"name": "offset", "type": ["null", "int"]}, {"name": "response", "type": ["null", "int"]},
        {"name": "weight", "type": ["null", "int"]}, {"name": "pkey", "type": ["null", "int"]},
        {"name": "unused", "type": ["null", {"type": "map", "values": "double"}]}]}


def lib():
    name = "libmlease_host_asan.so" if os.environ.get("MLX_ASAN", "0") not in ("", "0") else "libmlease_host.so"    # tools/run_asan.sh
    subprocess.check_call(["make", "-C", HOST, "-s", name])
    L = C.CDLL(os.path.join(HOST, name))
    L.mlh_last_error.restype = C.c_char_p
    L.mlh_build.restype = C.c_void_p
    L.mlh_build.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_ulonglong, C.c_int, C.c_int]
    L.mlh_free.argtypes = [C.c_void_p]
    L.mlh_n_global.argtypes = [C.c_void_p]
    L.mlh_feature_name.restype = C.c_char_p
    L.mlh_feature_name.argtypes = [C.c_void_p, C.c_int]
    L.mlh_part_sizes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    for nme, ty in (("rowptr", C.c_longlong), ("col", C.c_int), ("val", C.c_float), ("y", C.c_byte), ("weight", C.c_float),
                    ("offset", C.c_float), ("l2g", C.c_int)):
        f = getattr(L, "mlh_part_" + nme)
        f.restype = C.POINTER(ty)
        f.argtypes = [C.c_void_p, C.c_int]
    L.mlh_test_rows.restype = C.c_void_p
    L.mlh_test_rows.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_longlong]
    L.mlh_test_free.argtypes = [C.c_void_p]
    L.mlh_test_sizes.restype = C.c_double
    L.mlh_test_sizes.argtypes = [C.c_void_p, C.c_void_p]
    for nme, ty in (("rowptr", C.c_longlong), ("gidx", C.c_int), ("val", C.c_double), ("response", C.c_byte),
                    ("weight", C.c_double), ("offset", C.c_double)):
        f = getattr(L, "mlh_test_" + nme)
        f.restype = C.POINTER(ty)
        f.argtypes = [C.c_void_p]
    L.mlh_float_to_string.argtypes = [C.c_float, C.c_char_p, C.c_int]
    L.mlh_float_string_roundtrip.restype = C.c_double
    L.mlh_float_string_roundtrip.argtypes = [C.c_float]
    return L


def native_blocks(L, h, nb):
    out = []
    for k in range(nb):
        sz = (C.c_longlong * 3)()
        L.mlh_part_sizes(h, k, sz)
        l, nloc, nnz = [int(x) for x in sz]
        arr = lambda f, n, dt: np.ctypeslib.as_array(f(h, k), shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)
        vp = L.mlh_part_val(h, k)
        out.append(dict(l=l, n_local=nloc, row_ptr=arr(L.mlh_part_rowptr, l + 1, np.int64), col=arr(L.mlh_part_col, nnz, np.int32),
                        val=(np.ctypeslib.as_array(vp, shape=(nnz,)).copy() if vp and nnz else None),
                        y=arr(L.mlh_part_y, l, np.int8), weight=arr(L.mlh_part_weight, l, np.float32),
                        offset=arr(L.mlh_part_offset, l, np.float32), l2g=arr(L.mlh_part_l2g, nloc, np.int32)))
    return out


from fixtures import c1_raw_records  # noqa: E402


@pytest.fixture(scope="module")
def c1():
    return load_c1()


def test_native_indexing_equals_python_mirror(tmp_path, c1):
    L = lib()
    recs = c1_raw_records(c1)
    p = str(tmp_path / "raw" / "part-00000.avro")
    avro_io.write_container(p, PIG_SCHEMA, recs[:600], codec="deflate", block_records=97)
    avro_io.write_container(str(tmp_path / "raw" / "part-00001.avro"), PIG_SCHEMA, recs[600:], codec="null")
    h = L.mlh_build(str(tmp_path / "raw").encode(), 8, b"pkey", 0, 1, 0, 0, 0)
    assert h, L.mlh_last_error()
    nat = native_blocks(L, h, 8)
    rows = dataset.prepare_rows(avro_io.read_records(str(tmp_path / "raw")), 8, map_key="pkey")
    py = dataset.build_partitions(rows, 8)
    assert L.mlh_n_global(h) == py.n_global
    assert [L.mlh_feature_name(h, j).decode() for j in range(py.n_global - 1)] == py.feature_names
    for a, b in zip(nat, py.blocks):
        assert (a["l"], a["n_local"]) == (b.l, b.n_local)
        for k, v in (("row_ptr", b.row_ptr), ("col", b.col_idx), ("val", b.val), ("y", b.y), ("weight", b.weight),
                     ("offset", b.offset), ("l2g", b.local_to_global)):
            assert np.array_equal(a[k], v), k
    # test rows: first file only, unknown features -> -1, n = sum of weights as strings
    t = L.mlh_test_rows(p.encode(), h, 0, 250)
    assert t, L.mlh_last_error()
    sz = (C.c_longlong * 2)()
    n = L.mlh_test_sizes(t, sz)
    tr = dataset.build_test_rows(recs[:250], py.feature_names)
    assert int(sz[0]) == 250 and n == tr.n
    assert np.array_equal(np.ctypeslib.as_array(L.mlh_test_gidx(t), shape=(int(sz[1]),)), tr.global_idx)
    assert np.array_equal(np.ctypeslib.as_array(L.mlh_test_val(t), shape=(int(sz[1]),)), tr.val)
    L.mlh_test_free(t)
    L.mlh_free(h)


def test_native_prepare_semantics(tmp_path):
    L = lib()
    schema = {"type": "record", "name": "R", "fields": [
        {"name": "response", "type": "int"}, {"name": "click", "type": ["null", "boolean"]},
        {"name": "weight", "type": ["null", "double"]}, {"name": "offset", "type": ["null", "float"]},
        {"name": "features", "type": {"type": "array", "items": {"type": "record", "name": "F", "fields": [
            {"name": "name", "type": "string"}, {"name": "term", "type": ["null", "string"]}, {"name": "value", "type": "double"}]}}}]}
    recs = [{"response": 1, "click": None, "weight": 4.0, "offset": 1.5, "features": [{"name": "b", "term": None, "value": 0.1}, {"name": "a", "term": "t", "value": 2.0}]},
            {"response": 0, "click": True, "weight": None, "offset": None, "features": [{"name": "a", "term": "t", "value": -1.0}]},
            {"response": 1, "click": None, "weight": 1.0, "offset": None, "features": []}]
    p = str(tmp_path / "in.avro")
    avro_io.write_container(p, schema, recs, codec="null")
    # no map.key: seeded random key; positives divided by / replicated over num.click.replicates consecutive partitions
    h = L.mlh_build(p.encode(), 4, b"", 0, 2, 7, 0, 0)
    assert h, L.mlh_last_error()
    nat = native_blocks(L, h, 4)
    assert sum(b["l"] for b in nat) == 2 + 1 + 2          # two positives x2 replicas, one negative once
    pos_parts = [k for k, b in enumerate(nat) for w in b["weight"] if w == np.float32(2.0)]
    assert len(pos_parts) == 2 and (pos_parts[1] - pos_parts[0]) % 4 in (1, 3)        # consecutive (with wrap)
    for b in nat:
        for i in range(b["l"]):
            seg = b["col"][b["row_ptr"][i]:b["row_ptr"][i + 1]]
            assert np.all(np.diff(seg) > 0)
    names = [L.mlh_feature_name(h, j).decode() for j in range(L.mlh_n_global(h) - 1)]
    assert set(names) == {"b", "a" + dataset.TERM_SEP + "t"}
    L.mlh_free(h)
    # binary.feature ignores values; a prepared file written by the Python mirror reads back identically
    h = L.mlh_build(p.encode(), 4, b"", 1, 1, 7, 0, 0)
    assert h and native_blocks(L, h, 4)[0]["val"] is None
    L.mlh_free(h)
    for bad, msg in ((dict(recs[0], response=2), b"only 1, 0, -1"), (dict(recs[0], weight=-1.0), b"weight cannot")):
        avro_io.write_container(p, schema, [bad], codec="null")
        assert not L.mlh_build(p.encode(), 4, b"", 0, 1, 0, 0, 0) and msg in L.mlh_last_error()
    avro_io.write_container(p, schema, recs, codec="null")
    assert not L.mlh_build(p.encode(), 4, b"nokey", 0, 1, 0, 0, 0) and b"map.key is wrongly specified" in L.mlh_last_error()


def test_native_prepared_input_and_java_strings(tmp_path, c1):
    L = lib()
    rows = dataset.prepare_rows(c1_raw_records(c1), 8, map_key="pkey")
    p = str(tmp_path / "tmp-data" / "part-00000.avro")
    avro_io.write_container(p, avro_io.PREPARE_OUTPUT_SCHEMA, [r.to_avro() for r in rows])
    h = L.mlh_build(p.encode(), 8, b"", 0, 1, 0, 1, 0)
    assert h, L.mlh_last_error()
    nat = native_blocks(L, h, 8)
    py = dataset.build_partitions(rows, 8)
    for a, b in zip(nat, py.blocks):
        assert np.array_equal(a["col"], b.col_idx) and np.array_equal(a["val"], b.val) and np.array_equal(a["y"], b.y)
    L.mlh_free(h)
    buf = C.create_string_buffer(64)
    e = np.float32(0.01)
    for _ in range(60):
        L.mlh_float_to_string(float(e), buf, 64)
        assert buf.value.decode() == admm.java_float_to_string(e)
        assert L.mlh_float_string_roundtrip(float(e)) == admm.float_string_roundtrip(e)
        e = np.float32(e / np.float32(10))
    for v in (1.0, 10.0, 100.0, 1000.0, 0.3, 1e7, 123456.7, 5e-5):
        L.mlh_float_to_string(v, buf, 64)
        assert buf.value.decode() == admm.java_float_to_string(v)


@pytest.mark.skipif(not os.path.exists("/root/reference/examples/sample-data.avro"), reason="reference checkout not present")
def test_native_reads_the_reference_sample_file():
    L = lib()
    h = L.mlh_build(b"/root/reference/examples/sample-data.avro", 8, b"", 0, 1, 3, 0, 0)
    assert h, L.mlh_last_error()
    nat = native_blocks(L, h, 8)
    assert sum(b["l"] for b in nat) == 1000 + 299 * 0 and L.mlh_n_global(h) == 201      # 1000 rows, 200 features
    assert sum(len(b["col"]) for b in nat) == 100326
    L.mlh_free(h)


@pytest.mark.gpu
def test_cli_end_to_end_matches_golden(tmp_path, c1):
    """`mlease_admm_train sample.job` on the C1 rows (raw Pig-style avro, map.key) -> final-model avro equals the
    committed golden z of iteration 20 (float32), sample-test-loglik files written."""
    subprocess.check_call(["make", "-C", HOST, "-s"])
    recs = c1_raw_records(c1)
    avro_io.write_container(str(tmp_path / "in" / "part-00000.avro"), PIG_SCHEMA, recs, codec="deflate")
    avro_io.write_container(str(tmp_path / "test" / "part-00000.avro"), PIG_SCHEMA, recs[:200], codec="null")
    job = tmp_path / "sample.job"
    job.write_text("# sample\ninput.paths=%s\noutput.base.path=%s\ntest.path=%s\nnum.blocks=8\nlambda=1.0\nnum.iters=20\n"
                   "regularizer=2\nmap.key=pkey\nforce.output.overwrite=true\n" % (tmp_path / "in", tmp_path / "out", tmp_path / "test"))
    r = subprocess.run([os.path.join(HOST, "mlease_admm_train"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    gold = load_c1_golden()
    models = admm.read_linear_models(str(tmp_path / "out" / "final-model" / "part-r-00000.avro"), c1.feature_names)
    assert list(models) == ["1.0"]
    want = gold["Z"][-1][0].astype(np.float32)
    err = np.abs(models["1.0"].astype(np.float64) - want) / np.maximum(np.abs(want), 1e-2 * np.max(np.abs(want)))
    assert np.max(err) <= 1e-5
    ll = avro_io.read_records(str(tmp_path / "out" / "sample-test-loglik" / "iteration-20.avro"))
    assert ll[0]["lambda"] == "1.0" and ll[0]["iter"] == 20 and -1.0 < ll[0]["testLoglik"] < 0.0
    assert os.path.isdir(tmp_path / "out" / "best-model") and os.path.exists(tmp_path / "out" / "lambda-rho" / "part-r-00000.avro")
    # updateLogLikBestModel deletes the directory before every write (jobs/RegressionAdmmTrain.java:838-840): ONE file, the
    # iteration with the best test loglik; and a second run into the same output directory leaves no stale files behind
    best = sorted(os.listdir(tmp_path / "out" / "best-model"))
    lls = [avro_io.read_records(str(tmp_path / "out" / "sample-test-loglik" / ("iteration-%d.avro" % i)))[0]["testLoglik"] for i in range(1, 21)]
    assert best == ["best-iteration-%d.avro" % (int(np.argmax(np.asarray(lls, np.float32))) + 1)], (best, lls)
    (tmp_path / "out" / "best-model" / "best-iteration-99.avro").write_bytes(b"stale")
    (tmp_path / "out" / "sample-test-loglik" / "iteration-77.avro").write_bytes(b"stale")
    r2 = subprocess.run(r.args, capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert sorted(os.listdir(tmp_path / "out" / "best-model")) == best
    assert not os.path.exists(tmp_path / "out" / "sample-test-loglik" / "iteration-77.avro")


@pytest.mark.gpu
def test_cli_job_key_selects_the_reference_order_numerics(tmp_path, c1):
    """mlease.numerics=reference_order in the .job file (-> mlx_set_option before the partitions are added): the CLI's final-model is
    bit-identical to the oracle twin's 5-iteration run of the same job (portable exp / log1p on both sides); an unknown value fails
    the job with the library's message."""
    import oracle_lib as ol
    subprocess.check_call(["make", "-C", HOST, "-s"])
    recs = c1_raw_records(c1)
    avro_io.write_container(str(tmp_path / "in" / "part-00000.avro"), PIG_SCHEMA, recs, codec="deflate")
    job = tmp_path / "ro.job"
    text = ("input.paths=%s\noutput.base.path=%s\nnum.blocks=8\nlambda=1.0\nnum.iters=5\nregularizer=2\nmap.key=pkey\n"
            "force.output.overwrite=true\nmlease.numerics=%%s\n" % (tmp_path / "in", tmp_path / "out"))
    job.write_text(text % "reference_order")
    r = subprocess.run([os.path.join(HOST, "mlease_admm_train"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    models = admm.read_linear_models(str(tmp_path / "out" / "final-model" / "part-r-00000.avro"), c1.feature_names)
    oc = ol.OracleAdmm(c1.blocks, c1.n_global, [1.0], [1.0], pm=True)
    for it in range(5):
        oc.iterate(0.01, 1.0, nthreads=2)
    assert np.array_equal(models["1.0"].astype(np.float32), oc.z()[1][0]), "reference-order job: final-model differs from the oracle twin"
    # round 6: the run says which contract produced the model -- in the log and in <out>/_mlease_run.json (hidden from Hadoop listings)
    import json
    meta = json.loads((tmp_path / "out" / "_mlease_run.json").read_text())
    assert meta["numerics"] == "reference_order" and meta["numerics_kernels"].startswith("reference_order") and meta["admm_iterations"] == 5
    assert "numerics contract: reference_order" in r.stderr
    job.write_text((text % "fast").replace("mlease.numerics=fast\n", ""))          # no key: the drop-in's default is the reference-order contract (round 6)
    r = subprocess.run([os.path.join(HOST, "mlease_admm_train"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and json.loads((tmp_path / "out" / "_mlease_run.json").read_text())["numerics"] == "reference_order"
    assert "numerics contract: reference_order" in r.stderr and "mlease.numerics=fast" in r.stderr
    models = admm.read_linear_models(str(tmp_path / "out" / "final-model" / "part-r-00000.avro"), c1.feature_names)
    assert np.array_equal(models["1.0"].astype(np.float32), oc.z()[1][0]), "default job: final-model differs from the oracle twin"
    job.write_text(text % "fast")                                                   # the faster contract, on request
    r = subprocess.run([os.path.join(HOST, "mlease_admm_train"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and json.loads((tmp_path / "out" / "_mlease_run.json").read_text())["numerics"] == "fast"
    assert "numerics contract: fast" in r.stderr and "mlease.numerics=reference_order" in r.stderr
    job.write_text(text % "fastest")
    r = subprocess.run([os.path.join(HOST, "mlease_admm_train"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "numerics must be" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_cli_writes_the_iteration_files(tmp_path, c1):
    """write.iter.files=true: iter-<i>/{u, init-value, model} as the reference leaves them (jobs/RegressionAdmmTrain.java:309-334,
    reducer output avro/RegressionTrainOutput.avsc:17-39; keys "<lambda>" and "<lambda>#<partition>") -- checked against the committed
    oracle golden of the same job: z entering iteration i, the reducers' beta_k / u_k + beta_k of iterations 1 and 2, and u_k."""
    subprocess.check_call(["make", "-C", HOST, "-s"])
    recs = c1_raw_records(c1)
    avro_io.write_container(str(tmp_path / "in" / "part-00000.avro"), PIG_SCHEMA, recs, codec="deflate")
    job = tmp_path / "iter.job"
    job.write_text("input.paths=%s\noutput.base.path=%s\nnum.blocks=8\nlambda=1.0\nnum.iters=3\nregularizer=2\nmap.key=pkey\n"
                   "write.iter.files=true\n" % (tmp_path / "in", tmp_path / "out"))
    r = subprocess.run([os.path.join(HOST, "mlease_admm_train"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    gold = load_c1_golden()
    names = c1.feature_names

    def close(a, want, what):
        a, want = a.astype(np.float64), want.astype(np.float64)
        err = np.abs(a - want) / np.maximum(np.abs(want), 1e-4 * np.max(np.abs(want)) + 1e-300)
        assert np.max(err) <= 1e-5, (what, float(np.max(err)))

    out = tmp_path / "out"
    assert avro_io.read_records(str(out / "iter-1" / "u")) == [] and avro_io.read_records(str(out / "iter-1" / "init-value")) == []
    for i in (2, 3):
        z = admm.read_linear_models(str(out / ("iter-%d" % i) / "init-value" / "part-r-00000.avro"), names)
        assert list(z) == ["1.0"]
        close(z["1.0"], gold["Z"][i - 2][0].astype(np.float32), "init-value of iteration %d" % i)
    for i in (1, 2):
        recs_m = {r_["key"]: r_ for r_ in avro_io.read_records(str(out / ("iter-%d" % i) / "model"))}
        assert sorted(recs_m) == sorted("1.0#%d" % k for k in range(8))
        u_next = admm.read_linear_models(str(out / ("iter-%d" % (i + 1)) / "u" / "part-r-00000.avro"), names)
        for k in range(8):
            for field, gkey in (("model", "B_it%d" % i), ("uplusx", "UPX_it%d" % i)):
                v = np.zeros(len(names) + 1, np.float32)
                idx = {n: j for j, n in enumerate(names)}
                for f in recs_m["1.0#%d" % k][field]:
                    v[-1 if f["name"] == admm.INTERCEPT_NAME else idx[f["name"]]] = np.float32(f["value"])
                close(v, gold[gkey][k, 0], "%s of partition %d, iteration %d" % (field, k, i))
            close(u_next["1.0#%d" % k], gold["Unext_it%d" % i][k, 0], "u of partition %d entering iteration %d" % (k, i + 1))


@pytest.mark.gpu
def test_cli_mean_model_warm_start(tmp_path, c1):
    """initialize.boost.rate in the job file: CLI (native host + HIP) == the Python driver loop over the oracle."""
    subprocess.check_call(["make", "-C", HOST, "-s"])
    recs = c1_raw_records(c1)
    avro_io.write_container(str(tmp_path / "in" / "part-00000.avro"), PIG_SCHEMA, recs, codec="deflate")
    avro_io.write_container(str(tmp_path / "test" / "part-00000.avro"), PIG_SCHEMA, recs[:200], codec="null")
    job = tmp_path / "boost.job"
    job.write_text("input.paths=%s\noutput.base.path=%s\ntest.path=%s\nnum.blocks=8\nlambda=1.0,10\nnum.iters=3\nregularizer=2\n"
                   "map.key=pkey\ninitialize.boost.rate=2.0\nliblinear.epsilon=0.001\n" % (tmp_path / "in", tmp_path / "out", tmp_path / "test"))
    r = subprocess.run([os.path.join(HOST, "mlease_admm_train"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "mean model initialised" in r.stderr
    models = admm.read_linear_models(str(tmp_path / "out" / "final-model" / "part-r-00000.avro"), c1.feature_names)
    cfg = admm.AdmmConfig.from_properties(admm.parse_job_file(str(job)))
    from engines import OracleEngine
    lam, rho = cfg.sorted_lambda_rho()
    eng = OracleEngine(c1.blocks, c1.n_global, lam, rho, 8)
    tr = admm.AdmmTrain(cfg, eng)
    tr.run()
    for key, want in tr.final_models().items():
        err = np.abs(models[key].astype(np.float64) - want) / np.maximum(np.abs(want), 1e-2 * np.max(np.abs(want)))
        assert np.max(err) <= 1e-5, key
    ll0 = avro_io.read_records(str(tmp_path / "out" / "sample-test-loglik" / "iteration-0.avro"))
    assert [x["iter"] for x in ll0] == [0, 0] and not os.path.exists(tmp_path / "out" / "best-model" / "best-iteration-0.avro")


@pytest.mark.gpu
def test_cli_several_handles_in_one_process(tmp_path, c1):
    """`gpus=0,0,0` with MLX_COMM_LOCAL=1: mlease_admm_train's multi-device logic -- partition k -> handle k mod G, one thread per
    handle for comm init / warm start / every iteration, the exchange inside mlx_naive_init and mlx_admm_iterate, test
    loglik and z from handle 0 -- runs on ONE GPU (the in-process communicator stands in for RCCL, which refuses two ranks on
    one device). The models must equal the one-handle run's (the partial sums associate differently: 1e-5, not bits)."""
    subprocess.check_call(["make", "-C", HOST, "-s"])
    recs = c1_raw_records(c1)
    avro_io.write_container(str(tmp_path / "in" / "part-00000.avro"), PIG_SCHEMA, recs, codec="deflate")
    avro_io.write_container(str(tmp_path / "test" / "part-00000.avro"), PIG_SCHEMA, recs[:200], codec="null")
    outs = {}
    for tag, gpus in (("one", "0"), ("three", "0,0,0")):
        job = tmp_path / (tag + ".job")
        job.write_text("input.paths=%s\noutput.base.path=%s\ntest.path=%s\nnum.blocks=8\nlambda=1.0,10\nnum.iters=6\nregularizer=2\n"
                       "map.key=pkey\ninitialize.boost.rate=2.0\ngpus=%s\n" % (tmp_path / "in", tmp_path / ("out_" + tag), tmp_path / "test", gpus))
        # the in-process communicator is experimental-build code: the CLI is linked against libmlease_hip.so, so the
        # experimental build under that name (csrc/exp/) is put first on the loader's path for this run
        env = dict(os.environ, MLX_COMM_LOCAL="1", LD_LIBRARY_PATH=os.path.join(os.path.dirname(HOST), "csrc", "exp") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([os.path.join(HOST, "mlease_admm_train"), str(job)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = (admm.read_linear_models(str(tmp_path / ("out_" + tag) / "final-model" / "part-r-00000.avro"), c1.feature_names),
                     [avro_io.read_records(str(tmp_path / ("out_" + tag) / "sample-test-loglik" / ("iteration-%d.avro" % i))) for i in range(0, 7)],
                     r.stderr)
    assert list(outs["one"][0]) == list(outs["three"][0]) == ["1.0", "10.0"]
    for key, want in outs["one"][0].items():
        got = outs["three"][0][key].astype(np.float64)
        err = np.abs(got - want) / np.maximum(np.abs(want), 1e-2 * np.max(np.abs(want)))
        assert np.max(err) <= 1e-5, (key, float(np.max(err)))
    for a, b in zip(outs["one"][1], outs["three"][1]):
        assert [x["lambda"] for x in a] == [x["lambda"] for x in b]
        assert np.allclose([x["testLoglik"] for x in a], [x["testLoglik"] for x in b], rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_cli_regression_test_matches_python_mirror(tmp_path, c1):
    """`mlease_regression_test job` (native host + mlx_score_rows) == admm.regression_test over the oracle scorer: same
    files, same schema, same float32 predictions in the same order; input records copied through unchanged."""
    from engines import OracleScorer
    subprocess.check_call(["make", "-C", HOST, "-s"])
    recs = c1_raw_records(c1)[:400]
    recs[3]["features"].append({"name": "never-seen", "term": "", "value": 2.5})
    recs[9]["offset"] = -0.5
    avro_io.write_container(str(tmp_path / "test" / "part-00000.avro"), PIG_SCHEMA, recs[:250], codec="deflate")
    avro_io.write_container(str(tmp_path / "test" / "part-00001.avro"), PIG_SCHEMA, recs[250:], codec="null")
    rng = np.random.default_rng(2)
    models = {"1.0": rng.normal(0, 0.3, c1.n_global).astype(np.float32), "0.5": rng.normal(0, 0.2, c1.n_global).astype(np.float32)}
    admm.write_linear_models(str(tmp_path / "model" / "final-model" / "part-r-00000.avro"), models, c1.feature_names)
    admm.write_linear_models(str(tmp_path / "model" / "best-model" / "best-iteration-2.avro"), {"0.5": models["0.5"]}, c1.feature_names)
    job = tmp_path / "test.job"
    job.write_text("input.paths=%s\noutput.base.path=%s\nmodel.base.path=%s\nlambda=1,0.5\n" % (tmp_path / "test", tmp_path / "out", tmp_path / "model"))
    r = subprocess.run([os.path.join(HOST, "mlease_regression_test"), str(job)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    props = admm.parse_job_file(str(job))
    props["output.base.path"] = str(tmp_path / "pyout")
    written = admm.regression_test(props, OracleScorer())
    assert len(written) == 3
    for w in written:
        rel = os.path.relpath(w, tmp_path / "pyout")
        s1, it1 = avro_io.read_container(str(tmp_path / "out" / rel))
        s2, it2 = avro_io.read_container(w)
        a, b = list(it1), list(it2)
        assert s1["name"] == s2["name"] == "AdmmTestOutput" and [f["name"] for f in s1["fields"]] == [f["name"] for f in s2["fields"]]
        assert len(a) == len(b) == 400
        assert np.array_equal(np.array([x["pred"] for x in a], np.float32), np.array([x["pred"] for x in b], np.float32))
        assert a == b                                            # every input field survived the byte-level copy
