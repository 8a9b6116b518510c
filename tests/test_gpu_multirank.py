"""N > 1 control flow on ONE GPU (-m gpu): what the first real 8-GPU run must not be the first execution of.

  * bench.py under torch.distributed.run with 2 ranks sharing device 0 (MLX_BENCH_SHARE_GPU=1: collectives over gloo through
    host staging) -- sharding k -> rank k mod N, the [xbar | ubar] exchange, the max / sum reductions of the report, the
    sparse configs[3] leg and the lambda-sweep leg -- against the one-rank run of the same job, bit for bit on the dense job;
  * the library's own exchange between several handles (what the Java host / the CLI's `gpus=0,1,..` use: mlx_comm_init +
    mlx_admm_iterate), two handles on two threads over the in-process communicator of the experimental build, including a
    rank whose solve fails (the status slot of the all-reduce: jobs/RegressionAdmmTrain.java:355-364 fails the whole job
    when any reducer throws; utils/LinearModelUtils.java:77-84 refuses a mean over fewer models than num.blocks).
"""
import copy
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

import mlease_amd  # noqa: F401
from mlease_amd import dataset
from mlease_amd.hip_engine import HipAdmmEngine
from fixtures import load_c1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(cmd, env, tmp_path, tag):
    """Runs bench.py; returns its FULL record (--full-json) after checking what the driver sees: exactly one stdout line, compact
    (< 4 KB), strict JSON, carrying the same headline as the full record."""
    full_path = str(tmp_path / ("bench_full_%s.json" % tag))
    r = subprocess.run(cmd + ["--full-json", full_path], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-500:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0].encode()) < 4096, (len(lines), len(lines[-1]))
    line = json.loads(lines[0])
    full = json.loads(open(full_path).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype"):
        assert line[k] == full[k], k
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["sparse"]["value"] == full["sparse"]["value"]
    return full


def test_bench_two_ranks_on_one_gpu_reproduce_the_one_rank_consensus(tmp_path):
    flags = ["--steps", "3", "--warmup", "1", "--rows", "65536", "--partitions", "8", "--no-cpu-baseline", "--no-gram", "--loglik-iters", "3",
             "--test-rows", "4096", "--sparse-rows", "160000", "--sparse-partitions", "8", "--sparse-steps", "2", "--sparse-warmup", "1",
             "--sparse-cpu-sample", "0", "--sweep-partitions", "2", "--sweep-steps", "1", "--sweep-warmup", "1", "--sweep-cpu-sample", "0"]
    # (no chunking pinned: the work of a pass workgroup adapts to what a handle holds, but the partial sums are per 256-row unit /
    # 64-row group and added in unit order, so one rank with 8 partitions and two ranks with 4 each produce the same bits -- DESIGN 8)
    env = dict(os.environ)
    env.pop("MLX_BENCH_SHARE_GPU", None)
    env.pop("MLX_DENSE_RPB", None)
    one = _bench([sys.executable, "bench.py"] + flags, env, tmp_path, "one")
    env2 = dict(env, MLX_BENCH_SHARE_GPU="1")
    two = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + flags, env2, tmp_path, "two")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and "test_mode" in two
    assert two["config"]["partitions"] == 8 and two["config"]["partitions_per_gpu"] == 4 and two["scaling"] == "strong"
    assert two["work"]["solves"] == one["work"]["solves"] == 24
    assert two["work"]["last_maxdiff"] == one["work"]["last_maxdiff"]
    assert two["work"]["z32_sha1_after_timed_steps"] == one["work"]["z32_sha1_after_timed_steps"]
    assert two["time_to_ref_loglik"]["loglik_by_iteration"] == one["time_to_ref_loglik"]["loglik_by_iteration"]
    # the sparse legs: same job sharded (the consensus sum associates differently and one-hot solves amplify that)
    for leg, nprob in (("sparse", 8 * 2), ("lambda_sweep", None)):
        assert two[leg]["n_gpus"] == 2 and two[leg]["value"] > 0 and one[leg]["value"] > 0
    assert abs(two["sparse"]["last_maxdiff"] - one["sparse"]["last_maxdiff"]) <= 0.05 * abs(one["sparse"]["last_maxdiff"])
    assert two["lambda_sweep"]["problems_per_gpu"] == 2 * 8


def _run_ranks(engs, fn):
    out, err = [None] * len(engs), [None] * len(engs)

    def work(r):
        try:
            out[r] = fn(r, engs[r])
        except Exception as e:          # noqa: BLE001 -- handed to the asserting thread
            err[r] = e

    ts = [threading.Thread(target=work, args=(r,)) for r in range(len(engs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a rank is blocked in the exchange"
    return out, err


def _two_handles(blocks, n_global, lam, rho, monkeypatch):
    monkeypatch.setenv("MLX_EXPERIMENTAL", "1")        # libmlease_hip_exp.so: the in-process communicator is not product code
    monkeypatch.setenv("MLX_COMM_LOCAL", "1")
    engs = []
    for r in range(2):
        e = HipAdmmEngine(n_global, lam, rho, len(blocks))
        for b in blocks[r::2]:
            e.add_partition(b)                          # partition k -> handle k mod 2
        e.finalize()
        engs.append(e)
    uid = HipAdmmEngine.comm_unique_id()
    _, err = _run_ranks(engs, lambda r, e: e.comm_init(uid, 2, r))
    assert err == [None, None], err
    return engs


def test_two_handles_exchange_inside_the_library(monkeypatch):
    c1 = load_c1()
    lam, rho = [1.0, 10.0], [1.0, 1.0]
    ref = HipAdmmEngine(c1.n_global, lam, rho, 8)
    for b in c1.blocks:
        ref.add_partition(b)
    ref.finalize()
    engs = _two_handles(c1.blocks, c1.n_global, lam, rho, monkeypatch)
    ref.naive_init(0.01)
    _, err = _run_ranks(engs, lambda r, e: e.naive_init(0.01))
    assert err == [None, None], err
    for it in range(4):
        sref = ref.iterate(0.01)
        out, err = _run_ranks(engs, lambda r, e: e.iterate(0.01))
        assert err == [None, None], err
        z0, z1, zr = engs[0].z()[0], engs[1].z()[0], ref.z()[0]
        assert np.array_equal(z0, z1), "iteration %d: the two ranks hold different consensus models" % (it + 1)
        # per-handle partial means summed in rank order vs one handle's sum over all 8: association only
        assert np.max(np.abs(z0 - zr)) <= 1e-5 * np.max(np.abs(zr))
        assert abs(out[0].maxdiff - sref.maxdiff) <= 1e-9 and out[0].maxdiff == out[1].maxdiff
        assert out[0].solves + out[1].solves == sref.solves == 16
    for e in engs + [ref]:
        e.close()


def test_two_handles_failed_rank_fails_both(monkeypatch):
    """The partition with a NaN offset lives on rank 1: rank 1 reports its own failure, rank 0 -- whose solves were fine --
    learns it from the status slot, and neither blocks."""
    c1 = load_c1()
    blocks = [copy.copy(b) for b in c1.blocks]
    bad = copy.copy(blocks[3])
    bad.offset = bad.offset.copy()
    bad.offset[5] = np.nan
    blocks[3] = bad
    engs = _two_handles(blocks, c1.n_global, [1.0], [1.0], monkeypatch)
    _, err = _run_ranks(engs, lambda r, e: e.iterate(0.01))
    assert all(isinstance(e, dataset.ModelFittingError) for e in err), err
    assert "another rank" in str(err[0]) and "another rank" not in str(err[1])
    for e in engs:
        e.close()


def test_library_rccl_two_processes_on_two_devices(tmp_path):
    """The library's OWN RCCL path with nranks = 2 (what the driver's 8-GPU run and a JVM host with `gpus=0,1,...` use): two
    processes, one device each, mlx_comm_init over a file-shared unique id, mlx_naive_init + four mlx_admm_iterate -- against one
    handle holding all eight partitions. Both ranks end with the SAME consensus, within 1e-5 of the one-handle run (the all-reduce
    associates the two partial means differently). Skipped on a one-GPU box (RCCL refuses two ranks on one device; the in-process
    communicator of the experimental build covers the control flow there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    script = tmp_path / "rank.py"
    script.write_text('''
import os, sys, time
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import mlease_amd
from mlease_amd.hip_engine import HipAdmmEngine
from fixtures import load_c1
rank, out = int(sys.argv[1]), sys.argv[2]
c1 = load_c1()
lam, rho = [1.0, 10.0], [1.0, 1.0]
uidf = os.path.join(out, "uid.bin")
if rank == 0:
    uid = HipAdmmEngine.comm_unique_id()
    with open(uidf + ".tmp", "wb") as fh:
        fh.write(uid)
    os.replace(uidf + ".tmp", uidf)
else:
    for _ in range(600):
        if os.path.exists(uidf):
            break
        time.sleep(0.1)
    uid = open(uidf, "rb").read()
eng = HipAdmmEngine(c1.n_global, lam, rho, 8, device=rank)
for b in c1.blocks[rank::2]:
    eng.add_partition(b)
eng.finalize()
eng.comm_init(uid, 2, rank)
eng.naive_init(0.01)
md = []
for it in range(4):
    md.append(eng.iterate(0.01).maxdiff)
np.savez(os.path.join(out, "rank%%d.npz" %% rank), z=eng.z()[0], maxdiff=np.array(md))
eng.close()
''' % (ROOT, ROOT))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank is blocked in the exchange")
        assert p.returncode == 0, e[-3000:]
    c1 = load_c1()
    ref = HipAdmmEngine(c1.n_global, [1.0, 10.0], [1.0, 1.0], 8)
    for b in c1.blocks:
        ref.add_partition(b)
    ref.finalize()
    ref.naive_init(0.01)
    mdr = [ref.iterate(0.01).maxdiff for _ in range(4)]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["z"], r1["z"]) and np.array_equal(r0["maxdiff"], r1["maxdiff"])
    zr = ref.z()[0]
    assert np.max(np.abs(r0["z"] - zr)) <= 1e-5 * np.max(np.abs(zr))
    assert np.max(np.abs(r0["maxdiff"] - np.array(mdr))) <= 1e-9
    ref.close()


@pytest.mark.parametrize("numerics", ["fast", "reference_order"])
@pytest.mark.parametrize("kind", ["dense", "onehot"])
def test_one_two_four_eight_shards_give_the_same_bits(kind, numerics):
    """(numerics: round 6 runs the same check under the reference-order contract too -- dense TILES on k_ro_dense_rows / k_ro_dense_cols,
    one-hot CSR partitions on the reference-order tick kernels with the wave folds of mlx_seqfold.h; there the one-shard run is also
    compared with the oracle twin, bit for bit, so every shard count equals the reference's sequential arithmetic.)
    VERDICT r4 item 6: the 64-partition job strong-scaled over 1 / 2 / 4 / 8 GPUs must be ONE computation. Since round 4 a
    partition's partial sums do not depend on the chunking its handle picks, and with num.blocks a power of two the consensus sums
    (float32 values times 1 / num.blocks, added in double) are exact, so their association over ranks cannot matter either. Here:
    N handles on one device, partition k -> handle k mod N, the split API (solve_local, the caller's sum of the [xbar | ubar]
    buffers in rank order, consensus_finish) for 3 iterations: the double z, maxdiff and every partition's TRON counters are
    identical for N = 1, 2, 4, 8 -- dense tiles (k_xpass_dense + k_tron_step) and one-hot CSR partitions on the tick kernels."""
    import torch
    from fixtures import dense_blocks, onehot_blocks
    if kind == "dense":
        pd, eps = dense_blocks(8 * 4200, 96, 8), [1e-2, 1e-2, 1e-4]
    else:
        pd, eps = onehot_blocks(8 * 8000, 8), [1e-2, 1e-2, 1e-2]
        assert all(len(b.col_idx) > 65536 for b in pd.blocks)           # large enough for the tick kernels
    runs = {}
    for N in (1, 2, 4, 8):
        engs = []
        for r in range(N):
            e = HipAdmmEngine(pd.n_global, [1.0], [1.0], 8, numerics=None if numerics == "fast" else numerics)
            for b in pd.blocks[r::N]:
                e.add_partition(b)
            e.finalize()
            if numerics != "fast":
                assert e.get_option("numerics_kernels") == "reference_order_ticks"
                assert int(e.get_option("dense_tiles")) == (len(pd.blocks[r::N]) if kind == "dense" else 0)
            engs.append(e)
        rec = []
        for ep in eps:
            for e in engs:
                e.solve_local(ep, 1.0)
            bufs = [e.consensus_tensor() for e in engs]
            tot = bufs[0].clone()
            for t in bufs[1:]:
                tot += t
            for t in bufs:
                t.copy_(tot)
            torch.cuda.synchronize()
            fins = [e.consensus_finish() for e in engs]
            cnt = np.zeros((8, 4), np.int64)
            for r, e in enumerate(engs):
                cnt[r::N] = e.solve_counters()
            zs = [e.z()[0].copy() for e in engs]
            assert all(np.array_equal(z, zs[0]) for z in zs) and all(f.maxdiff == fins[0].maxdiff for f in fins)
            rec.append((zs[0], fins[0].maxdiff, cnt))
        runs[N] = rec
        for e in engs:
            e.close()
    for N in (2, 4, 8):
        for it, (a, b) in enumerate(zip(runs[1], runs[N])):
            assert np.array_equal(a[2], b[2]), "%s, %d shards, iteration %d: TRON counters differ from the one-shard run" % (kind, N, it + 1)
            assert np.array_equal(a[0], b[0]) and a[1] == b[1], "%s, %d shards, iteration %d: z (double) differs" % (kind, N, it + 1)
    assert runs[1][-1][2][:, 2].sum() > 0
    if numerics != "fast":
        import oracle_lib as ol
        oc = ol.OracleAdmm(pd.blocks, pd.n_global, [1.0], [1.0], pm=True)
        for it, ep in enumerate(eps):
            oc.iterate(ep, 1.0, nthreads=8)
            cc = np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in oc.stats()], np.int64)
            assert np.array_equal(runs[8][it][2], cc) and np.array_equal(runs[8][it][0], oc.z()[0]), "8 shards, iteration %d: not the oracle twin's bits" % (it + 1)
