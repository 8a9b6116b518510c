#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X-native ADMM L2-logistic hot path (one JSON line on stdout).

Headline (BASELINE.json metric, configs[1]): partition Newton-solves/sec on synthetic dense 1M x 1K, 64 partitions
(row % 64), single lambda. One "step" = one ADMM iteration = one batched TRON solve of every (partition, lambda)
problem + the consensus z/u update. Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU. Default is STRONG scaling, the metric's "1Mx1K at 1/2/4/8 GPU": the same 64-partition job,
partition k -> rank k mod N (64/N per GPU), the consensus means [xbar | ubar] all-reduced over RCCL (torch.distributed
backend "nccl"). --scaling weak keeps 64 partitions per GPU (num.blocks = 64 N).

Extra top-level keys of the same JSON line:
  roofline      dominant kernel of the headline (k_xpass_dense): algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  cpu_baseline  the C oracle on the host cores, ALL 64 partitions, the SAME ADMM iterations as the first timed ones
                (it starts from the GPU's state after the warm-up iterations), one thread per partition solve
  parity_check  the GPU re-run of exactly those iterations against the oracle's result
  time_to_ref_loglik  metric (ii): full run from z = 0, test log-likelihood per iteration against the ORACLE's
                committed 20-iteration value (tests/golden/c2_ref_loglik.json, tests/golden/make_ref_loglik.py)
  sparse        BASELINE configs[2] (N = 1) / configs[3] (N > 1): one-hot 10M x 100K, 20 nnz/row, 256 / 1024
                partitions, binary.feature, with per-kernel rooflines (row pass, column pass, TRON/CG step)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MATRIX_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet, fp64 matrix (v_mfma_f64_*); the in-image guide lists no fp64 figure
ROWS, NFEAT, PARTS = 1_000_000, 1000, 64           # BASELINE configs[1]
SP_ROWS, SP_PARTS_1GPU, SP_PARTS_MULTI = 10_000_000, 256, 1024     # configs[2] / configs[3]


class EpsSchedule:
    """liblinear epsilon of the driver loop (jobs/RegressionAdmmTrain.java:279,338-346): float32, /10 once mindiff < 1e-3."""

    def __init__(self, admm):
        self.admm = admm
        self.e = np.float32(0.01)
        self.mindiff = 99999999.0
        self.it = 0

    def next(self):
        self.it += 1
        if self.it > 1 and self.mindiff < 0.001:
            self.e = np.float32(self.e / np.float32(10))
        return self.admm.float_string_roundtrip(self.e)


def mem_available_gb():
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


def usable_cores():
    """Cores this process can really use: os.cpu_count() capped by the scheduler affinity and by the cgroup CPU quota (the GPU
    box shows 256 CPUs but grants the container 16 cores' worth of time: more threads than that are only throttled)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            if q != "max" and int(period) > 0:
                n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--rows", type=int, default=ROWS, help="rows of the dense job (strong) / per 64 partitions (weak)")
    ap.add_argument("--features", type=int, default=NFEAT)
    ap.add_argument("--partitions", type=int, default=PARTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=4, help="ADMM iterations of the CPU-baseline leg (the first timed ones)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--loglik-iters", type=int, default=20, help="ADMM iterations of the time-to-reference-loglik run (0 = skip)")
    ap.add_argument("--test-rows", type=int, default=100000)
    ap.add_argument("--no-sparse", action="store_true", help="skip the configs[2]/[3] leg")
    ap.add_argument("--no-gram", action="store_true", help="skip the fp64-MFMA Gram measurement (posterior covariance of one partition)")
    ap.add_argument("--sparse-only", action="store_true", help="run only the sparse leg (development / profiling)")
    ap.add_argument("--sparse-steps", type=int, default=3)
    ap.add_argument("--sparse-warmup", type=int, default=1)
    ap.add_argument("--sparse-rows", type=int, default=SP_ROWS)
    ap.add_argument("--sparse-partitions", type=int, default=0, help="0 = 256 at one GPU, 1024 sharded otherwise")
    ap.add_argument("--sparse-cpu-sample", type=int, default=16, help="partitions of the sparse CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    # stdout must carry exactly ONE line (the JSON): anything a library prints to fd 1 (RCCL prints a version banner
    # through C stdio, flushed at exit) is sent to stderr instead; the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import mlease_amd  # noqa: F401
    from mlease_amd import admm
    from mlease_amd.hip_engine import HipAdmmEngine
    import synth_data as sd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # MLX_BENCH_SHARE_GPU=1 (test mode, never a measurement): every rank uses device 0 and the collectives run over gloo
    # through host staging -- lets ONE GPU execute the whole N>1 control flow of this file (sharding, exchange, reductions).
    share = os.environ.get("MLX_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:          # under torch.distributed.run also at N=1 (RCCL path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def _collective(t, op):
        if share:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    def all_reduce(t):
        # The library runs on its own HIP stream and is blocking, so the buffer is complete when we get here; the
        # collective runs on RCCL's stream, ordered against torch's current stream only. Wait for it on the host
        # before the library's next kernels (consensus_finish) read the summed buffer.
        if dist is not None:
            _collective(t, dist.ReduceOp.SUM)
            torch.cuda.current_stream().synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        _collective(t, dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(xs):
        if dist is None:
            return [float(x) for x in xs]
        t = torch.tensor(list(xs), device=dev, dtype=torch.float64)
        _collective(t, dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    ctx = dict(torch=torch, dev=dev, dist=dist, world=world, rank=rank, local_rank=local_rank, stream=stream, admm=admm,
               sd=sd, HipAdmmEngine=HipAdmmEngine, all_reduce=all_reduce, barrier=barrier, reduce_max=reduce_max,
               reduce_sum=reduce_sum)
    out = {}
    if not args.sparse_only:
        out = run_dense(args, ctx)
    if not args.no_sparse:
        sp = run_sparse(args, ctx)
        if rank == 0:
            if args.sparse_only:
                out = sp
            else:
                out["sparse"] = sp
    if rank == 0:
        if share:
            out["test_mode"] = "MLX_BENCH_SHARE_GPU=1: all ranks on ONE device, collectives over gloo -- control-flow check, not a measurement"
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ======================================================================================================================
def run_dense(args, C):
    torch, dev, world, rank, sd, admm = C["torch"], C["dev"], C["world"], C["rank"], C["sd"], C["admm"]
    nf = args.features
    N = args.partitions * (world if args.scaling == "weak" else 1)          # num.blocks of the job
    rows_total = args.rows * (world if args.scaling == "weak" else 1)
    rows = rows_total // N                                                  # rows per partition (row % N assignment)
    mine = [k for k in range(N) if k % world == rank]                       # partition k -> rank k mod G
    P = len(mine)
    eng = C["HipAdmmEngine"](nf + 1, [1.0], [1.0], N, device=C["local_rank"], stream=C["stream"], profiling=not args.no_profile)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if want_cpu and mem_available_gb() < 48:
        want_cpu = False
        print("[bench] CPU baseline skipped: less than 48 GB of host memory available", file=sys.stderr)
    sample = []
    for k in mine:
        X, y = sd.dense_rows_torch(torch, dev, k, rows, nf, stride=N)
        torch.cuda.synchronize()
        eng.add_partition_dense_device(k, X.data_ptr(), rows, nf, nf, y.data_ptr())
        if want_cpu:
            sample.append((X.cpu().numpy(), y.cpu().numpy()))
        del X, y
    eng.finalize()
    torch.cuda.synchronize()

    sched = EpsSchedule(admm)
    eps_used = []
    acc = dict(solves=0, newton=0, cg=0, passes_ref=0, passes_dev=0, ticks=0, alg_bytes=0.0, xpass_ms=0.0,
               total_ms=0.0, launches=0)
    allrun = dict(alg_bytes=P * (4.0 * rows * nf + 8.0 * rows + 8.0 * (nf + 1)), launches=1)   # + the c0 pass of finalize

    def step(timed):
        eps = sched.next()
        eps_used.append(eps)
        st = eng.solve_local(eps, 1.0)
        C["all_reduce"](eng.consensus_tensor())
        fin = eng.consensus_finish()
        sched.mindiff = fin.mindiff
        allrun["alg_bytes"] += st.alg_bytes_dev
        allrun["launches"] += st.ticks
        if timed:
            acc["solves"] += st.solves; acc["newton"] += st.newton_iters; acc["cg"] += st.cg_iters
            acc["passes_ref"] += st.x_passes_ref; acc["passes_dev"] += st.x_passes_dev; acc["ticks"] += st.ticks
            acc["alg_bytes"] += st.alg_bytes_dev; acc["xpass_ms"] += st.xpass_ms; acc["total_ms"] += st.total_ms
            acc["launches"] += st.xpass_launches
        return st, fin

    for _ in range(args.warmup):
        step(False)
    snap = None
    if want_cpu:                       # the state every timed iteration starts from (outside the timed region)
        snap = (eng.z()[0].copy(), np.stack([eng.partition_model(i, 0)[2] for i in range(P)])[:, None, :].copy(),
                sched.e, sched.mindiff, sched.it)
    C["barrier"]()
    t0 = time.perf_counter()
    step_times = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        st, fin = step(True)
        step_times.append((time.perf_counter() - ts, st.solves, st.x_passes_ref))
    C["barrier"]()
    dt = C["reduce_max"](time.perf_counter() - t0)
    tot_solves, tot_pref, tot_pdev, tot_cg, tot_newton = C["reduce_sum"](
        [acc["solves"], acc["passes_ref"], acc["passes_dev"], acc["cg"], acc["newton"]])

    # ---- metric (ii): ADMM wall-clock to the reference test log-likelihood: a full run from z = u = 0 with
    # the per-iteration test loglik (jobs/RegressionAdmmTrain.java:766-845) on the 100 000 held-out rows; the target is
    # the ORACLE's value after its 20th iteration on the same data (tests/golden/c2_ref_loglik.json). Outside the timed region.
    loglik = None
    if args.loglik_iters > 0:
        loglik = loglik_run(args, C, eng, P, nf, N, rows_total)

    out = None
    if rank == 0:
        value = tot_solves / dt
        roof = None
        if acc["xpass_ms"] > 0:
            achieved = acc["alg_bytes"] / (acc["xpass_ms"] * 1e-3) / 1e9
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as fh:
                    tj = json.load(fh)
                traffic = tj["hbm_bytes_per_alg_byte"] * acc["alg_bytes"] / max(1, acc["launches"])
                tsrc = "profiles/traffic.json: (2 x FETCH_SIZE + WRITE_SIZE) / algorithmic bytes = %.4f from the committed rocprofv3 " \
                       "--pmc passes of `%s`, times this run's algorithmic bytes per launch (not a counter read in this run)" % (
                           tj["hbm_bytes_per_alg_byte"], tj.get("command", "bench.py"))
            roof = {"bound": "hbm", "kernel": "k_xpass_dense<4,4>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                    "alg_bytes_per_launch": acc["alg_bytes"] / max(1, acc["launches"]),
                    "avg_launch_ms": acc["xpass_ms"] / max(1, acc["launches"]), "launches": acc["launches"],
                    "xpass_share_of_step": round(acc["xpass_ms"] / (dt * 1e3), 4)}
        out = {"metric": "partition Newton-solves/sec (ADMM L2-LR, dense 1Mx1K, 64 partitions)",
               "value": round(value, 3), "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: synthetic dense %d rows x %d features, %d partitions (row %% %d)%s, lambda=1, "
                                      "rho=1, fp32-stored X, fp64 arithmetic" % (
                                          rows * N, nf, N, N, " sharded k -> rank k mod %d" % world if world > 1 else ""),
                          "rows": rows * N, "features": nf, "partitions": N, "partitions_per_gpu": P, "lambda": [1.0], "rho": [1.0],
                          "admm_iterations_timed": [args.warmup + 1, args.warmup + args.steps],
                          "exchange": "rccl all_reduce of [xbar|ubar] (%d doubles)" % (2 * (nf + 1)) if world > 1 else "none (1 GPU)"},
               "work": {"solves": tot_solves, "tron_iters_per_s": round(tot_newton / dt, 2), "cg_steps_per_s": round(tot_cg / dt, 2),
                        "x_passes_ref_per_s": round(tot_pref / dt, 2), "x_passes_dev_per_s": round(tot_pdev / dt, 2),
                        "passes_ref_per_solve": round(tot_pref / max(1.0, tot_solves), 2),
                        "passes_dev_per_solve": round(tot_pdev / max(1.0, tot_solves), 2), "ticks": acc["ticks"],
                        "last_maxdiff": fin.maxdiff},
               "roofline": roof,
               "time_to_ref_loglik": loglik,
               "all_launches": {"xpass_launches_incl_c0_and_warmup": allrun["launches"], "alg_bytes": allrun["alg_bytes"]}}
        if not args.no_gram:
            out["gram"] = gram_leg(eng, rows, nf)
        if want_cpu:
            cpu_leg(args, C, eng, out, sample, snap, eps_used, step_times, nf, N)
    eng.close()
    return out


def gram_leg(eng, rows, nf):
    """north_star's "MFMA utilisation for the dense Gram path": the fp64-MFMA build of X'DX (full posterior covariance at the
    current consensus, llf/LibLinear.java:314-337 -> mlx_posterior_variance) on one partition of the job, outside the timed
    region. Flops: executed = the lower-triangle 128x128 blocks the kernel computes (intercept = an implicit ones column);
    algorithmic = the reference's triangle loop 2 * l * n(n+1)/2."""
    w = np.ascontiguousarray(eng.z()[0][0], np.float64)
    pv = np.ones(nf + 1)
    ms, wall = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        _, _, m = eng.posterior_variance(0, w, pv, True)
        wall.append(time.perf_counter() - t0)
        ms.append(m)
    nb = (nf + 1 + 127) // 128
    flops_exec = 2.0 * (nb * (nb + 1) // 2) * 128 * 128 * rows
    flops_alg = 1.0 * rows * (nf + 1) * (nf + 2)
    best = min(ms)
    return {"kernel": "k_gram_f64 (v_mfma_f64_16x16x4_f64)", "bound": "mfma", "workload": "X'DX of one %d x %d partition + intercept column" % (rows, nf),
            "achieved": round(flops_exec / best / 1e9, 2), "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flops_exec / best / 1e9 / FP64_MATRIX_PEAK_TFLOPS, 4), "achieved_algorithmic": round(flops_alg / best / 1e9, 2),
            "kernel_ms": [round(x, 4) for x in ms], "wall_s_incl_threaded_host_cholesky_inverse": [round(x, 3) for x in wall],
            "peak_source": "AMD MI355X datasheet fp64 matrix rate (SURVEY 8d); not in the in-image guide",
            "note": "off the ADMM path: the solve is matrix-free (DESIGN 4); this is where the reference builds the dense Hessian"}


def loglik_run(args, C, eng, P, nf, N, rows_total):
    torch, dev, sd, admm, rank = C["torch"], C["dev"], C["sd"], C["admm"], C["rank"]
    lt = args.test_rows
    if rank == 0:
        chunks = []
        for c0 in range(0, lt, 16384):
            Xt, yt = sd.dense_rows_torch(torch, dev, rows_total + c0, min(16384, lt - c0), nf)      # rows beyond the training range
            chunks.append((Xt.cpu().numpy(), yt.cpu().numpy()))
            del Xt, yt
        Xt = np.concatenate([c[0] for c in chunks])
        yt = np.concatenate([c[1] for c in chunks])
        eng.set_test_data(np.arange(0, (lt + 1) * nf, nf, dtype=np.int64), np.tile(np.arange(nf, dtype=np.int32), lt),
                          Xt.reshape(-1), np.where(yt == 1, 1, 0).astype(np.int8))
        del Xt, chunks
    eng.set_state(np.zeros((1, nf + 1)), np.zeros((P, 1, nf + 1), np.float32))
    sched = EpsSchedule(admm)
    lls, walls = [], []
    C["barrier"]()
    tl0 = time.perf_counter()
    for _ in range(args.loglik_iters):
        eng.solve_local(sched.next(), 1.0)
        C["all_reduce"](eng.consensus_tensor())
        sched.mindiff = eng.consensus_finish().mindiff
        if rank == 0:
            lls.append(float(eng.test_loglik_sums()[0]) / lt)
        walls.append(time.perf_counter() - tl0)
    if rank != 0:
        return None
    ref, ref_src = None, None
    gpath = os.path.join(ROOT, "tests", "golden", "c2_ref_loglik.json")
    default_job = (rows_total == ROWS and nf == NFEAT and N == PARTS and lt == 100000)
    if os.path.exists(gpath) and default_job:
        with open(gpath) as fh:
            gj = json.load(fh)
        if len(gj["loglik_by_iteration"]) >= args.loglik_iters:
            ref = gj["loglik_by_iteration"][args.loglik_iters - 1]
            ref_src = "tests/golden/c2_ref_loglik.json: oracle/admm_oracle.c after ADMM iteration %d of the same job (tests/golden/make_ref_loglik.py)" % args.loglik_iters
    res = {"test_rows": lt, "iterations": args.loglik_iters, "seconds_all_iterations": round(walls[-1], 4),
           "loglik_by_iteration": [round(v, 8) for v in lls]}
    if ref is not None:
        # "reached" = the first iteration from which the test loglik STAYS within 1e-5 of the oracle's final value (the
        # sequence is not monotone: it overshoots in the first iterations and settles from above)
        tol = 1e-5
        inside = [abs(v - ref) <= tol for v in lls]
        reached = None
        for i in range(len(lls)):
            if all(inside[i:]):
                reached = i
                break
        res.update({"ref_loglik": ref, "ref_source": ref_src, "tolerance": tol,
                    "abs_diff_to_oracle_by_iteration_max": max(abs(a - b) for a, b in zip(lls, gj["loglik_by_iteration"])),
                    "reached_at_iteration": None if reached is None else reached + 1,
                    "seconds_to_ref_loglik": None if reached is None else round(walls[reached], 4)})
    else:
        res.update({"ref_loglik": None, "ref_source": "no committed oracle value for this job shape"})
    return res


def cpu_leg(args, C, eng, out, sample, snap, eps_used, step_times, nf, N):
    """cpu_baseline + parity_check: the oracle on ALL partitions of the job, the same ADMM iterations as the first
    `cpu_iters` timed GPU iterations (both start from the GPU's state after the warm-up), one thread per partition solve."""
    import oracle_lib as ol
    from mlease_amd.dataset import PartitionBlock
    kc = max(1, min(args.cpu_iters, args.steps))
    Z0, u0, _, _, it0 = snap
    eps = eps_used[it0:it0 + kc]
    blocks = []
    col = None
    for k, (Xh, yh) in enumerate(sample):
        l = Xh.shape[0]
        if col is None or len(col) != l * nf:
            col = np.tile(np.arange(nf, dtype=np.int32), l)
        blocks.append(PartitionBlock(k, l, nf + 1, np.arange(0, (l + 1) * nf, nf, dtype=np.int64), col, Xh.reshape(-1), yh,
                                     np.ones(l, np.float32), np.zeros(l, np.float32), np.arange(nf + 1, dtype=np.int32)))
    oc = ol.OracleAdmm(blocks, nf + 1, [1.0], [1.0], num_blocks=N)
    del blocks
    sample.clear()
    oc.set_state(Z0, u0)
    threads = min(usable_cores(), N)
    solves = passes = 0
    cnts = []
    t0 = time.perf_counter()
    for e in eps:
        oc.iterate(e, 1.0, nthreads=threads)
        st = oc.stats()
        solves += len(st)
        passes += sum(s.x_passes for s in st)
        cnts.append(np.array([(s.newton_iters, s.accepted, s.cg_iters, s.x_passes) for s in st]))
    cdt = time.perf_counter() - t0
    z_orc = oc.z()[1][0].astype(np.float64)
    # the GPU on exactly these iterations, from the same state (timed again here: the like-for-like ratio)
    eng.set_state(Z0, u0)
    same = True
    C["torch"].cuda.synchronize()
    tg = time.perf_counter()
    per_it = []
    only_last_accept = True        # every mismatch = same TRON iterations and CG steps, `accepted` (and with it the passes) off by one
    for i, e in enumerate(eps):
        eng.solve_local(e, 1.0)
        eng.consensus_finish()
        gc = eng.solve_counters()
        eq = np.all(gc == cnts[i], axis=1)
        same = same and bool(eq.all())
        bad = np.nonzero(~eq)[0]
        for k in bad:
            dlt = gc[k].astype(np.int64) - cnts[i][k].astype(np.int64)
            only_last_accept = only_last_accept and dlt[0] == 0 and dlt[2] == 0 and abs(int(dlt[1])) == 1 and dlt[3] == dlt[1]
        per_it.append({"iteration": it0 + i + 1, "liblinear_epsilon": e, "partitions_with_equal_counters": int(eq.sum()),
                       "first_mismatches_[partition,gpu(newton,accepted,cg,passes),cpu(...)]":
                           [[int(k), [int(x) for x in gc[k]], [int(x) for x in cnts[i][k]]] for k in bad[:4]]})
    gdt = time.perf_counter() - tg
    z = eng.z()[1][0].astype(np.float64)
    floor = 1e-4 * float(np.max(np.abs(z_orc)))
    err = float(np.max(np.abs(z - z_orc) / np.maximum(np.abs(z_orc), floor)))
    ident = float(np.mean(z.astype(np.float32) == z_orc.astype(np.float32)))
    v = solves / cdt
    out["cpu_baseline"] = {"value": round(v, 4), "unit": "solves/s", "cores": threads, "kind": "port",
                           "sample": "oracle/admm_oracle.c (-O2, fp64, one thread per partition solve) on ALL %d partitions of the "
                                     "job, ADMM iterations %d..%d (started from the GPU's z/u after iteration %d, same epsilon "
                                     "schedule), %.1f s wall" % (N, it0 + 1, it0 + kc, it0, cdt),
                           "x_passes_ref_per_s": round(passes / cdt, 2), "host_cpus_listed": os.cpu_count(), "host_cores_usable": usable_cores()}
    g_solves = sum(s[1] for s in step_times[:kc])
    g_time = sum(s[0] for s in step_times[:kc])
    g_pass = sum(s[2] for s in step_times[:kc])
    out["gpu_over_cpu"] = {"same_iterations": [it0 + 1, it0 + kc],
                           "solves_per_s": round(g_solves / g_time / v, 2),
                           "x_passes_ref_per_s": round(g_pass / g_time / (passes / cdt), 2),
                           "gpu_solves_per_s_on_these_iterations": round(g_solves / g_time, 2),
                           "gpu_rerun_seconds": round(gdt, 4), "cpu_seconds": round(cdt, 2)}
    out["parity_check"] = {"what": "GPU vs oracle, all %d partitions (%d x %d each), ADMM iterations %d..%d from the same state" % (
                               N, out["config"]["rows"] // N, nf, it0 + 1, it0 + kc),
                           "max_rel_err_z": err, "rel_err_floor": "1e-4 * max|z|", "tolerance": 1e-5, "tron_counters_equal": same,
                           "bit_identical_float32_fraction": round(ident, 4),
                           "counters_equal_through_iteration": next((x["iteration"] - 1 for x in per_it if x["partitions_with_equal_counters"] < N), it0 + kc),
                           "mismatch_kind": None if same else (
                               "accept/reject of the LAST TRON step only (same TRON iterations, same CG steps): at liblinear epsilon <= 1e-7 the solve "
                               "ends on bw/Tron.java:115-122's |actred|, |prered| <= 1e-12 |f| tests, where actred = f - fnew is the rounding "
                               "noise of two 15 625-term sums and its comparison with eta0 * prered (:102) depends on the summation order"
                               if only_last_accept else "other (see per_iteration)"),
                           "per_iteration": per_it}


# ======================================================================================================================
def run_sparse(args, C):
    """BASELINE configs[2] at one GPU (256 partitions), configs[3] sharded (1024 partitions, k -> rank k mod N)."""
    world, rank, sd, admm = C["world"], C["rank"], C["sd"], C["admm"]
    from mlease_amd.dataset import PartitionBlock
    Ptot = args.sparse_partitions or (SP_PARTS_1GPU if world == 1 else SP_PARTS_MULTI)
    rows = args.sparse_rows // Ptot
    mine = [k for k in range(Ptot) if k % world == rank]
    t0 = time.time()
    blocks, ng = [], None
    for k in mine:
        rp, ci, y, l2g, ng = sd.onehot_partition(k, rows)
        blocks.append(PartitionBlock(k, rows, len(l2g), rp, ci, None, y, np.ones(rows, np.float32), np.zeros(rows, np.float32), l2g))
    tgen = time.time() - t0
    eng = C["HipAdmmEngine"](ng, [1.0], [1.0], Ptot, device=C["local_rank"], stream=C["stream"], profiling=True)
    t0 = time.time()
    eng.add_partitions(blocks)
    eng.finalize()
    tup = time.time() - t0
    nnz = sum(b.nnz for b in blocks)
    nloc = np.array([b.n_local for b in blocks])
    sched = EpsSchedule(admm)
    acc = dict(solves=0, cg=0, newton=0, pref=0, pdev=0, ticks=0, alg=0.0, rms=0.0, cms=0.0, sms=0.0, tms=0.0)
    # every pass launch of this leg (finalize's c0 pass: one row + one column pass per partition, + warm-up + timed): what a
    # rocprofv3 run of this command sees, used to turn its FETCH_SIZE / WRITE_SIZE sums into bytes per algorithmic byte
    allrun = dict(alg=sum(2.0 * (4.0 * b.nnz + 8.0 * b.l + 8.0 * b.n_local) for b in blocks), ticks=1)
    fin = None
    for it in range(1, args.sparse_warmup + args.sparse_steps + 1):
        if it == args.sparse_warmup + 1:
            C["barrier"]()
            tstart = time.perf_counter()
        st = eng.solve_local(sched.next(), 1.0)
        C["all_reduce"](eng.consensus_tensor())
        fin = eng.consensus_finish()
        sched.mindiff = fin.mindiff
        allrun["alg"] += st.alg_bytes_dev
        allrun["ticks"] += st.ticks
        if it > args.sparse_warmup:
            acc["solves"] += st.solves; acc["cg"] += st.cg_iters; acc["newton"] += st.newton_iters
            acc["pref"] += st.x_passes_ref; acc["pdev"] += st.x_passes_dev; acc["ticks"] += st.ticks
            acc["alg"] += st.alg_bytes_dev; acc["tms"] += st.total_ms
            acc["rms"] += st.rowpass_ms; acc["cms"] += st.colpass_ms; acc["sms"] += st.step_ms
    C["barrier"]()
    dt = C["reduce_max"](time.perf_counter() - tstart)
    tot_solves, tot_pref, tot_pdev, tot_alg = C["reduce_sum"]([acc["solves"], acc["pref"], acc["pdev"], acc["alg"]])
    if rank != 0:
        eng.close()
        return None
    # one X pass over one partition moves B_pass = nnz*4 + 8 l + 8 n bytes (binary.feature: no value array);
    # the library counts 2 passes (row + column) per tick and active problem in alg_bytes_dev
    half = acc["alg"] / 2.0
    n_mean = float(nloc.mean())

    def roof(kernel, ms, alg_bytes, note):
        a = alg_bytes / max(1e-9, ms * 1e-3) / 1e9
        return {"kernel": kernel, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 4), "ms": round(ms, 3), "share_of_step": round(ms / (dt * 1e3), 4),
                "us_per_tick": round(1e3 * ms / max(1, acc["ticks"]), 1), "alg_bytes": alg_bytes, "note": note}

    step_model = 13.0 * 8.0 * n_mean * (acc["pdev"] / 2.0)        # 13 n-vector streams per problem and tick
    res = {"workload": "BASELINE configs[%d]: synthetic one-hot %d rows x %d binary features (20 fields x 5000 Zipf(1.1) levels, 20 nnz/row), "
                       "%d partitions%s, lambda=1, rho=1" % (2 if world == 1 else 3, rows * Ptot, ng - 1, Ptot,
                                                             " sharded k -> rank k mod %d" % world if world > 1 else ""),
           "value": round(tot_solves / dt, 2), "unit": "solves/s", "n_gpus": world, "steps": args.sparse_steps, "warmup": args.sparse_warmup,
           "ms_per_step": round(dt * 1e3 / args.sparse_steps, 3),
           "nnz": int(nnz), "rows_per_partition": rows, "n_local_mean": n_mean, "gen_s": round(tgen, 1), "upload_s": round(tup, 1),
           "x_passes_ref_per_s": round(tot_pref / dt, 1), "x_passes_dev_per_s": round(tot_pdev / dt, 1),
           "ticks_per_step": acc["ticks"] / args.sparse_steps, "cg_per_solve": round(acc["cg"] / max(1, acc["solves"]), 2),
           "whole_step": {"alg_bytes_per_s_GB": round(tot_alg / dt / 1e9, 1), "frac_of_hbm_peak": round(tot_alg / dt / 1e9 / (HBM_PEAK_GBS * world), 4),
                          "definition": "sum over solves of device passes x B_pass (SURVEY 8d) / wall time of the timed iterations"},
           "roofline": [roof("k_rowcold + k_rowpass_lds<binary>", acc["rms"], half, "B_pass = nnz*4 + 8l + 8n per active problem; the cold column slices run as their own launch in front of the row kernel"),
                        roof("k_colpass_lds<binary>", acc["cms"], half, "B_pass = nnz*4 + 8l + 8n per active problem"),
                        roof("k_step_a+b+c+commit", acc["sms"], 0.0,
                             "no algorithmic X bytes (SURVEY 8d counts the n-vector work as zero); streams ~13 x 8n bytes per problem and "
                             "tick = %.1f GB/s" % (step_model / max(1e-9, acc["sms"] * 1e-3) / 1e9))],
           "last_maxdiff": fin.maxdiff,
           "all_launches": {"ticks_incl_c0_and_warmup": allrun["ticks"], "alg_bytes_row_plus_column": allrun["alg"]}}
    tpath = os.path.join(ROOT, "profiles", "traffic_sparse.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tj = json.load(fh)
        res["traffic"] = {"source": "profiles/traffic_sparse.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s` (committed; not a counter read in this run)" % tj.get("command", ""),
                          "hbm_bytes_per_alg_byte": tj.get("hbm_bytes_per_alg_byte"), "per_kernel": tj.get("per_kernel")}
    if args.sparse_cpu_sample > 0 and world == 1:
        import oracle_lib as ol
        ns = min(args.sparse_cpu_sample, len(blocks))
        oc = ol.OracleAdmm(blocks[:ns], ng, [1.0], [1.0], num_blocks=Ptot)
        threads = min(usable_cores(), ns)
        t0 = time.perf_counter()
        oc.solve_local(0.01, 1.0, nthreads=threads)
        cdt = time.perf_counter() - t0
        ps = sum(s.x_passes for s in oc.stats())
        res["cpu_baseline"] = {"value": round(ns / cdt, 3), "unit": "solves/s", "cores": threads, "kind": "port",
                               "sample": "oracle/admm_oracle.c on %d of the %d partitions, the solves of ADMM iteration 1 (z = u = 0, "
                                         "epsilon 0.01), %.1f s wall" % (ns, Ptot, cdt),
                               "x_passes_ref_per_s": round(ps / cdt, 1)}
    eng.close()
    return res


if __name__ == "__main__":
    main()
