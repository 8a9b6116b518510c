#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ADMM L2-logistic hot path.

Metric (BASELINE.json): partition Newton-solves/sec on synthetic dense 1M x 1K, 64 partitions, single
lambda (configs[1]); one "step" = one ADMM iteration = one batched TRON solve of every (partition, lambda)
problem + the consensus z/u update. Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, weak scaling (64 partitions of 15 625 x 1000 per GPU, num.blocks = 64 N,
partition k -> rank k mod N), consensus means all-reduced over RCCL (torch.distributed backend "nccl").
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SEED = 20260925


def gen_partition(torch, dev, pid, rows, nfeat, beta, bias):
    """Synthetic C2 partition, generated on the GPU: x ~ N(0,1) fp32, y ~ Bernoulli(sigmoid(x.beta* + b))."""
    g = torch.Generator(device=dev)
    g.manual_seed(SEED + 7919 * pid)
    X = torch.randn((rows, nfeat), generator=g, device=dev, dtype=torch.float32)
    logit = X.double() @ beta + bias
    yy = torch.bernoulli(torch.sigmoid(logit), generator=g)
    y = torch.where(yy > 0.5, 1, -1).to(torch.int8)
    return X, y


def cpu_baseline(sample, nfeat, iters, threads):
    """The oracle (C restatement of the reference path) on the host cores: its own ADMM job over the sampled
    partitions (num.blocks = len(sample)), `iters` iterations from z = 0, one thread per partition solve."""
    import oracle_lib as ol
    from mlease_amd.dataset import PartitionBlock
    blocks = []
    for k, (Xh, yh) in enumerate(sample):
        l = Xh.shape[0]
        blocks.append(PartitionBlock(k, l, nfeat + 1, np.arange(0, (l + 1) * nfeat, nfeat, dtype=np.int64),
                                     np.tile(np.arange(nfeat, dtype=np.int32), l), Xh.reshape(-1), yh,
                                     np.ones(l, np.float32), np.zeros(l, np.float32), np.arange(nfeat + 1, dtype=np.int32)))
    oc = ol.OracleAdmm(blocks, nfeat + 1, [1.0], [1.0])
    t0 = time.perf_counter()
    passes = 0
    for _ in range(iters):
        oc.iterate(0.01, 1.0, nthreads=threads)
        passes += sum(s.x_passes for s in oc.stats())
    dt = time.perf_counter() - t0
    counters = np.array([(st.newton_iters, st.accepted, st.cg_iters, st.x_passes) for st in oc.stats()])
    return len(blocks) * iters / dt, passes / dt, dt, oc.z()[1][0].copy(), counters


def parity_on_sample(HipAdmmEngine, sample, nfeat, iters, device, z_oracle, counters_oracle):
    """The same job the CPU baseline just ran (the sampled partitions as their own num.blocks ADMM problem, `iters`
    iterations from z = 0) on the GPU: full-size parity check inside the bench run, outside every timed region."""
    eng = HipAdmmEngine(nfeat + 1, [1.0], [1.0], len(sample), device=device)
    for k, (Xh, yh) in enumerate(sample):
        eng.add_partition_dense(k, Xh, yh)
    eng.finalize()
    for _ in range(iters):
        eng.iterate(0.01, 1.0)
    z = eng.z()[1][0].astype(np.float64)
    zo = z_oracle.astype(np.float64)
    floor = 1e-2 * float(np.max(np.abs(zo)))
    err = float(np.max(np.abs(z - zo) / np.maximum(np.abs(zo), floor)))
    same = bool(np.array_equal(eng.solve_counters(), counters_oracle))
    ident = float(np.mean(z.astype(np.float32) == zo.astype(np.float32)))
    eng.close()
    return {"what": "GPU vs oracle on the cpu_baseline job (%d partitions of %d x %d, %d iterations)" % (len(sample), sample[0][0].shape[0], nfeat, iters),
            "max_rel_err_z": err, "tolerance": 1e-5, "tron_counters_equal": same, "bit_identical_float32_fraction": round(ident, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows-per-partition", type=int, default=15625)
    ap.add_argument("--features", type=int, default=1000)
    ap.add_argument("--partitions-per-gpu", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32, help="partitions in the CPU-baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--loglik-iters", type=int, default=20, help="ADMM iterations of the time-to-reference-loglik run (0 = skip)")
    ap.add_argument("--test-rows", type=int, default=100000)
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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:          # under torch.distributed.run also at N=1 (RCCL path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    P, rows, nf = args.partitions_per_gpu, args.rows_per_partition, args.features
    N = P * world
    stream = torch.cuda.current_stream().cuda_stream
    eng = HipAdmmEngine(nf + 1, [1.0], [1.0], N, device=local_rank, stream=stream, profiling=not args.no_profile)
    gb = torch.Generator(device="cpu")
    gb.manual_seed(SEED)
    beta = (0.1 * torch.randn(nf, generator=gb, dtype=torch.float64)).to(dev)
    sample = []
    for i in range(P):
        pid = i * world + rank                                     # partition k -> rank k mod G
        X, y = gen_partition(torch, dev, pid, rows, nf, beta, -1.0)
        torch.cuda.synchronize()
        eng.add_partition_dense_device(pid, X.data_ptr(), rows, nf, nf, y.data_ptr())
        if rank == 0 and world == 1 and not args.no_cpu_baseline and i < args.cpu_sample:
            sample.append((X.cpu().numpy(), y.cpu().numpy()))
        del X, y
    eng.finalize()
    torch.cuda.synchronize()

    def all_reduce(t):
        # The library runs on its own HIP stream and is blocking, so the buffer is complete when we get here; the
        # collective runs on RCCL's stream, ordered against torch's current stream only. Wait for it on the host
        # before the library's next kernels (consensus_finish) read the summed buffer.
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.current_stream().synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # eps schedule of the driver loop (jobs/RegressionAdmmTrain.java:279,338-346)
    e = np.float32(0.01)
    mindiff = 99999999.0
    it = 0
    acc = dict(solves=0, newton=0, cg=0, passes_ref=0, passes_dev=0, ticks=0, alg_bytes=0.0, xpass_ms=0.0,
               total_ms=0.0, launches=0)
    # every X-pass launch of the process (finalize's c0 pass + warmup + timed): what a rocprofv3 run of this
    # command sees, used to turn its FETCH_SIZE/WRITE_SIZE sums into HBM bytes per algorithmic byte
    allrun = dict(alg_bytes=P * (4.0 * rows * nf + 8.0 * rows + 8.0 * (nf + 1)), launches=1)

    def step(timed):
        nonlocal e, mindiff, it
        it += 1
        if it > 1 and mindiff < 0.001:
            e = np.float32(e / np.float32(10))
        st = eng.solve_local(admm.float_string_roundtrip(e), 1.0)
        all_reduce(eng.consensus_tensor())
        fin = eng.consensus_finish()
        mindiff = fin.mindiff
        allrun["alg_bytes"] += st.alg_bytes_dev
        allrun["launches"] += st.ticks
        if timed:
            acc["solves"] += st.solves; acc["newton"] += st.newton_iters; acc["cg"] += st.cg_iters
            acc["passes_ref"] += st.x_passes_ref; acc["passes_dev"] += st.x_passes_dev; acc["ticks"] += st.ticks
            acc["alg_bytes"] += st.alg_bytes_dev; acc["xpass_ms"] += st.xpass_ms; acc["total_ms"] += st.total_ms
            acc["launches"] += st.xpass_launches
        return fin

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fin = step(True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cnt = torch.tensor([acc["solves"], acc["passes_ref"], acc["passes_dev"], acc["cg"], acc["newton"]], device=dev, dtype=torch.float64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_solves, tot_pref, tot_pdev, tot_cg, tot_newton = [float(x) for x in cnt.tolist()]
    else:
        tot_solves, tot_pref, tot_pdev, tot_cg, tot_newton = acc["solves"], acc["passes_ref"], acc["passes_dev"], acc["cg"], acc["newton"]

    # ---- metric (ii): ADMM wall-clock to the reference test log-likelihood. A full run from z = u = 0:
    # num.iters iterations with the per-iteration test loglik (jobs/RegressionAdmmTrain.java:766-845) on fresh test
    # rows from the same generator. "Reference loglik" = the loglik of the final iteration of that same algorithm
    # (what the reference run ends with; by parity the same iterates). Outside the timed region above.
    loglik = None
    if args.loglik_iters > 0:
        tpid = 1_000_000 + rank                      # a partition id no training partition uses -> fresh rows
        tparts = []
        for c in range((args.test_rows + rows - 1) // rows):
            Xt, yt = gen_partition(torch, dev, tpid + 1000 * c, rows, nf, beta, -1.0)
            tparts.append((Xt.cpu().numpy(), yt.cpu().numpy()))
            del Xt, yt
        Xt = np.concatenate([p[0] for p in tparts])[:args.test_rows]
        yt = np.concatenate([p[1] for p in tparts])[:args.test_rows]
        lt = Xt.shape[0]
        eng.set_test_data(np.arange(0, (lt + 1) * nf, nf, dtype=np.int64), np.tile(np.arange(nf, dtype=np.int32), lt),
                          Xt.reshape(-1), np.where(yt == 1, 1, 0).astype(np.int8))
        del Xt, tparts
        eng.set_state(np.zeros((1, nf + 1)), np.zeros((P, 1, nf + 1), np.float32))
        e = np.float32(0.01)
        mindiff = 99999999.0
        it = 0
        lls, walls = [], []
        barrier()
        tl0 = time.perf_counter()
        for _ in range(args.loglik_iters):
            step(False)
            lls.append(float(eng.test_loglik_sums()[0]) / lt)
            walls.append(time.perf_counter() - tl0)
        ref = lls[-1]
        reached = next(i for i, v in enumerate(lls) if v >= ref - 1e-12 * abs(ref))
        loglik = {"test_rows": lt, "iterations": args.loglik_iters, "ref_loglik": ref, "reached_at_iteration": reached + 1,
                  "seconds_to_ref_loglik": round(walls[reached], 4), "seconds_all_iterations": round(walls[-1], 4),
                  "loglik_by_iteration": [round(v, 6) for v in lls]}

    out = None
    if rank == 0:
        value = tot_solves / dt
        roof = None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            # HBM bytes per algorithmic byte of this kernel, from the committed rocprofv3 PMC passes of this very
            # command (profiles/README.md): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE over all its launches
            with open(tpath) as fh:
                tj = json.load(fh)
            traffic = tj["hbm_bytes_per_alg_byte"] * acc["alg_bytes"] / max(1, acc["launches"])
        if acc["xpass_ms"] > 0:
            achieved = acc["alg_bytes"] / (acc["xpass_ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_xpass_dense<4,4>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": acc["alg_bytes"] / max(1, acc["launches"]),
                    "avg_launch_ms": acc["xpass_ms"] / max(1, acc["launches"]), "launches": acc["launches"],
                    "xpass_share_of_step": round(acc["xpass_ms"] / (dt * 1e3), 4)}
        out = {"metric": "partition Newton-solves/sec (ADMM L2-LR, dense 1Mx1K, 64 partitions/GPU)",
               "value": round(value, 3), "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: synthetic dense %d rows x %d features, %d partitions%s, lambda=1, rho=1, "
                                      "fp32-stored X, fp64 arithmetic" % (rows * N, nf, N, " (64 per GPU)" if world > 1 else ""),
                          "rows": rows * N, "features": nf, "partitions": N, "lambda": [1.0], "rho": [1.0],
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
        if sample:
            threads = os.cpu_count() or 1
            threads = min(threads, len(sample))
            v, pps, cdt, z_orc, cnt_orc = cpu_baseline(sample, nf, args.cpu_iters, threads)
            out["cpu_baseline"] = {"value": round(v, 4), "unit": "solves/s", "cores": threads, "kind": "port",
                                   "sample": "oracle/admm_oracle.c (-O2, fp64, one thread per partition solve) on %d of the %d "
                                             "partitions as its own num.blocks=%d ADMM job, iterations 1..%d from z=0, %.1f s wall"
                                             % (len(sample), N, len(sample), args.cpu_iters, cdt),
                                   "x_passes_ref_per_s": round(pps, 2), "host_cores_available": os.cpu_count()}
            out["gpu_over_cpu"] = {"solves_per_s": round(value / v, 2), "x_passes_ref_per_s": round(tot_pref / dt / pps, 2)}
            out["parity_check"] = parity_on_sample(HipAdmmEngine, sample, nf, args.cpu_iters, local_rank, z_orc, cnt_orc)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
