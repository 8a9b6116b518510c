#!/usr/bin/env python3
"""Tick streams for DENSE problem lists (round 4): the 8-per-GPU shape and the 64-partition engine with MLX_STREAMS = 1 .. 4
(the library default is 2; round 3 measured 2 / 3 / 4 on the sparse leg only). 25 ADMM iterations of the driver's schedule, the last 20 timed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import mlease_amd  # noqa
from mlease_amd import admm
from mlease_amd.hip_engine import HipAdmmEngine
import synth_data as sd
import bench

dev = torch.device("cuda", 0)
rows, nf = 15625, 1000


def build(n):
    eng = HipAdmmEngine(nf + 1, [1.0], [1.0], n, device=0, stream=None)
    for k in range(n):
        X, y = sd.dense_rows_torch(torch, dev, k * (64 // n), rows, nf, stride=64)
        torch.cuda.synchronize()
        eng.add_partition_dense_device(k, X.data_ptr(), rows, nf, nf, y.data_ptr())
        del X, y
    eng.finalize()
    return eng


def run(eng):
    sched = bench.EpsSchedule(admm)
    solves = 0
    for it in range(25):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        st = eng.solve_local(sched.next(), 1.0)
        sched.mindiff = eng.consensus_finish().mindiff
        if it >= 5:
            solves += st.solves
    torch.cuda.synchronize()
    return solves / (time.perf_counter() - t0), eng.z()[0].tobytes()


for n in (8, 64, 16, 32):
    ref = None
    for rep in range(2):
        line = []
        for ns in (1, 2, 3, 4):
            os.environ["MLX_STREAMS"] = str(ns)
            eng = build(n)
            v, zb = run(eng)
            eng.close()
            ref = ref or zb
            line.append("%d streams %6.0f%s" % (ns, v, "" if zb == ref else " (z differs!)"))
        print("n=%2d  " % n + "   ".join(line), flush=True)
