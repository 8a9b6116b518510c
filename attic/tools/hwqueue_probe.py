#!/usr/bin/env python3
"""Round 4 blamed the slow 8-problem engine beside a 64-problem engine on "both handles on the legacy default stream". Wrong premise:
torch's default stream is the NULL pointer, and mlx_set_stream(h, NULL) means "own stream" -- it destroyed the handle's stream and
created a new one. This probe separates the candidates: re-created stream, number of live streams in the process (HIP maps streams
onto a limited number of hardware queues: GPU_MAX_HW_QUEUES), legacy stream for real."""
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
KEEP = "keep"


def build(n, stream=KEEP):
    eng = HipAdmmEngine(nf + 1, [1.0], [1.0], n, device=0, stream=None)
    if stream != KEEP:
        eng._ck(eng.L.mlx_set_stream(eng.h, stream))
    for k in range(n):
        X, y = sd.dense_rows_torch(torch, dev, k * (64 // n), rows, nf, stride=64)
        torch.cuda.synchronize()
        eng.add_partition_dense_device(k, X.data_ptr(), rows, nf, nf, y.data_ptr())
        del X, y
    eng.finalize()
    return eng


def run8(eng):
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
    return solves / (time.perf_counter() - t0)


def case(label, **kw):
    e = build(8, **kw)
    print("%-64s %5.0f solves/s" % (label, run8(e)), flush=True)
    e.close()


print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
case("(a) alone, the stream mlx_create made")
case("(a') alone, mlx_set_stream(NULL): stream re-created", stream=None)
dummies = [torch.cuda.Stream() for _ in range(4)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()
case("(e) four idle torch streams alive, stream as created")
case("(f) four idle torch streams alive, stream re-created", stream=None)
del dummies
big = build(64)
case("(b) 64-problem handle alive (idle), stream as created")
case("(c) 64-problem handle alive, stream re-created", stream=None)
s2 = torch.cuda.Stream()
case("(d) 64-problem handle alive, a torch stream", stream=s2.cuda_stream)
big.close()
case("(g) after it was closed, stream re-created", stream=None)
