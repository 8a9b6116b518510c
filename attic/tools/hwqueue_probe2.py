#!/usr/bin/env python3
"""After hwqueue_probe.py: the library now tests its tick-stream pair for a shared hardware queue (pick_tick_streams). The 8-problem dense
handle with the test off (MLX_NO_STREAM_PROBE=1) and on, alone / beside four live torch streams / beside an idle 64-problem handle."""
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


def build(n, probe=True, stream=None):
    os.environ["MLX_NO_STREAM_PROBE"] = "0" if probe else "1"
    eng = HipAdmmEngine(nf + 1, [1.0], [1.0], n, device=0, stream=stream)
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
    out = []
    for probe in (False, True):
        sys.stderr.write("== %s, probe %s\n" % (label, probe)); sys.stderr.flush()
        e = build(8, probe=probe, **kw)
        out.append(run8(e))
        e.close()
    print("%-52s probe off %5.0f   on %5.0f solves/s" % (label, out[0], out[1]), flush=True)


print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
case("alone")
dummies = [torch.cuda.Stream() for _ in range(4)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()
case("four torch streams alive")
case("four torch streams alive, handle on one of them", stream=dummies[0].cuda_stream)
del dummies
big = build(64)
case("64-problem handle alive (idle)")
s2 = torch.cuda.Stream()
case("64-problem handle alive, handle on a torch stream", stream=s2.cuda_stream)
big2 = build(64)
case("two 64-problem handles alive")
big.close(); big2.close()
case("after they were closed")
