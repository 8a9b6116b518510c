#!/usr/bin/env python3
"""Why does an 8-partition dense engine built AFTER the 64-partition engine stream 35 % slower? (round 4, profiles/r4_notes.md)
Builds the 8-partition engine (a) first, (b) after a 64-partition engine that is still alive, (c) after it was closed, (d) after it was
closed and torch's cache emptied, (e) alive but the 8-engine on its own (non-NULL) stream; prints solves/s of 25 ADMM iterations each."""
import os, sys, time
import numpy as np
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


def build(n, stream):
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


null = torch.cuda.current_stream().cuda_stream
e8 = build(8, null); print("(a) first                         %.0f solves/s" % run8(e8)); e8.close()
big = build(64, null)
e8 = build(8, null); print("(b) after the 64-engine, alive    %.0f" % run8(e8)); e8.close()
s2 = torch.cuda.Stream()
e8 = build(8, s2.cuda_stream); print("(e) alive, 8-engine on own stream %.0f" % run8(e8)); e8.close()
e8 = build(8, None); print("(f) alive, library-owned stream   %.0f" % run8(e8)); e8.close()
big.close()
e8 = build(8, null); print("(c) after it was closed           %.0f" % run8(e8)); e8.close()
torch.cuda.empty_cache()
e8 = build(8, null); print("(d) closed + empty_cache          %.0f" % run8(e8)); e8.close()
