#!/usr/bin/env python3
"""Where does one ADMM iteration's wall time go on a few-problem dense handle (the 8-GPU share of config #2: 8 partitions of
15 625 x 1000)?  solve_local wall vs its device time, consensus_finish wall."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
import mlease_amd  # noqa: F401
from mlease_amd.hip_engine import HipAdmmEngine
import synth_data as sd
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prof = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
nf, rows = 1000, 15625
dev = torch.device("cuda", 0)
eng = HipAdmmEngine(nf + 1, [1.0], [1.0], P, device=0, stream=torch.cuda.current_stream().cuda_stream, profiling=prof)
for k in range(P):
    X, y = sd.dense_rows_torch(torch, dev, k, rows, nf, stride=P)
    torch.cuda.synchronize()
    eng.add_partition_dense_device(k, X.data_ptr(), rows, nf, nf, y.data_ptr())
    del X, y
eng.finalize()
eps = 0.01
for it in range(12):
    if it > 4: eps = max(eps / 10, 1e-12)
    t0 = time.perf_counter(); st = eng.solve_local(eps, 1.0); t1 = time.perf_counter()
    fin = eng.consensus_finish(); t2 = time.perf_counter()
    print("it %2d eps %.0e ticks %3d solve_local %.3f ms (device total %.3f, xpass busy %.3f, step busy %.3f) finish %.3f ms" % (
        it + 1, eps, st.ticks, (t1 - t0) * 1e3, st.total_ms, st.xpass_busy_ms, st.step_busy_ms, (t2 - t1) * 1e3))
