#!/usr/bin/env python3
"""Development tool: where a k_solve_small tick of BASELINE configs[0] goes, from shader-clock stamps of workgroup 0 (a library built
with -DMLX_SMALL_PROFILE, passed as MLX_LIB_PATH; see profiles/r3_notes.md). Not part of the product or the tests."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mlease_amd  # noqa: F401
from mlease_amd import admm, hip_engine
from mlease_amd.hip_engine import HipAdmmEngine
from fixtures import load_c1
c1 = load_c1()
lib = ctypes.CDLL(os.environ["MLX_LIB_PATH"])
rd = lib.mlxk_small_prof_read
buf = (ctypes.c_ulonglong * 24)()
for rep in range(3):
    eng = HipAdmmEngine(c1.n_global, [1.0], [1.0], 8)
    for b in c1.blocks:
        eng.add_partition(b)
    eng.finalize()
    rd(buf, 1)
    e, mind, tk, ts = np.float32(0.01), 99999999.0, 0, 0.0
    for it in range(1, 21):
        if it > 1 and mind < 0.001:
            e = np.float32(e / np.float32(10))
        a = time.perf_counter()
        st = eng.solve_local(admm.float_string_roundtrip(e), 1.0)
        ts += time.perf_counter() - a
        mind = eng.consensus_finish().mindiff
        tk += st.ticks
    rd(buf, 0)
    v = list(buf)
    tot = sum(v[:5])
    print("rep %d: solve_local %.2f ms; workgroup 0: %d ticks; cycles per tick: row %.0f, row reduction %.0f, column pass %.0f, step %.0f, wait %.0f, total %.0f (sum over the run %.2f Mcycles)" % (
        rep, ts * 1e3, v[5], v[0] / v[5], v[1] / v[5], v[2] / v[5], v[3] / v[5], v[4] / v[5], tot / v[5], tot / 1e6))
    w = v[8:]
    if w[14] + w[15]:
        print("   step, cycles per CG tick (%d): Hd loop %.0f, its reduction %.0f, s update + norm %.0f, r / d updates + norm %.0f, end %.0f; per EVAL tick (%d): objective + gradient %.0f, rest %.0f" % (
            w[14], w[0] / max(w[14], 1), w[1] / max(w[14], 1), w[2] / max(w[14], 1), w[3] / max(w[14], 1), w[4] / max(w[14], 1), w[15], w[5] / max(w[15], 1), w[6] / max(w[15], 1)))
    eng.close()
