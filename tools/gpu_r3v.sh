#!/bin/bash
mkdir -p gpurun_out
for v in v7 v8; do MLX_LIB_PATH=$PWD/tools/libmlease_hip_$v.so python tools/dbg_probs.py $v 2>&1 | tail -2; done
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/probs_v7.npz", allow_pickle=True); b=np.load("gpurun_out/probs_v8.npz", allow_pickle=True)
names=["f","delta","gnorm","gnorm1","eps","rTr","cgtol","prered","gs"]
print("ints equal:", np.array_equal(a["ints"], b["ints"]))
print(a["ints"][:, [2,4,7,8,9,10]])
for q in range(10):
    print(q, " ".join("%s %.3e" % (n, abs(a["dbl"][q,i]-b["dbl"][q,i])/max(abs(b["dbl"][q,i]),1e-300)) for i,n in enumerate(names)))
PY
