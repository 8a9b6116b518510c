#!/bin/bash
# round 3, call W: global-address-space accesses (gld / gst) against the same build with flat accesses (-DMLX_NO_GLOBAL_AS), interleaved
OUT=gpurun_out/${1:-r3w}; mkdir -p $OUT
run() { # tag, lib
  export MLX_LIB_PATH=$2
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$1.json 2> $OUT/c3_$1.err
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep > $OUT/dense_$1.json 2> $OUT/dense_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c3_$1.json").read().strip().splitlines()[-1])
    print("$1 c3", d["solves_per_s"], d["ms_per_step"], d.get("us_per_tick"))
except Exception as e: print("ERR", e); print(open("$OUT/c3_$1.err").read()[-300:])
try:
    d=json.loads(open("$OUT/dense_$1.json").read().strip().splitlines()[-1])
    print("$1 dense", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("replay_ms_per_step"))
except Exception as e: print("ERR", e); print(open("$OUT/dense_$1.err").read()[-300:])
PY
}
run flat1 $PWD/tools/libmlease_hip_flat.so
run glob1 $PWD/ml-ease_amd/csrc/libmlease_hip.so
run flat2 $PWD/tools/libmlease_hip_flat.so
run glob2 $PWD/ml-ease_amd/csrc/libmlease_hip.so
