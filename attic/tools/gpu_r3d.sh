#!/bin/bash
# round 3, call D: two tick streams (MLX_STREAMS=2) on the C3 shape and the 8-lambda per-GPU shape, without profiling events
OUT=gpurun_out/${1:-r3d}; mkdir -p $OUT
for s in 1 3 2 3; do
  export MLX_STREAMS=$s
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 --no-profile > $OUT/c3_s$s.json 2> $OUT/c3_s$s.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c3_s$s.json").read().strip().splitlines()[-1])
    print("streams=$s c3", d["solves_per_s"], d["ms_per_step"], d["ticks_per_step"], d["cg_per_solve"], [r[-1] for r in d["iters"]])
except Exception as e: print("ERR", e); print(open("$OUT/c3_s$s.err").read()[-400:])
PY
done
for s in 3; do
  export MLX_STREAMS=$s
  timeout 300 python tools/bench_sparse.py --rows 1250048 --partitions 128 --lambdas 0.1,0.3,1,3,10,30,100,300 --steps 3 --warmup 1 --no-profile > $OUT/l8_s$s.json 2> $OUT/l8_s$s.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/l8_s$s.json").read().strip().splitlines()[-1])
    print("streams=$s 8lambda", d["solves_per_s"], d["ms_per_step"], d["ticks_per_step"])
except Exception as e: print("ERR", e); print(open("$OUT/l8_s$s.err").read()[-400:])
PY
done
