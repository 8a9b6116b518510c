#!/bin/bash
# round 3, call C: step launches grouped by problems (MLX_STEP_GROUP) on the C3 shape
OUT=gpurun_out/${1:-r3c}; mkdir -p $OUT
for g in 0 32 58 64 128; do
  export MLX_STEP_GROUP=$g
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_g$g.json 2> $OUT/c3_g$g.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c3_g$g.json").read().strip().splitlines()[-1])
    print("step_group=$g c3", d["solves_per_s"], d["us_per_tick"], d["ticks_per_step"], d["cg_per_solve"])
except Exception as e: print("ERR", e); print(open("$OUT/c3_g$g.err").read()[-400:])
PY
done
