#!/bin/bash
OUT=gpurun_out/${1:-r3p}; mkdir -p $OUT
for v in 1 0 1 0; do
  export MLX_COLD_SEP=$v
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$v.json 2> $OUT/c3_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/c3_$v.json").read().strip().splitlines()[-1]); print("cold_sep=$v c3", d["solves_per_s"], d["us_per_tick"])
PY
done
