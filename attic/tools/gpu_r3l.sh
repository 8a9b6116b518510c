#!/bin/bash
# round 3: the layout knobs again with the re-ordered rows / cold columns and the streaming step (C3 shape, events on, one stream)
OUT=gpurun_out/${1:-r3l}; mkdir -p $OUT
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["solves_per_s"], d["us_per_tick"])
except Exception as e: print("$tag ERR", e)
PY
}
run base A=1
run cunit128k MLX_CUNIT=131072
run cunit512k MLX_CUNIT=524288
run stepch1024 MLX_STEP_CH=1024
run stepch4096 MLX_STEP_CH=4096
run rowng64 MLX_ROW_NG=64
run seg128 MLX_SEG=128
run rbmax MLX_RBMAX=13056
