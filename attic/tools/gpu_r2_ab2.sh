#!/bin/bash
# A/B on one box: in-tree library vs tools/abl/libmlease_hip_<tag>.so, C3 and C4/GPU, optional MLX_ROW_NG settings
OUT=gpurun_out/${1:-r2ab2}; shift
mkdir -p $OUT
run() { # label
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$1.json 2> $OUT/c3_$1.err
  timeout 300 python tools/bench_sparse.py --rows 1250048 --partitions 128 --steps 3 --warmup 1 > $OUT/c4_$1.json 2> $OUT/c4_$1.err
  python - <<PY
import json
for c in ("c3","c4"):
    try:
        d=json.loads(open("$OUT/%s_$1.json"%c).read().strip().splitlines()[-1])
        print("$1", c, d["solves_per_s"], d["us_per_tick"])
    except Exception as e: print("$1", c, "ERR", e)
PY
}
unset MLX_LIB_PATH; run base
for tag in "$@"; do
  export MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_$tag.so; run $tag
  MLX_ROW_NG=64 run ${tag}_ng64
done
unset MLX_LIB_PATH; run base2
