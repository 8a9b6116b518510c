#!/bin/bash
# A/B on one box: tools/abl/libmlease_hip_<tag>.so (MLX_LIB_PATH) against the in-tree library, sparse shapes C3 and C4/GPU.
OUT=gpurun_out/${1:-r2ab}; shift
mkdir -p $OUT
for tag in new "$@" new2; do
  if [ $tag = new ] || [ $tag = new2 ]; then unset MLX_LIB_PATH; else export MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_$tag.so; fi
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$tag.json 2> $OUT/c3_$tag.err
  timeout 300 python tools/bench_sparse.py --rows 1250048 --partitions 128 --steps 3 --warmup 1 > $OUT/c4_$tag.json 2> $OUT/c4_$tag.err
  python - <<PY
import json
for c in ("c3","c4"):
    try:
        d=json.loads(open("$OUT/%s_$tag.json"%c).read().strip().splitlines()[-1])
        print("$tag", c, d["solves_per_s"], d["us_per_tick"], d.get("phase_us"))
    except Exception as e: print("$tag", c, "ERR", e)
PY
done
unset MLX_LIB_PATH
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
