#!/bin/bash
OUT=gpurun_out/${1:-r2nhot}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
for f in 1 2 3 4 2 1; do
  export MLX_NHOT=$f
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$f.json 2> $OUT/c3_$f.err
  timeout 300 python tools/bench_sparse.py --rows 1250048 --partitions 128 --steps 3 --warmup 1 > $OUT/c4_$f.json 2> $OUT/c4_$f.err
  python - <<PY
import json
for c in ("c3","c4"):
    try:
        d=json.loads(open("$OUT/%s_$f.json"%c).read().strip().splitlines()[-1])
        print("nhot=$f", c, d["solves_per_s"], d["us_per_tick"], d["ticks_per_step"])
    except Exception as e: print("nhot=$f", c, "ERR", e); print(open("$OUT/%s_$f.err"%c).read()[-600:])
PY
done
