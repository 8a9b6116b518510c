#!/bin/bash
OUT=gpurun_out/${1:-r2cold}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "cold" > $OUT/cold_test.log 2>&1; tail -5 $OUT/cold_test.log
for f in 1 0 1; do
  export MLX_COLD_SEP=$f
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$f.json 2> $OUT/c3_$f.err
  timeout 300 python tools/bench_sparse.py --rows 1250048 --partitions 128 --steps 3 --warmup 1 > $OUT/c4_$f.json 2> $OUT/c4_$f.err
  python - <<PY
import json
for c in ("c3","c4"):
    try:
        d=json.loads(open("$OUT/%s_$f.json"%c).read().strip().splitlines()[-1])
        print("cold_sep=$f", c, d["solves_per_s"], d["us_per_tick"], d["ticks_per_step"], d["cg_per_solve"])
    except Exception as e: print("cold_sep=$f", c, "ERR", e); print(open("$OUT/%s_$f.err"%c).read()[-600:])
PY
done
unset MLX_COLD_SEP
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
