#!/bin/bash
OUT=gpurun_out/${1:-r3q}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "cold or onehot or row_and_cold" > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
for v in 1 0 1 0; do
  export MLX_COLD_SORT=$v
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$v.json 2> $OUT/c3_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/c3_$v.json").read().strip().splitlines()[-1]); print("cold_sort=$v c3", d["solves_per_s"], d["us_per_tick"])
PY
done
