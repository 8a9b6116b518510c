#!/bin/bash
OUT=gpurun_out/${1:-r3i}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -n "FAILED\|Error" $OUT/pytest.log | head -5
for v in 1 0; do
export MLX_COLD_ROWS=$v
timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$v.json 2> $OUT/c3_$v.err
python - <<PY
import json
d=json.loads(open("$OUT/c3_$v.json").read().strip().splitlines()[-1])
print("cold_rows=$v c3", d["solves_per_s"], d["us_per_tick"], d["ticks_per_step"], d["upload_s"])
PY
done
