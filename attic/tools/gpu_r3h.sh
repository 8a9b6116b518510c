#!/bin/bash
OUT=gpurun_out/${1:-r3h}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "cold" > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
for rep in 1 2; do
timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$rep.json 2> $OUT/c3_$rep.err
python - <<PY
import json
d=json.loads(open("$OUT/c3_$rep.json").read().strip().splitlines()[-1])
print("c3", d["solves_per_s"], d["us_per_tick"], d["ticks_per_step"])
PY
done
