#!/bin/bash
OUT=gpurun_out/${1:-r3k}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or onehot or cold or sweep" > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
export TMPDIR=/tmp; R=$(pwd)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt -o sparse -- python $R/tools/bench_sparse.py --steps 3 --warmup 1 > $R/$OUT/kt.log 2>&1)
python tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) | head -9
timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3.json 2> $OUT/c3.err
python - <<PY
import json
d=json.loads(open("$OUT/c3.json").read().strip().splitlines()[-1])
print("c3", d["solves_per_s"], d["us_per_tick"], d["ticks_per_step"])
PY
