#!/bin/bash
# round 3, call A: cold-column ordering A/B on the C3 shape + kernel trace of the new layout + the GPU test suite
OUT=gpurun_out/${1:-r3a}; mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in 1 0; do
  if [ $f = 1 ]; then export MLX_NO_COLD_ORDER=1; else unset MLX_NO_COLD_ORDER; fi
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 --check 4 > $OUT/c3_noorder$f.json 2> $OUT/c3_noorder$f.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c3_noorder$f.json").read().strip().splitlines()[-1])
    print("no_cold_order=$f c3", d["solves_per_s"], d["us_per_tick"], d["ticks_per_step"], d["cg_per_solve"])
except Exception as e: print("ERR", e)
print(open("$OUT/c3_noorder$f.err").read()[-400:])
PY
done
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt -o sparse -- python $R/tools/bench_sparse.py --steps 3 --warmup 1 > $R/$OUT/kt.log 2>&1)
find $OUT/kt -name "*kernel_stats*" | head -1 | xargs -r head -20
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
