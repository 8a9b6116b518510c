#!/bin/bash
# Round-2 GPU check M: parity tests + the three sparse shapes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2m}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error\|Fatal" $OUT/tests.txt | tail -8
cd /tmp
run() { # name, env, args
  env $2 timeout 300 python $R/tools/bench_sparse.py $3 --steps 3 --warmup 1 > $OUT/$1.json 2> $OUT/$1.err
  echo "$1 [$2]: $(python - <<PY
import json
try:
    d=json.loads(open('$OUT/$1.json').read().strip().splitlines()[-1]); print(d['solves_per_s'], d['ms_per_step'], d['us_per_tick'])
except Exception as e: print('ERR', e)
PY
)"; tail -1 $OUT/$1.err | cut -c1-200
}
run c3_default "X=1" ""
run c4gpu_default "X=1" "--rows 1250000 --partitions 128"
run l8_default "X=1" "--rows 5000000 --partitions 128 --lambdas 0.01,0.1,0.3,1,3,10,30,100"
