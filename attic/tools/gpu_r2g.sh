#!/bin/bash
# Round-2 GPU check G: parity tests, row-pass variants (packs in flight / staging prefetch) on two shapes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2g}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error" $OUT/tests.txt | tail -8
grep -n "per iteration" $OUT/tests.txt | head -3 | cut -c1-700
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
for v in kp2 kp2nopf kp3nopf kp1nopf; do
  run c3_$v "MLX_LIB_PATH=$R/tools/abl/libmlease_hip_$v.so" ""
  run c4gpu_$v "MLX_LIB_PATH=$R/tools/abl/libmlease_hip_$v.so" "--rows 1250000 --partitions 128"
done
run c4gpu_ng16 "MLX_ROW_NG=16" "--rows 1250000 --partitions 128"
run c4gpu_ng64 "MLX_ROW_NG=64" "--rows 1250000 --partitions 128"
