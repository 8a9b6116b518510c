#!/bin/bash
# Round-2 GPU check C: parity tests, CSR development bench on three shapes with row-chunk variants, kernel trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2c}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1
grep -n "passed\|failed" $OUT/tests.txt | tail -3
cd /tmp
run() { # name, env, args
  env $2 timeout 300 python $R/tools/bench_sparse.py $3 --steps 3 --warmup 1 > $OUT/$1.json 2> $OUT/$1.err
  echo "$1 [$2]: $(python - <<PY
import json
try:
    d=json.loads(open('$OUT/$1.json').read().strip().splitlines()[-1]); print(d['solves_per_s'], d['ms_per_step'], d['us_per_tick'])
except Exception as e: print('ERR', e)
PY
)"; tail -2 $OUT/$1.err | cut -c1-300
}
run c3_default "X=1" "--check 4"
run c3_ng64 "MLX_ROW_NG=64" ""
run c4gpu_default "X=1" "--rows 1250000 --partitions 128"
run c4gpu_ng16 "MLX_ROW_NG=16" "--rows 1250000 --partitions 128"
run c4gpu_ng64 "MLX_ROW_NG=64" "--rows 1250000 --partitions 128"
run c4gpu_ng128 "MLX_ROW_NG=128" "--rows 1250000 --partitions 128"
run l8_default "X=1" "--rows 5000000 --partitions 128 --lambdas 0.01,0.1,0.3,1,3,10,30,100"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o sparse -- python $R/tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/sparse_kt.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $OUT/sparse_kernel_trace_stats.txt && head -9 $OUT/sparse_kernel_trace_stats.txt
rm -rf $OUT/kt
