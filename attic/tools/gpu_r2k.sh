#!/bin/bash
# Round-2 GPU check K: parity tests (shared-X lambda sweep), sweep shapes with / without the shared passes, step chunk variants,
# 8-partition dense shape.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2k}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error" $OUT/tests.txt | tail -8
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
L8="0.01,0.1,0.3,1,3,10,30,100"
run l8big_multi "X=1" "--rows 5000000 --partitions 128 --lambdas $L8"
run l8big_nomulti "MLX_NO_MULTI=1" "--rows 5000000 --partitions 128 --lambdas $L8"
run c5gpu_multi "X=1" "--rows 1250000 --partitions 128 --lambdas $L8"
run c5gpu_nomulti "MLX_NO_MULTI=1" "--rows 1250000 --partitions 128 --lambdas $L8"
run c3_ch1024 "MLX_STEP_CH=1024" ""
run c3_ch4096 "MLX_STEP_CH=4096" ""
run c4gpu_ch1024 "MLX_STEP_CH=1024" "--rows 1250000 --partitions 128"
timeout 300 python $R/bench.py --steps 20 --warmup 5 --partitions 8 --rows 125000 --no-sparse --no-cpu-baseline --loglik-iters 0 > $OUT/bench_8part.json 2> $OUT/bench_8part.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_8part.json").read().strip().splitlines()[-1])
print("8 partitions:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["xpass_share_of_step"])
PY
