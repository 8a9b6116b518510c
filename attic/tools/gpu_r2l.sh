#!/bin/bash
# Round-2 GPU check L: parity tests; column-pass batching; step XB variants; dense 8-partition chunk targets.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2l}
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
run c3_default "X=1" ""
run c4gpu_default "X=1" "--rows 1250000 --partitions 128"
run l8_default "X=1" "--rows 5000000 --partitions 128 --lambdas 0.01,0.1,0.3,1,3,10,30,100"
for v in xb8 xb2; do
  run c3_$v "MLX_LIB_PATH=$R/tools/abl/libmlease_hip_$v.so" ""
  run c4gpu_$v "MLX_LIB_PATH=$R/tools/abl/libmlease_hip_$v.so" "--rows 1250000 --partitions 128"
done
for w in 1024 512 256; do
  MLX_DENSE_WGS=$w timeout 300 python $R/bench.py --steps 20 --warmup 5 --partitions 8 --rows 125000 --no-sparse --no-cpu-baseline --loglik-iters 0 > $OUT/bench_8part_$w.json 2> $OUT/bench_8part_$w.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_8part_$w.json").read().strip().splitlines()[-1])
print("8 partitions, target $w workgroups:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["xpass_share_of_step"])
PY
done
