#!/bin/bash
# Round-2 GPU check N: parity tests + phase timing of the step kernels.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2n}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error\|Fatal" $OUT/tests.txt | tail -8
cd /tmp
for n in c3 c4; do continue;
  if [ $n = c3 ]; then A=""; else A="--rows 1250000 --partitions 128"; fi
  MLX_LIB_PATH=$R/tools/abl/libmlease_hip_pt.so timeout 300 python $R/tools/bench_sparse.py $A --steps 3 --warmup 1 > $OUT/${n}_pt.json 2> $OUT/${n}_pt.err
  python - <<PY
import json
d=json.loads(open("$OUT/${n}_pt.json").read().strip().splitlines()[-1])
print("$n", d['solves_per_s'], d['us_per_tick'], d.get('phase_us_sum_over_workgroups'))
PY
done
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
run c3_ch1024 "MLX_STEP_CH=1024" ""
run c3_ch4096 "MLX_STEP_CH=4096" ""
