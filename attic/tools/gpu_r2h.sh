#!/bin/bash
# Round-2 GPU check H: one-hot ADMM spread test + in-kernel phase timing of the sparse passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2h}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -k "onehot_admm_run or cli_end_to_end or row_blocked" -s > $OUT/tests.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error" $OUT/tests.txt | tail -5
grep -n "per iteration" $OUT/tests.txt | head -2 | cut -c1-1200
cd /tmp
for v in pt ptnopf; do
  MLX_LIB_PATH=$R/tools/abl/libmlease_hip_$v.so timeout 300 python $R/tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/c3_$v.json 2> $OUT/c3_$v.err
  MLX_LIB_PATH=$R/tools/abl/libmlease_hip_$v.so timeout 300 python $R/tools/bench_sparse.py --rows 1250000 --partitions 128 --steps 3 --warmup 1 > $OUT/c4_$v.json 2> $OUT/c4_$v.err
  python - <<PY
import json
for n in ("c3_$v","c4_$v"):
    d=json.loads(open("$OUT/%s.json"%n).read().strip().splitlines()[-1])
    print(n, d['solves_per_s'], d['us_per_tick'], d.get('phase_us_sum_over_workgroups'))
PY
done
