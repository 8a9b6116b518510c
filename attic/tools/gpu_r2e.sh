#!/bin/bash
# Round-2 GPU check E: ablation builds of the sparse passes + PMC passes (HBM traffic, LDS) of the sparse bench leg.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2e}
mkdir -p $OUT
export TMPDIR=/tmp
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
for v in nolds noidx none lsu3 lsu10; do
  run abl_$v "MLX_LIB_PATH=$R/tools/abl/libmlease_hip_$v.so" ""
done
run ng32 "MLX_ROW_NG=32" ""
run cunit128k "MLX_CUNIT=131072" ""
run cunit512k "MLX_CUNIT=524288" ""
SP="python $R/bench.py --sparse-only --sparse-cpu-sample 0"
pmc() { # tag, counters
  timeout 400 rocprofv3 --pmc $2 --kernel-trace -d $OUT/pmc_$1 -o sp -- $SP > $OUT/pmc_$1.log 2>&1
  DB=$(find $OUT/pmc_$1 -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB --pmc > $OUT/sparse_pmc_$1.txt
  rm -rf $OUT/pmc_$1
  grep -A12 "^# PMC" $OUT/sparse_pmc_$1.txt | cut -c1-160 | head -14
}
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
pmc lds "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE"
pmc sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
