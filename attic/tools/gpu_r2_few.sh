#!/bin/bash
# the per-GPU shapes of the strong-scaled dense job (64 partitions over 8 / 4 / 2 GPUs) on one GPU: bench line + kernel trace
OUT=$PWD/gpurun_out/${1:-r2few}; mkdir -p $OUT
R=$PWD
export TMPDIR=/tmp
cd /tmp
for P in 8 16 32; do
  rows=$((15625 * P))
  CMD="python $R/bench.py --steps 20 --warmup 5 --partitions $P --rows $rows --no-sparse --no-cpu-baseline --loglik-iters 0 --no-gram"
  timeout 300 $CMD > $OUT/b$P.json 2> $OUT/b$P.err
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt$P -o b -- $CMD > $OUT/kt$P.log 2>&1
  DB=$(find $OUT/kt$P -name '*.db' | head -1); [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $OUT/kt$P.txt; rm -rf $OUT/kt$P $OUT/kt$P.log
  python - <<PY
import json
d=json.loads(open("$OUT/b$P.json").read().strip().splitlines()[-1])
print("P=$P", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["xpass_share_of_step"], d["work"]["ticks"])
PY
  grep -E "k_xpass_dense|k_tron_step|k_setup|k_outputs|k_partial|k_z_update|k_u_update|copyBuffer|fillBuffer" $OUT/kt$P.txt | cut -c1-140
done
