#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the fp64-MFMA Gram kernel (k_gram_f64) under rocprofv3 -- kernel trace + two PMC passes
# (SQ counters only; --pmc never together with the sys/runtime trace domains). Outputs under gpurun_out/prof_gram_$1/;
# tools/rocpd_summary.py turns the databases into the text files committed under profiles/.
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_gram_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/bench_gram.py --reps 5"
timeout 300 $CMD > $OUT/bench_gram.json 2> $OUT/bench_gram.err; tail -1 $OUT/bench_gram.json
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o gram -- $CMD > $OUT/kt.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o gram -- $CMD > $OUT/pmc1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $OUT/pmc2 -o gram -- $CMD > $OUT/pmc2.log 2>&1
for d in kt pmc1 pmc2; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $([ $d != kt ] && echo --pmc) > $OUT/$d.txt 2>&1
done
grep -E "k_gram|kernel " $OUT/kt.txt | head -5
grep -E "k_gram_f64" $OUT/pmc1.txt | head -12
grep -E "k_gram_f64" $OUT/pmc2.txt | head -12
tail -3 $OUT/pmc1.log
