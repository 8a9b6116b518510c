#!/bin/bash
# round 3, final call 2: rocprofv3 kernel traces of the dense and the sparse leg on the final code (+ the two short bench lines they trace)
set -u
TAG=${1:-r3y}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DENSE="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep"
SPARSE="python $R/bench.py --sparse-only --sparse-cpu-sample 0"
timeout 300 $DENSE > $OUT/bench_dense_short.json 2> /dev/null
timeout 300 $SPARSE > $OUT/bench_sparse_only.json 2> /dev/null
summ() { DB=$(find $1 -name '*.db' | head -1); [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $2; rm -rf $1; }
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_d -o b -- $DENSE > $OUT/kt_dense.log 2>&1;   summ $OUT/kt_d $OUT/dense_kernel_trace.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_s -o b -- $SPARSE > $OUT/kt_sparse.log 2>&1; summ $OUT/kt_s $OUT/sparse_kernel_trace.txt
rm -f $OUT/*.log
head -8 $OUT/dense_kernel_trace.txt; head -12 $OUT/sparse_kernel_trace.txt
