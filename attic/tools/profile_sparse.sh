#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the CSR-path development bench (BASELINE configs[2] shape) plain and under
# rocprofv3 --kernel-trace --stats; outputs under gpurun_out/prof_sparse_$1/.
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_sparse_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 400 python $R/tools/bench_sparse.py --steps 3 --warmup 1 --check 4 > $OUT/bench_sparse.json 2> $OUT/bench_sparse.err
timeout 500 rocprofv3 --kernel-trace --stats -d $OUT/kt -o sparse -- python $R/tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/bench_sparse_kt.log 2>&1
timeout 400 python $R/tools/bench_sparse.py --rows 5000000 --partitions 128 --lambdas 0.01,0.1,0.3,1,3,10,30,100 --steps 3 --warmup 1 > $OUT/bench_multilambda.json 2> $OUT/bench_multilambda.err
tail -1 $OUT/bench_sparse.json | cut -c1-900
tail -1 $OUT/bench_multilambda.json | cut -c1-900
cat $OUT/bench_sparse.err | tail -3
