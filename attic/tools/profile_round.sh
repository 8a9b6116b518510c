#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel trace + PMC passes (separate runs, as the guide
# prescribes), outputs under gpurun_out/prof_$1/. Summaries are produced afterwards with tools/rocpd_summary.py.
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --loglik-iters 0"
python $R/bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $R/bench.py $ARGS > $OUT/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- python $R/bench.py $ARGS --no-profile > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- python $R/bench.py $ARGS --no-profile > $OUT/bench_pmc_write.log 2>&1
tail -1 $OUT/bench_default.json | cut -c1-3000
