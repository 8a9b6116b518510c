#!/bin/bash
# Round-2 GPU check A (run via gpurun): parity tests, then the CSR development bench on the configs[2] shape, the
# configs[3]-per-GPU shape and the 8-lambda shape, plus a rocprofv3 kernel trace. Outputs under gpurun_out/r2a/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2a}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
cd /tmp
timeout 300 python $R/tools/bench_sparse.py --steps 3 --warmup 1 --check 4 > $OUT/sparse_c3.json 2> $OUT/sparse_c3.err
tail -1 $OUT/sparse_c3.json | cut -c1-1200; tail -2 $OUT/sparse_c3.err
timeout 300 python $R/tools/bench_sparse.py --rows 1250000 --partitions 128 --steps 3 --warmup 1 > $OUT/sparse_c4gpu.json 2> $OUT/sparse_c4gpu.err
tail -1 $OUT/sparse_c4gpu.json | cut -c1-1200
timeout 300 python $R/tools/bench_sparse.py --rows 5000000 --partitions 128 --lambdas 0.01,0.1,0.3,1,3,10,30,100 --steps 3 --warmup 1 > $OUT/sparse_8lambda.json 2> $OUT/sparse_8lambda.err
tail -1 $OUT/sparse_8lambda.json | cut -c1-1200
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o sparse -- python $R/tools/bench_sparse.py --steps 3 --warmup 1 > $OUT/sparse_kt.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $OUT/sparse_kernel_trace_stats.txt && head -12 $OUT/sparse_kernel_trace_stats.txt
rm -rf $OUT/kt
