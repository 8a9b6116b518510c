#!/bin/bash
OUT=gpurun_out/${1:-r3g}; mkdir -p $OUT; R=${GRAFT_REPO_ROOT:-$(pwd)}
python tools/c1_latency.py 5 oracle 2>&1 | tee $OUT/c1.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt -o c1 -- python $R/tools/c1_latency.py 1 > $R/$OUT/kt.log 2>&1)
python tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) | head -14
