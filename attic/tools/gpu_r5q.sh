#!/bin/bash
# round 5, GPU call Q: Gram kernel timing probes (tools/bench_gram.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do timeout 300 python tools/bench_gram.py --reps 5 2>&1 | tail -1; done
timeout 300 python tools/bench_gram.py --reps 5 --rows 39062 --features 2000 2>&1 | tail -1
