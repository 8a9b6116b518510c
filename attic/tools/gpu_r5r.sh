#!/bin/bash
# round 5, GPU call R: Gram kernel with two operand buffers used in turn (tests + timing)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "posterior or variance or gram or hess" 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/bench_gram.py --reps 5 2>&1 | tail -1; done
timeout 300 python tools/bench_gram.py --reps 5 --rows 39062 --features 2000 2>&1 | tail -1
timeout 300 python tools/bench_gram.py --reps 3 --rows 15001 --features 777 2>&1 | tail -1
