#!/bin/bash
# round 4, batch k: 64 head columns as the default; final tests + envelope + profiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4k
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "follows_the_oracle" 2>&1 | grep -E "solves following|passed|failed" | head -3
timeout 900 python tools/sum_order_experiment.py --partitions 64 --rows 39063 --iters 6 --perms 8 --threads 16 --gpu --minimal --json $O/env64.json > $O/env64.log 2>&1; tail -12 $O/env64.log
python -c "import json; d=json.load(open('$O/env64.json')); print('gpu equal', [r['gpu']['equal'] for r in d['per_iteration']], sum(r['gpu']['equal'] for r in d['per_iteration']), 'easy', sum(r['easy_solves'] for r in d['per_iteration']), 'gpu on easy', sum(r['gpu_equal_on_easy_solves'] for r in d['per_iteration']))"
bash tools/profile_round4.sh r4 2>&1 | tail -32 | cut -c1-260
