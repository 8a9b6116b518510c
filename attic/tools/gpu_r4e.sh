#!/bin/bash
# round 4, batch e: head kernel for phase A, full bench line with the envelope summary
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
echo "--- sparse leg, head-grid dots on / off"
for v in "MLX_SEQ_DOTS=1" "MLX_SEQ_DOTS=0" "MLX_SEQ_DOTS=1" "MLX_SEQ_DOTS=0"; do
  env $v timeout 600 python bench.py --sparse-only --sparse-cpu-sample 0 --full-json $O/s_$v.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['whole_step']['frac_of_hbm_peak'], [(k['kernel'][:14], k['frac'], k['us_per_tick']) for k in d['roofline']['kernels']])"
done
for v in "MLX_SEQ_DOTS=1" "MLX_SEQ_DOTS=0"; do
  env $v timeout 600 python bench.py --sweep-only --sweep-cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sweep $v', d['value'], d['ms_per_step'], d['whole_step']['frac_of_hbm_peak'], [(k['kernel'][:14], k['frac'], k['us_per_tick']) for k in d['roofline']['kernels']])"
done
echo "--- the driver's command"
timeout 900 python bench.py --steps 20 --warmup 5 --full-json $O/bench_full.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json; cat $O/bench_line.json
echo "--- permutation envelope, 64 partitions, 8 perms"
MLX_SEQ_DOTS=1 timeout 900 python tools/sum_order_experiment.py --partitions 64 --rows 39063 --iters 6 --perms 8 --threads 16 --gpu --minimal --json $O/env64_seq1.json > $O/env64_seq1.log 2>&1; grep -E "gpu:|perm|easy" $O/env64_seq1.log | tail -24
python -c "import json; d=json.load(open('$O/env64_seq1.json')); print('gpu equal', [r['gpu']['equal'] for r in d['per_iteration']], sum(r['gpu']['equal'] for r in d['per_iteration']))"
