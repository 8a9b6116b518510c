#!/bin/bash
# round 4, batch c: pair-level dense partials, look-back dots with a larger sample
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
D="--no-sparse --no-sweep --no-config1 --no-gram --loglik-iters 0 --no-cpu-baseline --steps 20 --warmup 5"
for v in "MLX_X=0" "MLX_DENSE_UPW=1" "MLX_STREAMS=1"; do
  env $v timeout 300 python bench.py $D --full-json $O/d_$v.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel ms/step', r['kernel_ms_per_step'], 'in flight', r.get('launches_in_flight'), 'by durations', r.get('frac_by_launch_durations'), 'whole', d.get('whole_step_frac'))"
  python -c "import json; d=json.load(open('$O/d_$v.json')); r=d['roofline']; print('   step ms/step', r.get('tron_step_ms_per_step'), r.get('tron_step_busy_ms_per_step'), 'avg launch', r['avg_launch_ms'])"
done
timeout 300 python bench.py $D --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no events', d['value'], d['ms_per_step'], d.get('whole_step_frac'))"
echo "--- 8 problems"
for v in "MLX_X=0" "MLX_DENSE_UPW=2"; do
  env $v timeout 300 python bench.py $D --partitions 8 --rows 125000 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v 8 problems', d['value'], d['ms_per_step'], d.get('whole_step_frac'))"
done
echo "--- sparse leg, look-back dots on / off"
for v in "MLX_SEQ_DOTS=1" "MLX_SEQ_DOTS=0" "MLX_SEQ_DOTS=1" "MLX_SEQ_DOTS=0"; do
  env $v timeout 600 python bench.py --sparse-only --sparse-cpu-sample 0 --full-json $O/s_$v.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['whole_step']['frac_of_hbm_peak'], [(k['kernel'][:14], k['frac'], k['us_per_tick']) for k in d['roofline']['kernels']])"
done
echo "--- permutation envelope, 64 partitions, look-back dots on"
MLX_SEQ_DOTS=1 timeout 900 python tools/sum_order_experiment.py --partitions 64 --rows 39063 --iters 6 --perms 4 --threads 16 --gpu --minimal --json $O/env64_seq1.json > $O/env64_seq1.log 2>&1; grep -E "gpu:|perm" $O/env64_seq1.log | tail -12
python -c "import json; d=json.load(open('$O/env64_seq1.json')); print('gpu equal', [r['gpu']['equal'] for r in d['per_iteration']], sum(r['gpu']['equal'] for r in d['per_iteration']))"
echo "--- off"
MLX_SEQ_DOTS=0 timeout 900 python tools/sum_order_experiment.py --partitions 64 --rows 39063 --iters 6 --perms 4 --threads 16 --gpu --minimal --json $O/env64_seq0.json > $O/env64_seq0.log 2>&1; grep -E "gpu:" $O/env64_seq0.log | tail -6
python -c "import json; d=json.load(open('$O/env64_seq0.json')); print('gpu equal', [r['gpu']['equal'] for r in d['per_iteration']], sum(r['gpu']['equal'] for r in d['per_iteration']))"
