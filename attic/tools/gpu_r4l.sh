#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
D="--no-sparse --no-sweep --no-config1 --no-gram --loglik-iters 0 --no-cpu-baseline --steps 20 --warmup 5"
for v in 0 1 0 1; do
  MLX_BENCH_OWN_STREAM=$v timeout 300 python bench.py $D 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('own_stream=$v dense', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('whole_step_frac'), 'd8', d.get('dense_8_per_gpu'))"
done
for v in 0 1; do
  MLX_BENCH_OWN_STREAM=$v timeout 600 python bench.py --sparse-only --sparse-cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('own_stream=$v sparse', d['value'], d['ms_per_step'], d['whole_step']['frac_of_hbm_peak'])"
  MLX_BENCH_OWN_STREAM=$v timeout 600 python bench.py --sweep-only --sweep-cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('own_stream=$v sweep', d['value'], d['ms_per_step'], d['whole_step']['frac_of_hbm_peak'])"
done
