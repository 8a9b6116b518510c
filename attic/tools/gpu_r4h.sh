#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
D="--no-sparse --no-sweep --no-config1 --loglik-iters 0 --no-cpu-baseline --steps 20 --warmup 5"
for v in 1 2; do
  MLX_BENCH_D8_FIRST=$v timeout 300 python bench.py $D --full-json /tmp/f.json >/dev/null 2>&1; python -c "import json; d=json.load(open('/tmp/f.json')); print('D8_FIRST=$v', d['value'], {k: (d[k].get('value'), d[k].get('ms_per_step')) for k in d if k.startswith('dense_8')})"
done
