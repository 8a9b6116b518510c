#!/bin/bash
for ks in 7 14 21 28 35; do
  MLX_GRAM_KS=$ks python tools/bench_gram.py --reps 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ks=$ks', min(d['gram_ms']), d['frac_of_peak_executed'], d['wall_s_incl_host_cholesky'][-1])"
done
