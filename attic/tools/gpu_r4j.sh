#!/bin/bash
# round 4, batch j: batched commit gather (default), launch-bounds and head-column A/B builds
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4j
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for lib in "" tools/abl/libmlease_hip_ha64.so "" tools/abl/libmlease_hip_ha64.so; do
  MLX_LIB_PATH=${lib:+$R/$lib} timeout 600 python bench.py --sparse-only --sparse-cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib [$lib]', d['value'], d['ms_per_step'], d['whole_step']['frac_of_hbm_peak'], [(k['kernel'][:14], k['frac'], k['us_per_tick']) for k in d['roofline']['kernels']])"
done
for lib in "" tools/abl/libmlease_hip_ha64.so; do
  MLX_LIB_PATH=${lib:+$R/$lib} timeout 600 python bench.py --sweep-only --sweep-cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sweep lib [$lib]', d['value'], d['ms_per_step'], d['whole_step']['frac_of_hbm_peak'], [(k['kernel'][:14], k['frac'], k['us_per_tick']) for k in d['roofline']['kernels']])"
done
echo "--- envelope with 64 head columns in phase A"
MLX_LIB_PATH=$R/tools/abl/libmlease_hip_ha64.so timeout 900 python tools/sum_order_experiment.py --partitions 64 --rows 39063 --iters 6 --perms 4 --threads 16 --gpu --minimal --json $O/env64_ha64.json > $O/env64_ha64.log 2>&1; tail -6 $O/env64_ha64.log
python -c "import json; d=json.load(open('$O/env64_ha64.json')); print('gpu equal', [r['gpu']['equal'] for r in d['per_iteration']], sum(r['gpu']['equal'] for r in d['per_iteration']))"
