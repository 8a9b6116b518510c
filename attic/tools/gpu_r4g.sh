#!/bin/bash
# round 4, batch g: 8-per-GPU leg oddity, faster head kernel, final profiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4g
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
D="--no-sparse --no-sweep --no-config1 --loglik-iters 0 --no-cpu-baseline --steps 20 --warmup 5"
for extra in "" "--no-gram"; do
  timeout 300 python bench.py $D $extra --full-json $O/d8_$extra.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense+dense8 [$extra]', d['value'], d['ms_per_step'], d.get('dense_8_per_gpu'))"
done
timeout 300 python bench.py $D --no-gram --no-dense8 --partitions 8 --rows 125000 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('standalone 8 problems', d['value'], d['ms_per_step'], d.get('whole_step_frac'))"
for v in "MLX_SEQ_DOTS=1" "MLX_SEQ_DOTS=0"; do
  env $v timeout 600 python bench.py --sparse-only --sparse-cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['whole_step']['frac_of_hbm_peak'], [(k['kernel'][:14], k['frac'], k['us_per_tick']) for k in d['roofline']['kernels']])"
done
bash tools/profile_round4.sh r4 2>&1 | tail -45
