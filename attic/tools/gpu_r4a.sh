#!/bin/bash
# round 4, batch a: GPU tests, the driver's bench command, pipeline A/B, permutation-envelope experiment with the GPU in it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --full-json $O/bench_full.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json; cat $O/bench_line.json
D="--no-sparse --no-sweep --no-config1 --no-gram --loglik-iters 0 --no-cpu-baseline --steps 20 --warmup 5"
for v in "MLX_DENSE_PIPE=1" "MLX_DENSE_PIPE=0" "MLX_DENSE_PARTS=3" "MLX_DENSE_PARTS=4" "MLX_STREAMS=1"; do
  for rep in 1 2; do
    env $v timeout 300 python bench.py $D --full-json $O/d_${v}_$rep.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d.get('whole_step_frac'))"
  done
done
echo "--- no events in the timed region"
for v in "MLX_DENSE_PIPE=1" "MLX_DENSE_PIPE=0"; do
  env $v timeout 300 python bench.py $D --no-profile --full-json $O/dn_${v}.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v noprofile', d['value'], d['ms_per_step'], d.get('whole_step_frac'))"
done
echo "--- 8 problems per GPU (the 8-GPU share)"
for v in "MLX_DENSE_PIPE=1" "MLX_DENSE_PIPE=0" "MLX_DENSE_PARTS=4"; do
  env $v timeout 300 python bench.py $D --partitions 8 --rows 125000 --no-profile --full-json $O/d8_${v}.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v 8 problems', d['value'], d['ms_per_step'], d.get('whole_step_frac'))"
  env $v timeout 300 python bench.py $D --partitions 8 --rows 125000 --full-json $O/d8p_${v}.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v 8 problems, events', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
echo "--- permutation envelope with the GPU"
timeout 900 python tools/sum_order_experiment.py --partitions 16 --rows 39063 --iters 6 --perms 8 --threads 16 --gpu --json $O/sum_order_gpu.json > $O/sum_order_gpu.log 2>&1; tail -30 $O/sum_order_gpu.log
