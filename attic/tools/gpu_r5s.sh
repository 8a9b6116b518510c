#!/bin/bash
# round 5, GPU call S: the valued (HASVAL) sparse kernels on a configs[2]-size job: row chunk of 128 groups (GPW = 8, spills) against 64
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for ng in 128 64; do
  echo "MLX_ROW_NG=$ng"
  MLX_ROW_NG=$ng timeout 600 python tools/bench_sparse.py --rows 5000000 --partitions 128 --valued --steps 3 --warmup 1 --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('solves_per_s','ms_per_step','ticks_per_step','gen_s','upload_s')})"
done
echo "binary, for scale"
timeout 600 python tools/bench_sparse.py --rows 5000000 --partitions 128 --steps 3 --warmup 1 --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('solves_per_s','ms_per_step','ticks_per_step','gen_s','upload_s')})"
