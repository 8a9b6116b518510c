#!/bin/bash
# round 5, GPU call Z: the product path on 2 / 3 / 4 tick streams (hardware-queue probe on), sparse and dense
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for ns in 2 3 4; do
  echo "MLX_STREAMS=$ns sparse"
  MLX_STREAMS=$ns MLX_TRACE=0 timeout 600 python tools/bench_sparse.py --steps 4 --warmup 1 --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('solves_per_s','ms_per_step','ticks_per_step')})"
done
for ns in 2 3 4; do
  echo "MLX_STREAMS=$ns dense"
  MLX_STREAMS=$ns timeout 600 python bench.py --steps 10 --warmup 5 --no-sparse --no-sweep --no-cpu-baseline --no-handover --no-dense8 --no-gram --no-config1 --loglik-iters 0 --dense-ro-partitions 0 --no-profile --full-json gpurun_out/r5z_d$ns.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
