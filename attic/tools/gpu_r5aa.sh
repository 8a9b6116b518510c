#!/bin/bash
# round 5, GPU call AA: valued against binary CSR passes, every launch alone on the chip (one tick stream, per-class HIP events)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in "--valued" ""; do
  echo "bench_sparse $v (one stream)"
  MLX_PROFILE_ONE_STREAM=1 timeout 600 python tools/bench_sparse.py --rows 5000000 --partitions 128 $v --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('solves_per_s','us_per_tick','xpass_GBps_alg','nnz')})"
done
