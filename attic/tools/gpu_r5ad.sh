#!/bin/bash
# round 5, GPU call AD: two pass workgroups per CU (half-size LDS footprints): hot slice of 9 728 columns (MLX_SLW), row blocks of 10 048 rows (MLX_RBMAX)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
one() { lab=$1; shift; env "$@" timeout 600 python tools/bench_sparse.py --steps 4 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab', d['solves_per_s'], d['us_per_tick'])"; }
one default A=1
one slw9728 MLX_SLW=9728
one slw9728_2hot MLX_SLW=9728 MLX_NHOT=2
one rbmax10048 MLX_RBMAX=10048
one both MLX_SLW=9728 MLX_RBMAX=10048
