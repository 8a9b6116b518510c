#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
MLX_TRACE=0 timeout 600 python bench.py --steps 3 --warmup 1 --rows 65536 --partitions 8 --no-cpu-baseline --no-gram --loglik-iters 3 --test-rows 4096 --sparse-rows 160000 --sparse-partitions 8 --sparse-steps 2 --sparse-warmup 1 --sparse-cpu-sample 0 --sweep-partitions 2 --sweep-steps 1 --sweep-warmup 1 --sweep-cpu-sample 0 --full-json gpurun_out/r5e_full.json > gpurun_out/r5e_line.json 2> gpurun_out/r5e.err
echo "rc=$?"
grep -n "\[bench\] leg\|fault\|Error\|error" gpurun_out/r5e.err | head -20
tail -c 600 gpurun_out/r5e.err
