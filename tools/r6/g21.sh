mkdir -p gpurun_out/r6u
F="--steps 3 --warmup 1 --rows 65536 --partitions 8 --no-cpu-baseline --no-gram --loglik-iters 3 --test-rows 4096 --sparse-rows 160000 --sparse-partitions 8 --sparse-steps 2 --sparse-warmup 1 --sparse-cpu-sample 0 --sweep-partitions 2 --sweep-steps 1 --sweep-warmup 1 --sweep-cpu-sample 0"
python bench.py $F > gpurun_out/r6u/a.json 2> gpurun_out/r6u/a.err; echo "rc=$?"; grep -v "full record" gpurun_out/r6u/a.err | tail -8 | cut -c1-400
