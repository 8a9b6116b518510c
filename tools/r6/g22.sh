mkdir -p gpurun_out/r6v
F="--steps 3 --warmup 1 --rows 65536 --partitions 8 --no-cpu-baseline --no-gram --loglik-iters 3 --test-rows 4096 --sparse-rows 160000 --sparse-partitions 8 --sparse-steps 2 --sparse-warmup 1 --sparse-cpu-sample 0 --sweep-partitions 2 --sweep-steps 1 --sweep-warmup 1 --sweep-cpu-sample 0 --no-config1"
for v in "--no-sparse" "--no-dense-ro" "--no-sparse --no-dense-ro"; do
python bench.py $F $v > gpurun_out/r6v/a.json 2> gpurun_out/r6v/a.err; echo "[$v] rc=$?"; grep -v "full record" gpurun_out/r6v/a.err | tail -2 | cut -c1-200
done
