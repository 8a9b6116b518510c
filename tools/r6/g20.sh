mkdir -p gpurun_out/r6t
F="--sparse-rows 160000 --sparse-partitions 8 --sparse-steps 2 --sparse-warmup 1 --sparse-cpu-sample 0 --sweep-partitions 2 --sweep-steps 1 --sweep-warmup 1 --sweep-cpu-sample 0 --no-cpu-baseline"
python bench.py --sparse-only $F > gpurun_out/r6t/a.json 2> gpurun_out/r6t/a.err; echo "sparse-only rc=$?"; grep -v "full record" gpurun_out/r6t/a.err | tail -4 | cut -c1-300
python bench.py --sweep-only $F > gpurun_out/r6t/b.json 2> gpurun_out/r6t/b.err; echo "sweep-only rc=$?"; grep -v "full record" gpurun_out/r6t/b.err | tail -4 | cut -c1-300
