mkdir -p gpurun_out/r6s
F="--steps 3 --warmup 1 --rows 65536 --partitions 8 --no-cpu-baseline --no-gram --loglik-iters 3 --test-rows 4096 --no-sparse --no-sweep --no-config1"
python bench.py $F --no-dense-ro > gpurun_out/r6s/a.json 2> gpurun_out/r6s/a.err; echo "no-dense-ro rc=$?"; tail -2 gpurun_out/r6s/a.err | cut -c1-300
python bench.py $F > gpurun_out/r6s/b.json 2> gpurun_out/r6s/b.err; echo "with dense-ro rc=$?"; tail -3 gpurun_out/r6s/b.err | cut -c1-400
python tools/ro_dense_probe.py 8 3 0 8192 1000 > gpurun_out/r6s/p.json 2> gpurun_out/r6s/p.err; echo "probe rc=$?"; tail -3 gpurun_out/r6s/p.err | cut -c1-300
