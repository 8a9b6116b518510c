mkdir -p gpurun_out/r6a
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_order_dense_tiles or scratch_problem_forgets or tight_epsilon or order_faithful or reference_order_numerics_cover" > gpurun_out/r6a/pytest_ro.log 2>&1; tail -5 gpurun_out/r6a/pytest_ro.log
timeout 900 python tools/ro_dense_probe.py 64 8 4 > gpurun_out/r6a/probe64.json 2> gpurun_out/r6a/probe64.err; tail -50 gpurun_out/r6a/probe64.json; tail -5 gpurun_out/r6a/probe64.err
