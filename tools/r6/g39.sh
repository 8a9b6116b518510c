mkdir -p gpurun_out/r6ap
timeout 1500 python -m pytest tests/test_native_host.py tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r6ap/pytest.log 2>&1; tail -4 gpurun_out/r6ap/pytest.log
bash tools/e2e_cli.sh > gpurun_out/r6ap/e2e.log 2>&1; tail -12 gpurun_out/r6ap/e2e.log
