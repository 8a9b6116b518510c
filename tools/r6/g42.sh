mkdir -p gpurun_out/r6au
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_launch_equals" > gpurun_out/r6au/pytest.log 2>&1; tail -5 gpurun_out/r6au/pytest.log
