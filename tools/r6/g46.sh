mkdir -p gpurun_out/r6ay
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_order or one_launch_equals or norm_tests" > gpurun_out/r6ay/pytest.log 2>&1; tail -3 gpurun_out/r6ay/pytest.log
