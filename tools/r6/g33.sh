mkdir -p gpurun_out/r6ag
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "norm_tests_decided or reference_order" > gpurun_out/r6ag/pytest.log 2>&1; tail -5 gpurun_out/r6ag/pytest.log
