#!/bin/bash
mkdir -p gpurun_out/r3v
MLX_LIB_PATH=$PWD/tools/libmlease_hip_sprof.so python tools/small_profile.py | tail -2
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3v/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r3v/pytest.log | tail -1
