#!/bin/bash
python tools/c1_latency.py 7
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "c1 or relaunch or solve_one or mean_model or sparse_absent or ill_conditioned or ragged or degenerate" 2>&1 | grep -E "passed|failed" | tail -1
