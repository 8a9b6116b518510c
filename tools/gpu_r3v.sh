#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "relaunch or c1_admm or butterflies" 2>&1 | grep -E "passed|failed|^E  " | head -5
