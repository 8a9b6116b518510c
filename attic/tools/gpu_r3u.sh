#!/bin/bash
OUT=gpurun_out/${1:-r3u}; mkdir -p $OUT
lscpu | grep -E "Model name" | head -1
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -n "FAILED\|Error" $OUT/pytest.log | head -8
python tools/c1_latency.py 5 oracle
MLX_SMALL_WAVE_STEP=0 python tools/c1_latency.py 3
MLX_NO_SMALL_X=1 python tools/c1_latency.py 3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
