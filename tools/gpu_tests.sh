#!/bin/bash
OUT=gpurun_out/${1:-tests}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -n "FAILED\|Error\|assert" $OUT/pytest.log | head -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
