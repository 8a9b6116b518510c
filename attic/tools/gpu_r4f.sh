#!/bin/bash
# round 4, batch f: full GPU suite incl. the envelope test, then the round's profiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4f
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "follows_the_oracle" 2>&1 | grep -E "solves following|passed|failed" | head
bash tools/profile_round4.sh r4 2>&1 | tail -60
