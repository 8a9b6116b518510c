#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4i/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4i/pytest.log
bash tools/profile_round4.sh r4 2>&1 | tail -40
