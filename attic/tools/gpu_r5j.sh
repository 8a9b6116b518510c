#!/bin/bash
# round 5, GPU call J: whole GPU suite, then the round's profiles (tools/profile_round5.sh r5)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r5j_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/r5j_gpu_tests.log
bash tools/profile_round5.sh r5 2>&1 | tail -120
