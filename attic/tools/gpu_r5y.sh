#!/bin/bash
# round 5, GPU call Y: the full GPU suite, then the round's evidence (tools/profile_round5.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r5y_gpu_tests.log 2>&1
echo "gpu tests rc=$? ($SECONDS s)"; tail -4 gpurun_out/r5y_gpu_tests.log
bash tools/profile_round5.sh r5y > gpurun_out/r5y_profile.log 2>&1
echo "profile rc=$? ($SECONDS s)"; grep -E "^driver-flag|^profiles:|^multirank:" gpurun_out/r5y_profile.log
cat gpurun_out/prof_r5y/bench_driver.json | cut -c1-600
