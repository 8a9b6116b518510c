#!/bin/bash
# round 5, GPU call A: reference-order numerics on the tick kernels -- parity tests, timing probe, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "order_faithful or tight_epsilon" > gpurun_out/r5a_ro_tests.log 2>&1
echo "ro tests rc=$?" | tee -a gpurun_out/r5a_ro_tests.log
tail -15 gpurun_out/r5a_ro_tests.log
timeout 900 python tools/ro_probe.py 128 3 8 > gpurun_out/r5a_ro_probe.json 2> gpurun_out/r5a_ro_probe.err
echo "probe rc=$?"
tail -60 gpurun_out/r5a_ro_probe.json
tail -5 gpurun_out/r5a_ro_probe.err
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r5a_gpu_tests.log 2>&1
echo "gpu tests rc=$?"
tail -12 gpurun_out/r5a_gpu_tests.log
