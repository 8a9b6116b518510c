#!/bin/bash
# round 5, GPU call AC: full GPU suite + the driver-flag bench on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof_r5ac
export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r5ac_gpu_tests.log 2>&1
echo "gpu tests rc=$? ($SECONDS s)"; tail -4 gpurun_out/r5ac_gpu_tests.log | head -2
timeout 900 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/prof_r5ac/bench_driver_full.json > gpurun_out/prof_r5ac/bench_driver.json 2> gpurun_out/prof_r5ac/bench_driver.err
echo "bench rc=$? ($SECONDS s)"
cut -c1-300 gpurun_out/prof_r5ac/bench_driver.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5ac/kt_ro -o b -- python $GRAFT_REPO_ROOT/tools/ro_probe.py 256 3 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r5ac/kt_ro.log
DB=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_r5ac/kt_ro -name '*.db' | head -1)
[ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --busy k_rowpass_lds --busy k_colpass_lds --busy k_ro_step > $GRAFT_REPO_ROOT/gpurun_out/prof_r5ac/ro_kernel_trace.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r5ac/kt_ro $GRAFT_REPO_ROOT/gpurun_out/prof_r5ac/kt_ro.log
head -6 $GRAFT_REPO_ROOT/gpurun_out/prof_r5ac/ro_kernel_trace.txt | cut -c1-140
