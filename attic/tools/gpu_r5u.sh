#!/bin/bash
# round 5, GPU call U: in-kernel phase times of the reference-order passes (tools/ablate_build.sh pt:-DMLX_PHASE_TIMING)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
RO_ONLY=1 MLX_LIB_PATH=$GRAFT_REPO_ROOT/tools/abl/libmlease_hip_pt.so timeout 600 python tools/ro_probe.py 256 3 1 > gpurun_out/r5u.json 2> gpurun_out/r5u.err
tail -2 gpurun_out/r5u.err
python - <<PY
import json
d = json.load(open("gpurun_out/r5u.json"))
print(d.get("phase_us_sum_over_workgroups"))
print([ (x["solves_per_s"], x.get("ticks")) for x in d["reference_order"]["per_iteration"]])
PY
