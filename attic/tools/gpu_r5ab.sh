#!/bin/bash
# round 5, GPU call AB: reference-order column pass, chained sums handed on through slots indexed by the receiving item
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "order_faithful or tight_epsilon or small_partition or reference_order" > gpurun_out/r5ab_ro_tests.log 2>&1
echo "ro tests rc=$?"; tail -4 gpurun_out/r5ab_ro_tests.log
timeout 900 python tools/ro_probe.py 256 4 8 > gpurun_out/r5ab_ro_probe.json 2> gpurun_out/r5ab_ro_probe.err
echo "probe rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5ab_ro_probe.json"))
print(json.dumps({k: d[k] for k in ("solves_per_s_after_first_iteration", "vs_oracle_twin")}))
for k in ("fast", "reference_order"):
    print(k, [x["solves_per_s"] for x in d[k]["per_iteration"]], d[k]["one_stream_profile_of_next_iteration"])
PY
tail -3 gpurun_out/r5ab_ro_probe.err
