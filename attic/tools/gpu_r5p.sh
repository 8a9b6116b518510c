#!/bin/bash
# round 5, GPU call P: grid-rounded dots of the small solver, cheaper form; the one-launch against ticks test under both contracts
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r5p_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -15 gpurun_out/r5p_gpu_tests.log
for i in 1 2; do
timeout 600 python bench.py --steps 2 --warmup 1 --no-sparse --no-sweep --no-cpu-baseline --no-handover --no-dense8 --no-gram --loglik-iters 0 --dense-ro-partitions 0 --no-profile --full-json gpurun_out/r5p_c1_$i.json > gpurun_out/r5p_c1_$i.line 2> gpurun_out/r5p_c1_$i.err
echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("gpurun_out/r5p_c1_$i.json"))
print({k: v for k, v in d.items() if "config1" in k or "latency" in k})
PY
done
