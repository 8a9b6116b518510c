#!/bin/bash
# round 5, GPU call K: late-tick compaction A/B (MLX_COMPACT=0 / 1) on the sparse leg and the dense headline, then the GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 0 1 0 1; do
  MLX_COMPACT=$c timeout 600 python bench.py --sparse-only --sparse-cpu-sample 0 --sparse-loglik-iters 0 --no-ingest --full-json gpurun_out/r5k_sparse_c$c.json > /dev/null 2> /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r5k_sparse_c$c.json"))
print("compact=$c sparse", d["value"], d["ms_per_step"], d["whole_step"]["frac_of_hbm_peak"], d["last_maxdiff"], d["ticks_per_step"])
PY
done
for c in 0 1; do
  MLX_COMPACT=$c timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep --no-config1 --no-gram --no-dense8 --no-handover --full-json gpurun_out/r5k_dense_c$c.json > /dev/null 2> /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r5k_dense_c$c.json"))
print("compact=$c dense", d["value"], d["ms_per_step"], d["whole_step"]["frac_of_hbm_peak"], d["work"]["last_maxdiff"], d["work"]["z32_sha1_after_timed_steps"])
PY
done
timeout 900 python tools/ro_probe.py 256 4 8 > gpurun_out/r5k_ro_probe.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5k_ro_probe.json"))
print(json.dumps({k: d[k] for k in ("solves_per_s_after_first_iteration", "vs_oracle_twin")}))
PY
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r5k_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/r5k_gpu_tests.log
