#!/bin/bash
# round 5, GPU call X: the tick log (test) and the sparse leg's active-set histogram
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tick_log or valued or absent_features" 2>&1 | tail -3
timeout 900 python bench.py --sparse-only --no-cpu-baseline --no-ingest --sparse-loglik-iters 0 --sparse-cpu-sample 0 --full-json gpurun_out/r5x_sparse.json > gpurun_out/r5x.line 2> gpurun_out/r5x.err
echo "bench rc=$?"; tail -3 gpurun_out/r5x.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5x_sparse.json"))
sp = d.get("sparse", d)
print("value", sp.get("value"), sp.get("ms_per_step"))
h = (sp.get("roofline") or {}).get("active_histogram")
print(json.dumps(h, indent=0))
PY
