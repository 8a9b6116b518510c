#!/bin/bash
# round 5, GPU call H: reference-order numerics with k_ro_colhot (tests + probe), then the default bench run with its full record
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "order_faithful or tight_epsilon or small_partition" > gpurun_out/r5h_ro_tests.log 2>&1
echo "ro tests rc=$?"; tail -4 gpurun_out/r5h_ro_tests.log
timeout 900 python tools/ro_probe.py 256 4 8 > gpurun_out/r5h_ro_probe.json 2> gpurun_out/r5h_ro_probe.err
echo "probe rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5h_ro_probe.json"))
print(json.dumps({k: d[k] for k in ("solves_per_s_after_first_iteration", "vs_oracle_twin")}))
for k in ("fast", "reference_order"):
    print(k, [x["solves_per_s"] for x in d[k]["per_iteration"]], d[k]["one_stream_profile_of_next_iteration"])
PY
tail -3 gpurun_out/r5h_ro_probe.err
SECONDS=0
timeout 1200 python bench.py --full-json gpurun_out/r5h_bench_default_full.json > gpurun_out/r5h_bench_default.json 2> gpurun_out/r5h_bench_default.err
echo "bench rc=$? in $SECONDS s"
grep -E "\[bench\] leg" gpurun_out/r5h_bench_default.err
cat gpurun_out/r5h_bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5h_bench_default_full.json"))
print("dense RO:", json.dumps(d.get("reference_order")))
sp = d.get("sparse", {})
print("sparse RO:", json.dumps(sp.get("reference_order")))
print("sparse loglik:", json.dumps({k: v for k, v in (sp.get("time_to_ref_loglik") or {}).items() if k != "loglik_by_iteration"}))
print("ingest:", json.dumps(sp.get("ingest")))
print("summary:", json.dumps((sp.get("parity_check") or {}).get("summary")))
print("sweep RO:", json.dumps((d.get("lambda_sweep") or {}).get("reference_order")))
PY
