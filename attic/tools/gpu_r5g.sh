#!/bin/bash
# round 5, GPU call G: fold micro-probe, then the default bench run (all legs incl. the new ones) with its full record
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I $R/ml-ease_amd/csrc $R/attic/tools/ro_fold_probe.hip -o /tmp/ro_fold_probe 2>/dev/null && /tmp/ro_fold_probe | tee $R/gpurun_out/r5g_fold_probe.txt
cd $R
/usr/bin/time -v timeout 1200 python bench.py --full-json gpurun_out/r5g_bench_default_full.json > gpurun_out/r5g_bench_default.json 2> gpurun_out/r5g_bench_default.err
echo "bench rc=$?"
grep -E "Elapsed|\[bench\] leg" gpurun_out/r5g_bench_default.err
cat gpurun_out/r5g_bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g_bench_default_full.json"))
print("dense RO:", json.dumps(d.get("reference_order")))
sp = d.get("sparse", {})
print("sparse RO:", json.dumps(sp.get("reference_order")))
print("sparse loglik:", json.dumps({k: v for k, v in (sp.get("time_to_ref_loglik") or {}).items() if k != "loglik_by_iteration"}))
print("ingest:", json.dumps(sp.get("ingest")))
print("summary:", json.dumps((sp.get("parity_check") or {}).get("summary")))
print("sweep RO:", json.dumps((d.get("lambda_sweep") or {}).get("reference_order")))
PY
