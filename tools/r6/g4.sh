mkdir -p gpurun_out/r6d
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_order_dense_tiles or scratch_problem_forgets or tight_epsilon or order_faithful or reference_order_numerics_cover or limits_lifted or posterior" > gpurun_out/r6d/pytest_ro.log 2>&1; tail -5 gpurun_out/r6d/pytest_ro.log
RO_FAST=1 timeout 900 python tools/ro_dense_probe.py 64 10 2 > gpurun_out/r6d/probe64.json 2> gpurun_out/r6d/probe64.err; python - <<PY
import json
dd=json.load(open("gpurun_out/r6d/probe64.json"))
for k in ("fast","reference_order"):
    d=dd[k]; p=d["one_stream_profile"]
    print(k, "streams", d["tick_streams"], "solves/s per it", [x["solves_per_s"] for x in d["per_iteration"]], "ticks", [x["ticks"] for x in d["per_iteration"]], "us/tick", p.get("us_per_tick"), "frac", p.get("row_frac_of_hbm_peak"), p.get("col_frac_of_hbm_peak"), p.get("xpass_frac_of_hbm_peak"))
print(dd.get("vs_oracle_twin"))
PY
