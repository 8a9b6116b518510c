#!/bin/bash
# round 3, call B: GPU test suite with the N>1 tests + the default bench (dense + sparse with parity/cpu legs + lambda sweep)
OUT=gpurun_out/${1:-r3b}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
grep -n "Error\|FAILED" $OUT/pytest.log | head -10
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
tail -5 $OUT/bench_default.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
vo=(d.get("time_to_ref_loglik") or {}).get("vs_oracle_run",{})
print("dense", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v for k,v in vo.items() if k not in ("per_iteration","source")})
print([ (r["iteration"], r["liblinear_epsilon"], r["max_rel_err_z32"], r.get("partitions_with_equal_counters")) for r in vo.get("per_iteration",[])])
sp=d["sparse"]; print("sparse", sp["value"], sp["ms_per_step"], sp["whole_step"], [ (r["us_per_tick"], r["frac"]) for r in sp["roofline"]["kernels"]], sp["roofline"]["measured_in"][-60:])
print(" cpu", sp.get("cpu_baseline"), sp.get("gpu_over_cpu"))
pc=sp.get("parity_check",{}); print(" faithful", pc.get("order_faithful_mode_vs_oracle_twin"))
for r in pc.get("product_path_solve_level",{}).get("per_iteration",[]): print("  ", r)
for k,v in pc.items():
    if k.startswith("product_path_admm"): 
        for r in v: print("  ", r)
sw=d["lambda_sweep"]; print("sweep", sw["value"], sw["ms_per_step"], sw["whole_step"], [ (r["us_per_tick"], r["frac"]) for r in sw["roofline"]["kernels"]])
print(" cpu", sw.get("cpu_baseline"), sw.get("gpu_over_cpu"))
for r in sw.get("parity_check",{}).get("product_path_solve_level",{}).get("per_iteration",[]): print("  ", r)
PY
