#!/bin/bash
# round 3, final call 1: the GPU test suite, smoke(), and bench.py with the driver's flags, on the final code
OUT=gpurun_out/${1:-r3x}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -n "^FAILED" $OUT/pytest.log | head -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print("driver-like:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("reproduced_timed_run"))
print("cpu:", d.get("cpu_baseline",{}).get("value"), d.get("gpu_over_cpu"), d.get("parity_check"))
s=d["sparse"]; print("sparse:", s["value"], s["ms_per_step"], s["whole_step"], [(r["kernel"], r["frac"], r["us_per_tick"]) for r in s["roofline"]["kernels"]])
w=d["lambda_sweep"]; print("sweep:", w["value"], w["ms_per_step"], w["whole_step"])
PY
