#!/bin/bash
# round 3, final bench line with the driver's flags (final code, incl. the config1_latency leg)
OUT=gpurun_out/${1:-r3x}; mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print("driver-like:", d["value"], d["ms_per_step"], d["roofline"]["frac"])
s=d["sparse"]; print("sparse:", s["value"], s["whole_step"]["frac_of_hbm_peak"]); print("sweep:", d["lambda_sweep"]["value"])
print("config1:", d.get("config1_latency"))
PY
