#!/bin/bash
# round 3, last call: the no-flag bench line and the config #1 latency on the final code
OUT=gpurun_out/${1:-r3z}; mkdir -p $OUT
python tools/c1_latency.py 9 oracle > $OUT/c1_latency.txt 2>&1; cat $OUT/c1_latency.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["roofline"]["frac"])
s=d["sparse"]; print("sparse:", s["value"], s["whole_step"]["frac_of_hbm_peak"]); w=d["lambda_sweep"]; print("sweep:", w["value"])
PY
