#!/bin/bash
OUT=gpurun_out/${1:-r3n}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -q -x -k "multirank or two_ranks or two_handles or dense or tight or split_api or reproduc" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-sparse --no-sweep > $OUT/bench.json 2> $OUT/bench.err; echo rc=$?
tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"], d["roofline"]["measured_in"])
print(d["cpu_baseline"]["value"], d["gpu_over_cpu"], d["parity_check"]["per_iteration"][-1]["partitions_with_equal_counters"], d["time_to_ref_loglik"]["seconds_all_iterations"], d["time_to_ref_loglik"]["reached_at_iteration"])
PY
