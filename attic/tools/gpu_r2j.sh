#!/bin/bash
# Round-2 GPU check J: full parity tests (RCCL status slot), bench.py default + the per-GPU shape of the 8-GPU strong scaling
# (8 partitions of 15 625 x 1000 on one GPU), torchrun N=1 path.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2j}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error" $OUT/tests.txt | tail -8
cd /tmp
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print("driver-like:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["time_to_ref_loglik"]["reached_at_iteration"], d["time_to_ref_loglik"]["seconds_all_iterations"], d["time_to_ref_loglik"].get("abs_diff_to_oracle_by_iteration_max"))
print("cpu:", d.get("cpu_baseline",{}).get("value"), d.get("gpu_over_cpu"), d.get("parity_check"))
s=d["sparse"]; print("sparse:", s["value"], s["ms_per_step"], s["whole_step"], [(r["kernel"], r["frac"], r["us_per_tick"]) for r in s["roofline"]], s.get("cpu_baseline"))
PY
for rpb in default 512; do
  if [ $rpb = default ]; then E="X=1"; else E="MLX_DENSE_RPB=$rpb"; fi
  env $E timeout 300 python $R/bench.py --steps 20 --warmup 5 --partitions 8 --rows 125000 --no-sparse --no-cpu-baseline --loglik-iters 0 > $OUT/bench_8part_$rpb.json 2> $OUT/bench_8part_$rpb.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_8part_$rpb.json").read().strip().splitlines()[-1])
print("8 partitions (rpb $rpb):", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["xpass_share_of_step"])
PY
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --loglik-iters 0 > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_torchrun1.json").read().strip().splitlines()[-1])
print("torchrun N=1:", d["value"], d["ms_per_step"], d["sparse"]["value"])
PY
tail -3 $OUT/bench_torchrun1.err | cut -c1-300
