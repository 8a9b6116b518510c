#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the measured evidence of a round, outputs under gpurun_out/prof_$1/ (copy the summaries
# into profiles/ afterwards: tools/make_traffic_json.py does that and derives the HBM-traffic ratios).
#   bench_driver.json          python bench.py --steps 20 --warmup 5            (what the driver runs)
#   dense_kernel_trace.txt     rocprofv3 --kernel-trace --stats of the dense leg
#   sparse_kernel_trace.txt    ... of the sparse leg (bench.py --sparse-only)
#   {dense,sparse}_pmc_{fetch,write}.txt   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass, --kernel-trace only)
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DENSE="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep"
SPARSE="python $R/bench.py --sparse-only --sparse-cpu-sample 0"
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 600 $DENSE > $OUT/bench_dense_short.json 2> /dev/null
timeout 600 $SPARSE > $OUT/bench_sparse_only.json 2> /dev/null
summ() { # dir, out, [--pmc]
  DB=$(find $1 -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB ${3:-} > $2
  rm -rf $1
}
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_d -o b -- $DENSE > $OUT/kt_dense.log 2>&1;   summ $OUT/kt_d $OUT/dense_kernel_trace.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_s -o b -- $SPARSE > $OUT/kt_sparse.log 2>&1; summ $OUT/kt_s $OUT/sparse_kernel_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/pd_$n -o b -- $DENSE --no-profile > $OUT/pmc_dense_$n.log 2>&1;  summ $OUT/pd_$n $OUT/dense_pmc_$n.txt --pmc
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/ps_$n -o b -- $SPARSE > $OUT/pmc_sparse_$n.log 2>&1;             summ $OUT/ps_$n $OUT/sparse_pmc_$n.txt --pmc
done
rm -f $OUT/*.log
head -8 $OUT/dense_kernel_trace.txt; head -10 $OUT/sparse_kernel_trace.txt
python - <<PY
import json
d=json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print("driver-like:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["time_to_ref_loglik"]["reached_at_iteration"], d["time_to_ref_loglik"]["seconds_all_iterations"])
print("cpu:", d.get("cpu_baseline",{}).get("value"), d.get("gpu_over_cpu"), d.get("parity_check"))
s=d["sparse"]; print("sparse:", s["value"], s["ms_per_step"], s["whole_step"], [(r["kernel"], r["frac"], r["us_per_tick"]) for r in s["roofline"]["kernels"]])
w=d["lambda_sweep"]; print("sweep:", w["value"], w["ms_per_step"], w["whole_step"], [(r["kernel"][:12], r["frac"], r["us_per_tick"]) for r in w["roofline"]["kernels"]])
v=d["time_to_ref_loglik"].get("vs_oracle_run", {}); print("vs oracle run:", {k: v[k] for k in v if k not in ("per_iteration", "source", "reading")})
PY
