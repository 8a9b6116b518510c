#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the measured evidence of round 6 (round 5's script plus the dense reference-order kernels' trace:
# ro_dense_kernel_trace.txt / ro_dense_probe.json from tools/ro_dense_probe.py, configs[1]'s 64 tiles), outputs under gpurun_out/prof_$1/; tools/make_traffic_json.py
# copies the summaries into profiles/ and derives the HBM-traffic ratios bench.py quotes. As tools/profile_round4.sh, plus:
#   ro_kernel_trace.txt           rocprofv3 --kernel-trace of tools/ro_probe.py: the reference-order tick kernels (k_rowpass_lds<.., RO>,
#                                 k_colpass_lds<.., RO> per row block, k_ro_step) beside the product path's on the same data
#   ro_probe.json                 that probe run plain: solves/s of both numerics, bit-identity against the oracle twin
#   multirank_shared_gpu.txt      bench.py --gpus 8 under torch.distributed.run with all 8 ranks on THIS device (MLX_BENCH_SHARE_GPU=1, gloo):
#                                 the N = 8 control flow of all three legs (configs[1] strong-scaled, configs[3]'s 1024 partitions, the
#                                 8192-problem sweep) -- a check, not a measurement
set -u
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DRV="python $R/bench.py --steps 20 --warmup 5"
DENSE="python $R/bench.py --steps 5 --warmup 2 --no-dense-ro --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep --no-config1 --no-gram"
SPARSE="python $R/bench.py --sparse-only --no-sparse128 --sparse-cpu-sample 0 --sparse-loglik-iters 0 --no-ingest"
summ() { # dir, out, extra args
  DB=$(find $1 -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB ${@:3} > $2
  rm -rf $1
}
SECONDS=0
timeout 900 $DRV --full-json $OUT/bench_driver_full.json > $OUT/bench_driver.json 2> $OUT/bench_driver.err
echo "driver-flag bench: $SECONDS s"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_b -o b -- $DRV --full-json $OUT/bench_traced_full.json > $OUT/bench_traced.json 2> $OUT/kt_bench.log
summ $OUT/kt_b $OUT/bench_kernel_trace.txt --busy k_xpass_dense --busy k_tron_step --busy k_rowpass_lds --busy k_rowcold --busy k_colpass_lds --busy k_step_a --busy k_step_b --busy k_step_c --busy k_ro_step --busy k_ro_dense_rows --busy k_ro_dense_cols
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_s -o b -- $SPARSE --full-json $OUT/sparse_traced_full.json > /dev/null 2> $OUT/kt_sparse.log
summ $OUT/kt_s $OUT/sparse_kernel_trace.txt --busy k_rowpass_lds --busy k_rowcold --busy k_colpass_lds --busy k_step_a --busy k_step_b --busy k_step_c --busy k_step_commit
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/pd_$n -o b -- $DENSE --full-json $OUT/dense_pmc_${n}_full.json > /dev/null 2> $OUT/pmc_dense_$n.log;  summ $OUT/pd_$n $OUT/dense_pmc_$n.txt --pmc
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/ps_$n -o b -- $SPARSE --full-json $OUT/sparse_pmc_${n}_full.json > /dev/null 2> $OUT/pmc_sparse_$n.log; summ $OUT/ps_$n $OUT/sparse_pmc_$n.txt --pmc
done
timeout 600 python $R/tools/ro_probe.py 256 4 8 > $OUT/ro_probe.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_ro -o b -- python $R/tools/ro_probe.py 256 3 0 > /dev/null 2> $OUT/kt_ro.log
summ $OUT/kt_ro $OUT/ro_kernel_trace.txt --busy k_rowpass_lds --busy k_colpass_lds --busy k_ro_step
timeout 600 python $R/tools/ro_dense_probe.py 64 4 4 > $OUT/ro_dense_probe.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_rod -o b -- python $R/tools/ro_dense_probe.py 64 3 0 > /dev/null 2> $OUT/kt_rod.log
summ $OUT/kt_rod $OUT/ro_dense_kernel_trace.txt --busy k_ro_dense_rows --busy k_ro_dense_cols --busy k_ro_step
echo "profiles: $SECONDS s"
cd $R
( echo "# MLX_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 --steps 3 --warmup 1 --loglik-iters 3 --sparse-steps 2 --sparse-warmup 1 --sweep-steps 1 --sweep-warmup 1"
  echo "# all 8 ranks on ONE MI355X, collectives over gloo through host staging: the N = 8 control flow of every leg, not a measurement"
  MLX_BENCH_SHARE_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 \
      --steps 3 --warmup 1 --loglik-iters 3 --sparse-steps 2 --sparse-warmup 1 --sweep-steps 1 --sweep-warmup 1 --full-json $OUT/multirank8_full.json 2> $OUT/multirank8.err
  echo "# rc=$?"
  python - <<PY
import json
d = json.load(open("$OUT/multirank8_full.json"))
print(json.dumps({"n_gpus": d["n_gpus"], "config": d["config"], "work": d["work"], "scaling": d["scaling"], "test_mode": d.get("test_mode"),
                  "sparse": {k: d["sparse"][k] for k in ("workload", "value", "n_gpus", "partitions", "last_maxdiff", "ticks_per_step")},
                  "lambda_sweep": {k: d["lambda_sweep"][k] for k in ("workload", "value", "n_gpus", "problems_per_gpu", "last_maxdiff")}}, indent=1))
PY
) > $OUT/multirank_shared_gpu.txt 2>&1
echo "multirank: $SECONDS s"
rm -f $OUT/*.log $OUT/multirank8.err
cat $OUT/bench_driver.json
head -8 $OUT/bench_kernel_trace.txt; grep -A 12 "^# busy" $OUT/bench_kernel_trace.txt
head -14 $OUT/sparse_kernel_trace.txt; grep -A 12 "^# busy" $OUT/sparse_kernel_trace.txt
head -12 $OUT/ro_kernel_trace.txt
head -10 $OUT/ro_dense_kernel_trace.txt
tail -30 $OUT/multirank_shared_gpu.txt
python - <<PY
import json
for n in ("bench_driver_full", "bench_traced_full"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_busy_union"], d["roofline"]["kernel_ms_per_step"], d["whole_step"], d["all_launches"])
PY
