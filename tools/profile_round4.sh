#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the measured evidence of round 4, outputs under gpurun_out/prof_$1/; tools/make_traffic_json.py
# copies the summaries into profiles/ and derives the HBM-traffic ratios bench.py quotes.
#   bench_driver.json / bench_driver_full.json   python bench.py --steps 20 --warmup 5   (what the driver runs; line + full record)
#   bench_kernel_trace.txt    rocprofv3 --kernel-trace --stats of THE SAME command, + busy time of k_xpass_dense (union of its dispatch
#                             intervals): must agree with all_launches.{timed_by_events, avg_us, busy_ms} of that run's full record
#   dense_pmc_{fetch,write}.txt, sparse_pmc_{fetch,write}.txt    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass), with the
#                             full record of the SAME invocation beside them (dense_pmc_*_full.json: its all_launches.alg_bytes is the
#                             denominator of the traffic ratio -- round 3 divided counters of one command by bytes of another)
#   sparse_kernel_trace.txt   kernel trace of the sparse leg
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DRV="python $R/bench.py --steps 20 --warmup 5"
DENSE="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep --no-config1 --no-gram"
SPARSE="python $R/bench.py --sparse-only --sparse-cpu-sample 0"
summ() { # dir, out, extra args
  DB=$(find $1 -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB ${@:3} > $2
  rm -rf $1
}
timeout 900 $DRV --full-json $OUT/bench_driver_full.json > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_b -o b -- $DRV --full-json $OUT/bench_traced_full.json > $OUT/bench_traced.json 2> $OUT/kt_bench.log
summ $OUT/kt_b $OUT/bench_kernel_trace.txt --busy k_xpass_dense --busy k_tron_step --busy k_rowpass_lds --busy k_rowcold --busy k_colpass_lds --busy k_step_a --busy k_step_b --busy k_step_c
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_s -o b -- $SPARSE --full-json $OUT/sparse_traced_full.json > /dev/null 2> $OUT/kt_sparse.log
summ $OUT/kt_s $OUT/sparse_kernel_trace.txt --busy k_rowpass_lds --busy k_rowcold --busy k_colpass_lds --busy k_step_a --busy k_step_b --busy k_step_c --busy k_step_commit
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/pd_$n -o b -- $DENSE --full-json $OUT/dense_pmc_${n}_full.json > /dev/null 2> $OUT/pmc_dense_$n.log;  summ $OUT/pd_$n $OUT/dense_pmc_$n.txt --pmc
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/ps_$n -o b -- $SPARSE --full-json $OUT/sparse_pmc_${n}_full.json > /dev/null 2> $OUT/pmc_sparse_$n.log; summ $OUT/ps_$n $OUT/sparse_pmc_$n.txt --pmc
done
rm -f $OUT/*.log
cat $OUT/bench_driver.json
head -6 $OUT/bench_kernel_trace.txt; grep -A 12 "^# busy" $OUT/bench_kernel_trace.txt
head -12 $OUT/sparse_kernel_trace.txt; grep -A 12 "^# busy" $OUT/sparse_kernel_trace.txt
grep -E "k_xpass_dense" $OUT/dense_pmc_fetch.txt $OUT/dense_pmc_write.txt | head
python - <<PY
import json
for n in ("bench_driver_full", "bench_traced_full"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["whole_step"], d["all_launches"])
PY
