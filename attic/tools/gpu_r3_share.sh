#!/bin/bash
# bench.py's N>1 control flow at FULL size on ONE GPU (MLX_BENCH_SHARE_GPU=1: all ranks on device 0, collectives over gloo): 2 and 4
# ranks against the plain N=1 run of the same flags -- a correctness check of sharding / exchange / replay / reductions, NOT a scaling
# measurement (the ranks share one GPU).
OUT=gpurun_out/${1:-r3s}; mkdir -p $OUT
F="--steps 3 --warmup 1 --no-cpu-baseline --loglik-iters 3 --sparse-steps 2 --sparse-warmup 1 --sparse-cpu-sample 0 --sweep-steps 1 --sweep-warmup 1 --sweep-cpu-sample 0 --no-gram"
timeout 600 python bench.py $F > $OUT/n1.json 2> $OUT/n1.err
export MLX_BENCH_SHARE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 $F > $OUT/n2.json 2> $OUT/n2.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 4 $F --sparse-partitions 256 > $OUT/n4.json 2> $OUT/n4.err
python - <<PY > $OUT/summary.txt
import json
print("# run, n_gpus, dense solves/s, ms_per_step, partitions_per_gpu, last max|z - z_prev|, z32 sha1 (10), roofline reproduced, loglik by iteration | sparse solves/s, workload, last maxdiff | sweep solves/s")
for t in ("n1","n2","n4"):
    try:
        d=json.loads(open("$OUT/%s.json"%t).read().strip().splitlines()[-1])
        sp=d.get("sparse") or {}; sw=d.get("lambda_sweep") or {}
        print(t, d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["partitions_per_gpu"], d["work"]["last_maxdiff"], d["work"]["z32_sha1_after_timed_steps"][:10],
              d["roofline"]["reproduced_timed_run"], (d.get("time_to_ref_loglik") or {}).get("loglik_by_iteration"), "| sparse", sp.get("value"), sp.get("workload","")[:28], sp.get("last_maxdiff"),
              "| sweep", sw.get("value"), sw.get("problems_per_gpu"))
    except Exception as e:
        print(t, "ERR", e); print(open("$OUT/%s.err"%t).read()[-1500:])
PY
cat $OUT/summary.txt
