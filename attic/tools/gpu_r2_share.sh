#!/bin/bash
# bench.py's N>1 control flow on ONE GPU (MLX_BENCH_SHARE_GPU=1: all ranks on device 0, collectives over gloo): 2 and 4 ranks,
# strong and weak, against the plain N=1 run of the same flags.
OUT=gpurun_out/${1:-r2s}; mkdir -p $OUT
F="--steps 3 --warmup 1 --no-cpu-baseline --loglik-iters 3 --sparse-steps 2 --sparse-warmup 1 --sparse-cpu-sample 0"
timeout 600 python bench.py $F > $OUT/n1.json 2> $OUT/n1.err
export MLX_BENCH_SHARE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 $F > $OUT/n2.json 2> $OUT/n2.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 4 $F --sparse-partitions 256 > $OUT/n4.json 2> $OUT/n4.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29523 bench.py --gpus 2 --scaling weak --rows 250000 --partitions 16 $F --no-sparse > $OUT/n2weak.json 2> $OUT/n2weak.err
python - <<PY
import json
for t in ("n1","n2","n4","n2weak"):
    try:
        d=json.loads(open("$OUT/%s.json"%t).read().strip().splitlines()[-1])
        sp=d.get("sparse") or {}
        print(t, d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], d["config"]["partitions"], d["config"]["partitions_per_gpu"], d["work"]["solves"], d["work"]["last_maxdiff"],
              (d.get("time_to_ref_loglik") or {}).get("loglik_by_iteration"), "| sparse", sp.get("value"), sp.get("workload","")[:28], sp.get("last_maxdiff"))
    except Exception as e:
        print(t, "ERR", e); print(open("$OUT/%s.err"%t).read()[-1500:])
PY
