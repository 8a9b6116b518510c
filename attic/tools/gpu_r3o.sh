#!/bin/bash
OUT=gpurun_out/${1:-r3o}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "two_tick or onehot_admm or library_rccl" > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
F="--no-profile --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep --no-gram"
for s in 2 3 4 2 4; do
  export MLX_STREAMS=$s
  timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 --no-profile > $OUT/c3_s$s.json 2> $OUT/c3_s$s.err
  python bench.py --steps 20 --warmup 5 $F > $OUT/d64_s$s.json 2> $OUT/d64_s$s.err
  python bench.py --steps 20 --warmup 5 --partitions 8 --rows 125000 $F > $OUT/d8_s$s.json 2> $OUT/d8_s$s.err
  python - <<PY
import json
d=json.loads(open("$OUT/c3_s$s.json").read().strip().splitlines()[-1]); print("streams=$s c3", d["solves_per_s"], d["ms_per_step"])
for t in ("d64","d8"):
    d=json.loads(open("$OUT/%s_s$s.json"%t).read().strip().splitlines()[-1]); print("streams=$s", t, d["value"], d["ms_per_step"], d["work"]["z32_sha1_after_timed_steps"][:8])
PY
done
