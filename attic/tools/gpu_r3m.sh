#!/bin/bash
OUT=gpurun_out/${1:-r3m}; mkdir -p $OUT
F="--no-profile --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep --no-gram"
for s in 1 2 1 2; do
  export MLX_STREAMS=$s
  python bench.py --steps 20 --warmup 5 $F > $OUT/d64_s$s.json 2> $OUT/d64_s$s.err
  python bench.py --steps 20 --warmup 5 --partitions 8 --rows 125000 $F > $OUT/d8_s$s.json 2> $OUT/d8_s$s.err
  python - <<PY
import json
for t in ("d64","d8"):
    d=json.loads(open("$OUT/%s_s$s.json"%t).read().strip().splitlines()[-1])
    print("streams=$s", t, d["value"], d["ms_per_step"], d["work"]["last_maxdiff"], d["work"]["z32_sha1_after_timed_steps"][:10])
PY
done
