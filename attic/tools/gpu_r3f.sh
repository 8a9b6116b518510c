#!/bin/bash
OUT=gpurun_out/${1:-r3f}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "posterior or limits" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
timeout 300 python tools/bench_gram.py --reps 5 | tee $OUT/bench_gram.json | cut -c1-400
timeout 300 python tools/bench_gram.py --reps 3 --rows 39062 --features 2000 | cut -c1-400
