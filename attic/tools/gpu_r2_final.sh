#!/bin/bash
# end-of-round check: GPU tests, smoke, then the round's profile set
OUT=gpurun_out/${1:-r2final}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print('default:', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('gram',{}).get('frac'), d['sparse']['value'])"
bash tools/profile_round2.sh ${1:-r2final}
