#!/bin/bash
# Does a smaller lock-step group (vectors + index streams resident in the 256 MB Infinity Cache) tick faster per problem?
mkdir -p gpurun_out/r2g
for p in 16 32 64 128 256; do
  rows=$((39063 * p))
  timeout 300 python tools/bench_sparse.py --rows $rows --partitions $p --steps 3 --warmup 1 > gpurun_out/r2g/p$p.json 2> gpurun_out/r2g/p$p.err
done
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2g/pytest.log 2>&1
tail -3 gpurun_out/r2g/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r2g/smoke.log 2>&1
tail -2 gpurun_out/r2g/smoke.log
for p in 16 32 64 128 256; do tail -c 600 gpurun_out/r2g/p$p.json; echo; done
