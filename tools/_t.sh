cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python tools/bench_sparse.py --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['solves_per_s'], d['ms_per_step'], d['xpass_GBps_alg'], d['xpass_share'])"
timeout 300 python tools/bench_sparse.py --rows 1250000 --partitions 128 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C4gpu', d['solves_per_s'], d['ms_per_step'], d['xpass_GBps_alg'], d['xpass_share'])"
