#!/bin/bash
# round 5, GPU call V: reference-order column pass, launch of row block 0 against row block 1 (rocprofv3 kernel trace, one tick stream)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
RO_ONLY=1 RO_STREAMS=1 timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_r5v -o b -- python $R/tools/ro_probe.py 256 2 1 > $R/gpurun_out/r5v.json 2> $R/gpurun_out/r5v.err
db=$(find $R/gpurun_out/kt_r5v -name "*.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select start, duration from kernels where name like '%k_colpass_lds%true>%' order by start").fetchall()
d = [r[1] / 1e3 for r in rows]
print("launches", len(d))
ev, od = d[0::2], d[1::2]
import statistics as st
big = [(a, b) for a, b in zip(ev, od) if a > 50 and b > 50]
print("pairs with both > 50 us:", len(big), " mean block0 %.1f us, mean block1 %.1f us" % (st.mean(a for a, b in big), st.mean(b for a, b in big)))
print("first 16 pairs:", [(round(a), round(b)) for a, b in big[:16]])
PY
rm -rf $R/gpurun_out/kt_r5v
