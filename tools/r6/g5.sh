mkdir -p gpurun_out/r6e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && RO_STREAMS=1 rocprofv3 --kernel-trace -d $R/gpurun_out/r6e/prof -o A -- python $R/tools/ro_dense_probe.py 64 7 0 > /dev/null 2>&1; cd $R
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/r6e/prof/*.db")[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
sc = next(c for c in ("start", "start_time", "begin") if c in cols)
rows = con.execute("select name, %s, duration from kernels where name like '%%k_ro_%%' order by %s" % (sc, sc)).fetchall()
# the last solve's launches: print sequence of (kind, us)
seq = [("R" if "rows" in n else "C" if "cols" in n else "S", d / 1e3) for n, s, d in rows]
print(len(seq))
out = []
for k, d in seq[-3 * 150:]:
    out.append("%s%.0f" % (k, d))
print(" ".join(out))
PY
