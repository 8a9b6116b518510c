#!/bin/bash
# round 5, GPU call C: reference-order numerics -- parity tests, probe, and a kernel trace of the probe (which launch is slow?)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "order_faithful or tight_epsilon" > gpurun_out/r5c_ro_tests.log 2>&1
echo "ro tests rc=$?"; tail -3 gpurun_out/r5c_ro_tests.log
timeout 900 python tools/ro_probe.py 256 4 8 > gpurun_out/r5c_ro_probe.json 2> gpurun_out/r5c_ro_probe.err
echo "probe rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c_ro_probe.json"))
print(json.dumps({k: d[k] for k in ("solves_per_s_after_first_iteration", "vs_oracle_twin")}))
for k in ("fast", "reference_order"):
    print(k, [x["solves_per_s"] for x in d[k]["per_iteration"]], d[k]["one_stream_profile_of_next_iteration"])
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_ro -o b -- python $R/tools/ro_probe.py 128 2 0 > /dev/null 2> $R/gpurun_out/r5c_kt.log
DB=$(find $R/gpurun_out/kt_ro -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $DB --busy k_colpass_lds --busy k_ro_step --busy k_rowpass_lds > $R/gpurun_out/r5c_ro_kernel_trace.txt
python - <<PY
import sqlite3
con = sqlite3.connect("$DB")
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
print(cols)
for nm in ("%k_colpass_ldsILb0ELb0ELb1%", "%k_ro_step%", "%k_colpass_lds<false, false, true>%", "%k_colpass_lds%true>%"):
    rows = cur.execute("select duration from kernels where name like ? order by start", (nm,)).fetchall()
    d = [r[0] / 1e3 for r in rows]
    if d:
        print(nm, len(d), "first 24 durations (us):", [round(x) for x in d[:24]])
PY
rm -rf $R/gpurun_out/kt_ro
head -14 $R/gpurun_out/r5c_ro_kernel_trace.txt
