mkdir -p gpurun_out/r6b
run() { name=$1; shift; env "$@" timeout 600 python tools/ro_dense_probe.py 64 7 0 > gpurun_out/r6b/$name.json 2> gpurun_out/r6b/$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r6b/$name.json"))["reference_order"]
p=d["one_stream_profile"]
print("$name", "streams", d["tick_streams"], "solves/s per it", [x["solves_per_s"] for x in d["per_iteration"]], "us/tick", p.get("us_per_tick"), "frac row/col", p.get("row_frac_of_hbm_peak"), p.get("col_frac_of_hbm_peak"))
PY
}
run A X=1
run B MLX_ROD_NT=0
run C MLX_RO_LD_ALIGN=32
run D MLX_RO_LD_ALIGN=32 MLX_ROD_NT=0
run E MLX_ROD_CWG=4
run F RO_STREAMS=1
run G RO_STREAMS=2
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6b/prof -o A -- python $GRAFT_REPO_ROOT/tools/ro_dense_probe.py 64 4 0 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; ls gpurun_out/r6b/prof | head; python tools/rocpd_summary.py $(ls gpurun_out/r6b/prof/*.db | head -1) 2>/dev/null | head -30
