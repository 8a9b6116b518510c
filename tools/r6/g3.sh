mkdir -p gpurun_out/r6c
run() { name=$1; nv=$2; shift; shift; env "$@" timeout 600 python tools/ro_dense_probe.py 64 7 $nv > gpurun_out/r6c/$name.json 2> gpurun_out/r6c/$name.err; python - <<PY
import json
try:
    dd=json.load(open("gpurun_out/r6c/$name.json"))
    d=dd["reference_order"]
    p=d["one_stream_profile"]
    print("$name", "streams", d["tick_streams"], "solves/s per it", [x["solves_per_s"] for x in d["per_iteration"]], "us/tick", p.get("us_per_tick"), "frac row/col", p.get("row_frac_of_hbm_peak"), p.get("col_frac_of_hbm_peak"), dd.get("vs_oracle_twin"))
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/r6c/$name.err").read()[-800:])
PY
}
export MLX_RO_LD_ALIGN=32
run A2 0 MLX_ROD_CWG=1
run B2 2 MLX_ROD_CWG=2
run C2 2 MLX_ROD_CWG=3
run D2 0 MLX_ROD_CWG=2 MLX_ROD_NT=0
run E2 0 MLX_ROD_CWG=2 MLX_ROD_WGPC=0
run F2 0 MLX_ROD_CWG=2 RO_STREAMS=2
run G2 0 MLX_ROD_CWG=2 RO_STREAMS=1
