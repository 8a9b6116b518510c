mkdir -p gpurun_out/r6f
run() { name=$1; nv=$2; shift; shift; env "$@" timeout 600 python tools/ro_dense_probe.py 64 7 $nv > gpurun_out/r6f/$name.json 2> gpurun_out/r6f/$name.err; python - <<PY
import json
try:
    dd=json.load(open("gpurun_out/r6f/$name.json"))
    d=dd["reference_order"]
    p=d["one_stream_profile"]
    print("$name", "streams", d["tick_streams"], "solves/s per it", [x["solves_per_s"] for x in d["per_iteration"]], "us/tick", p.get("us_per_tick"), "frac row/col", p.get("row_frac_of_hbm_peak"), p.get("col_frac_of_hbm_peak"), dd.get("vs_oracle_twin"))
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/r6f/$name.err").read()[-800:])
PY
}
run A4 2 MLX_ROD_CWG=4
run A3 2 MLX_ROD_CWG=3
run A22 0 MLX_ROD_CWG=22
run A2 0 MLX_ROD_CWG=2
run A1 0 MLX_ROD_CWG=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_order_dense_tiles or tight_epsilon" > gpurun_out/r6f/pytest_ro.log 2>&1; tail -3 gpurun_out/r6f/pytest_ro.log
