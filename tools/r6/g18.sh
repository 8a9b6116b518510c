mkdir -p gpurun_out/r6r
run() { name=$1; shift; env "$@" timeout 900 python tools/ro_probe.py 256 4 2 > gpurun_out/r6r/$name.json 2> gpurun_out/r6r/$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r6r/$name.json"))
print("$name", d["solves_per_s_after_first_iteration"], d["reference_order"]["one_stream_profile_of_next_iteration"]["us_per_tick"], d.get("vs_oracle_twin",{}).get("bit_identical_beta_and_uplusx"))
PY
grep "intercept-column" gpurun_out/r6r/$name.err | head -2
}
run side2 MLX_TRACE=0 X=1
run side1 MLX_RO_CSUM_SIDE=1 RO_STREAMS=1
run side3 MLX_RO_CSUM_SIDE=3
run side4 MLX_RO_CSUM_SIDE=4
run off MLX_RO_CSUM_SIDE=0
