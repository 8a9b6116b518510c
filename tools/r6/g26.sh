mkdir -p gpurun_out/r6z
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_rop.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6z/rop1.json 2> gpurun_out/r6z/rop1.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6z/rop1.json"))
print(d.get("ro_step_pass_us_per_tick")); print(d["reference_order"]["one_stream_profile_of_next_iteration"])
PY
