mkdir -p gpurun_out/r6ak
for t in ropa32 ropa64; do
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_$t.so timeout 600 python tools/ro_probe.py 256 3 0 > gpurun_out/r6ak/$t.json 2> gpurun_out/r6ak/$t.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r6ak/$t.json"))
    pi=d["reference_order"]["per_iteration"]
    print("$t", [x["ticks"] for x in pi], [x["s"] for x in pi], d.get("ro_step_pass_us_per_tick"))
except Exception as e:
    print("$t failed", e); print(open("gpurun_out/r6ak/$t.err").read()[-400:])
PY
done
