mkdir -p gpurun_out/r6ae
for t in a4 a8 a16 a28; do
RO_ONLY=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_$t.so timeout 300 python tools/ro_probe.py 256 2 0 > gpurun_out/r6ae/$t.json 2> gpurun_out/r6ae/$t.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r6ae/$t.json"))
    print("$t", d["reference_order"]["one_stream_profile_of_next_iteration"], [x["ticks"] for x in d["reference_order"]["per_iteration"]])
except Exception as e:
    print("$t failed", e); print(open("gpurun_out/r6ae/$t.err").read()[-400:])
PY
done
