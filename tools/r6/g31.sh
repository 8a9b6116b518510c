mkdir -p gpurun_out/r6ae
for t in base a4 a8 a16 a28; do
L=$PWD/tools/abl/libmlease_hip_$t.so; [ $t = base ] && L=$PWD/ml-ease_amd/csrc/libmlease_hip.so
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$L timeout 300 python tools/ro_probe.py 256 2 0 > gpurun_out/r6ae/$t.json 2> gpurun_out/r6ae/$t.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r6ae/$t.json"))
    pi=d["reference_order"]["per_iteration"]
    print("$t", [x["ticks"] for x in pi], [x["s"] for x in pi], "us per tick", [round(x["s"]*1e6/x["ticks"],1) for x in pi])
except Exception as e:
    print("$t failed", e); print(open("gpurun_out/r6ae/$t.err").read()[-400:])
PY
done
