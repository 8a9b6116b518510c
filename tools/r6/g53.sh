mkdir -p gpurun_out/r6bi
for cu in 131072 196608 262144 393216; do
MLX_CUNIT=$cu RO_ONLY=1 timeout 600 python tools/ro_probe.py 256 4 0 > gpurun_out/r6bi/c$cu.json 2> gpurun_out/r6bi/c$cu.err; python - <<PY
import json
d=json.load(open("gpurun_out/r6bi/c$cu.json"))
print("MLX_CUNIT=$cu", [round(x["solves_per_s"]) for x in d["reference_order"]["per_iteration"]])
PY
done
