mkdir -p gpurun_out/r6be
for ns in 4 3 2 1; do
RO_STREAMS=$ns timeout 600 python tools/ro_probe.py 256 4 0 > gpurun_out/r6be/s$ns.json 2> gpurun_out/r6be/s$ns.err; python - <<PY
import json
d=json.load(open("gpurun_out/r6be/s$ns.json"))
print("streams $ns", d["solves_per_s_after_first_iteration"], [round(x["solves_per_s"]) for x in d["reference_order"]["per_iteration"]])
PY
done
