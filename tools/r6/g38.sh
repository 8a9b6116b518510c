mkdir -p gpurun_out/r6ao
for v in packs; do
if [ $v = entries ]; then export MLX_RO_SORT_ENTRIES=1; fi
timeout 900 python tools/ro_probe.py 256 4 4 > gpurun_out/r6ao/$v.json 2> gpurun_out/r6ao/$v.err; python - <<PY
import json
d=json.load(open("gpurun_out/r6ao/$v.json"))
print("$v", d["solves_per_s_after_first_iteration"], d["reference_order"]["one_stream_profile_of_next_iteration"]["us_per_tick"], d.get("vs_oracle_twin"))
PY
done
