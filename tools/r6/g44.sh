mkdir -p gpurun_out/r6aw
# the one-launch column pass at more than two row blocks, at scale: 48 partitions x 80 000 rows (4 row blocks), 3 ADMM iterations, 4 partitions against the oracle twin
timeout 1200 python tools/ro_probe.py 48 3 4 80000 > gpurun_out/r6aw/rows80k.json 2> gpurun_out/r6aw/rows80k.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6aw/rows80k.json"))
print("80k rows:", d["solves_per_s_after_first_iteration"], d["reference_order"]["one_stream_profile_of_next_iteration"]["us_per_tick"], d.get("vs_oracle_twin"))
PY
for i in 1 2 3; do
timeout 600 python tools/ro_probe.py 256 3 2 > gpurun_out/r6aw/soak$i.json 2> gpurun_out/r6aw/soak$i.err; python - <<PY
import json
d=json.load(open("gpurun_out/r6aw/soak$i.json"))
print("soak $i:", d["solves_per_s_after_first_iteration"], d.get("vs_oracle_twin"))
PY
done
