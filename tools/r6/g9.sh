mkdir -p gpurun_out/r6i
tools/seqfold_selftest | tail -1
timeout 900 python tools/ro_probe.py 256 4 4 > gpurun_out/r6i/ro_probe.json 2> gpurun_out/r6i/ro_probe.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6i/ro_probe.json"))
print(d["solves_per_s_after_first_iteration"]); print(d["reference_order"]["one_stream_profile_of_next_iteration"]); print(d.get("vs_oracle_twin"))
PY
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_pt.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6i/ro_probe_pt.json 2> gpurun_out/r6i/ro_probe_pt.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6i/ro_probe_pt.json"))
print([x["ticks"] for x in d["reference_order"]["per_iteration"]], [x["s"] for x in d["reference_order"]["per_iteration"]])
pt=d.get("phase_us_sum_over_workgroups")
print("fold us per wave 0..5:", pt[0:6], "stage:", pt[6], "folder0 barrier wait:", pt[7], "stager barrier wait:", pt[8], "other folders' wait:", pt[11])
print("grid iterations:", pt[12]*100, "literal (no grid / budget):", pt[13]*100, "failed checks:", pt[14]*100)
PY
