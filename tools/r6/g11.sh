mkdir -p gpurun_out/r6k
timeout 900 python tools/ro_probe.py 256 4 4 > gpurun_out/r6k/ro_probe.json 2> gpurun_out/r6k/ro_probe.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6k/ro_probe.json"))
print(d["solves_per_s_after_first_iteration"]); print(d["reference_order"]["one_stream_profile_of_next_iteration"]); print(d.get("vs_oracle_twin"))
PY
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_pt.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6k/ro_probe_pt.json 2> gpurun_out/r6k/ro_probe_pt.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6k/ro_probe_pt.json"))
print([x["ticks"] for x in d["reference_order"]["per_iteration"]], [x["s"] for x in d["reference_order"]["per_iteration"]])
pt=d.get("phase_us_sum_over_workgroups")
print("stager: load-issue+dot-writes %.3g emit %.3g scan+sync %.3g norm terms+writes %.3g barrier wait %.3g" % (pt[6],pt[9],pt[10],pt[15],pt[8]))
print("fold us per wave 0..3:", pt[0:4], "folder0 barrier wait:", pt[7])
print("grid iterations:", pt[12]*100, "literal (no grid / budget):", pt[13]*100, "failed checks:", pt[14]*100)
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_order or order_faithful or tight_epsilon or scratch_problem or sequential_sums" > gpurun_out/r6k/pytest_ro.log 2>&1; tail -3 gpurun_out/r6k/pytest_ro.log
