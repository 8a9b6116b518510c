mkdir -p gpurun_out/r6n
tools/seqfold_selftest | tail -1
timeout 900 python tools/ro_probe.py 256 4 4 > gpurun_out/r6n/ro_probe.json 2> gpurun_out/r6n/ro_probe.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6n/ro_probe.json"))
print(d["solves_per_s_after_first_iteration"]); print(d["reference_order"]["one_stream_profile_of_next_iteration"]); print(d.get("vs_oracle_twin"))
PY
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_pt.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6n/ro_probe_pt.json 2> gpurun_out/r6n/ro_probe_pt.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6n/ro_probe_pt.json"))
print([x["ticks"] for x in d["reference_order"]["per_iteration"]], [x["s"] for x in d["reference_order"]["per_iteration"]])
pt=[v*100 for v in d.get("phase_us_sum_over_workgroups")]
n=pt[13]/100*2
print("chunk-steps:", n)
print("per chunk-step cycles: stager load-issue+dot-writes %.0f emit %.0f scan+sync %.0f norm %.0f barrier-wait %.0f" % tuple(pt[i]/n for i in (8,9,10,11,12)))
print("per chunk-step cycles: fold wave0..3 %s, their barrier waits %s" % ([round(pt[i]/n) for i in range(4)], [round(pt[4+i]/n) for i in range(4)]))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_order or order_faithful or tight_epsilon or scratch_problem or sequential_sums" > gpurun_out/r6n/pytest_ro.log 2>&1; tail -3 gpurun_out/r6n/pytest_ro.log
