mkdir -p gpurun_out/r6am
timeout 900 python tools/ro_probe.py 256 4 4 > gpurun_out/r6am/ro_probe.json 2> gpurun_out/r6am/ro_probe.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6am/ro_probe.json"))
print(d["solves_per_s_after_first_iteration"]); print(d["reference_order"]["one_stream_profile_of_next_iteration"]); print(d.get("vs_oracle_twin"))
PY
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_ptc.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6am/ptc.json 2> gpurun_out/r6am/ptc.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6am/ptc.json"))
pt=d.get("phase_us_sum_over_workgroups")
n=pt[13]
print("units run:", n, " per unit us: staging %.1f relay %.1f offsets+dst+init %.1f first packs %.1f deep loop+stores %.1f" % (pt[8]/n, pt[12]/n, pt[9]/n, pt[10]/n, pt[11]/n))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x  > gpurun_out/r6am/pytest_ro.log 2>&1; tail -3 gpurun_out/r6am/pytest_ro.log
