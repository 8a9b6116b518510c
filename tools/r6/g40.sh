mkdir -p gpurun_out/r6as
tools/seqfold_selftest | tail -1
timeout 900 python tools/ro_probe.py 256 4 4 > gpurun_out/r6as/ro_probe.json 2> gpurun_out/r6as/ro_probe.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6as/ro_probe.json"))
print(d["solves_per_s_after_first_iteration"]); print(d["reference_order"]["one_stream_profile_of_next_iteration"]["us_per_tick"]); print(d.get("vs_oracle_twin"))
PY
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_rop.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6as/rop.json 2> gpurun_out/r6as/rop.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6as/rop.json"))
print(d.get("ro_step_pass_us_per_tick"))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_order or order_faithful or tight_epsilon or scratch_problem or sequential_sums or norm_tests" > gpurun_out/r6as/pytest_ro.log 2>&1; tail -3 gpurun_out/r6as/pytest_ro.log
