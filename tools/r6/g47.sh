mkdir -p gpurun_out/r6ba
timeout 900 python tools/ro_probe.py 256 4 4 > gpurun_out/r6ba/ro_probe.json 2> gpurun_out/r6ba/ro_probe.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6ba/ro_probe.json"))
print(d["solves_per_s_after_first_iteration"]); print(d["reference_order"]["one_stream_profile_of_next_iteration"]["us_per_tick"]); print(d.get("vs_oracle_twin"))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_native_host.py -m gpu -q -x > gpurun_out/r6ba/pytest.log 2>&1; tail -3 gpurun_out/r6ba/pytest.log
