mkdir -p gpurun_out/r6at
for v in 1 0; do
MLX_RO_COL_MERGED=$v timeout 900 python tools/ro_probe.py 256 4 4 > gpurun_out/r6at/m$v.json 2> gpurun_out/r6at/m$v.err; python - <<PY
import json
d=json.load(open("gpurun_out/r6at/m$v.json"))
print("merged=$v", d["solves_per_s_after_first_iteration"], d["reference_order"]["one_stream_profile_of_next_iteration"]["us_per_tick"], d.get("vs_oracle_twin"))
PY
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r6at/pytest.log 2>&1; tail -3 gpurun_out/r6at/pytest.log
