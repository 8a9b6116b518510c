mkdir -p gpurun_out/r6q
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_st.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6q/ro_probe_st.json 2> gpurun_out/r6q/ro_probe_st.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6q/ro_probe_st.json"))
pt=[v*100 for v in d.get("phase_us_sum_over_workgroups")]
it, failed, lit = pt[15], pt[14], pt[13]
print("grid iterations %.3g failed checks %.3g literal(no grid / budget) %.3g -> fold calls ~ %.3g; failed per call %.2f, no-grid/budget literals per call %.2f" % (it, failed, lit, it-failed, failed/max(1,it-failed), lit/max(1,it-failed)))
print([x["ticks"] for x in d["reference_order"]["per_iteration"]])
PY
