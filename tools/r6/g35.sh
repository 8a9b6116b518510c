mkdir -p gpurun_out/r6ai
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_st.so timeout 600 python tools/ro_probe.py 256 3 0 > gpurun_out/r6ai/st.json 2> gpurun_out/r6ai/st.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6ai/st.json"))
pt=[v*100 for v in d["phase_us_sum_over_workgroups"]]
pi=d["reference_order"]["per_iteration"]
pt_n = sum(x["ticks"] for x in pi) * 256
names=["CG A d.Hd","CG B r.r","CG other","EVAL rows","EVAL n"]
for p in range(5):
    lit, failed, it = pt[3*p], pt[3*p+1], pt[3*p+2]
    print("%-10s grid iterations %.3g failed checks %.3g literal(no grid/budget) %.3g  fold calls %.3g  failed/call %.2f literal/call %.2f" % (names[p], it, failed, lit, it-failed, failed/max(1,it-failed), lit/max(1,it-failed)))
PY
