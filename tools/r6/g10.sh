mkdir -p gpurun_out/r6j
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_pt.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6j/ro_probe_pt.json 2> gpurun_out/r6j/ro_probe_pt.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6j/ro_probe_pt.json"))
print([x["ticks"] for x in d["reference_order"]["per_iteration"]], [x["s"] for x in d["reference_order"]["per_iteration"]])
pt=d.get("phase_us_sum_over_workgroups")
tot=sum(pt[i] for i in (6,9,10,15,8))
print("stager: before-emit(load issue etc) %.3g emit(wait loads+elementwise+stores) %.3g scan+sync %.3g norm terms+writes %.3g dot writes(+rest) %.3g barrier wait %.3g" % (pt[6],pt[9],pt[10],pt[15],0,pt[8]))
print("fold us per wave 0..5:", pt[0:6], "folder0 barrier wait:", pt[7])
print("grid iterations:", pt[12]*100, "literal (no grid / budget):", pt[13]*100, "failed checks:", pt[14]*100)
PY
