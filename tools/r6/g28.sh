mkdir -p gpurun_out/r6ab
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_ptc.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6ab/ptc.json 2> gpurun_out/r6ab/ptc.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6ab/ptc.json"))
pt=d.get("phase_us_sum_over_workgroups")
print("ticks", [x["ticks"] for x in d["reference_order"]["per_iteration"]], "s", [x["s"] for x in d["reference_order"]["per_iteration"]])
n=pt[13]
print("units run:", n, " per unit us: staging %.1f relay %.1f offsets+dst+init %.1f first packs %.1f deep loop+stores %.1f" % (pt[8]/n, pt[12]/n, pt[9]/n, pt[10]/n, pt[11]/n))
print("row pass slots 0..6:", [round(x/1e3,1) for x in pt[0:7]])
PY
