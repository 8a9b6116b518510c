mkdir -p gpurun_out/r6p
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_pt.so timeout 900 python tools/ro_probe.py 256 3 0 > gpurun_out/r6p/ro_probe_pt.json 2> gpurun_out/r6p/ro_probe_pt.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6p/ro_probe_pt.json"))
pt=[v*100 for v in d.get("phase_us_sum_over_workgroups")]
n=pt[13]/100*2
print("chunk-steps:", n)
print("per chunk-step cycles: stager load-issue+dot-writes %.0f emit %.0f scan+sync %.0f norm %.0f barrier-wait %.0f" % tuple(pt[i]/n for i in (8,9,10,11,12)))
print("per chunk-step cycles: fold wave0..3 %s, their barrier waits %s" % ([round(pt[i]/n) for i in range(4)], [round(pt[4+i]/n) for i in range(4)]))
print("wave 1: LDS-read part %.0f cycles per chunk-step; grid iterations %.2f, failed checks %.2f, literal(no grid/budget) %.2f per chunk-step" % (pt[14]/n, pt[15]/100/n, pt[7]/100/n, pt[6]/100/n))
PY
