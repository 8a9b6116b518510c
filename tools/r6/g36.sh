mkdir -p gpurun_out/r6bg
for t in ptca1024 ptca2048; do
RO_ONLY=1 RO_STREAMS=1 MLX_LIB_PATH=$PWD/tools/abl/libmlease_hip_$t.so timeout 300 python tools/ro_probe.py 256 2 0 > gpurun_out/r6bg/$t.json 2> gpurun_out/r6bg/$t.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r6bg/$t.json"))
    pt=d.get("phase_us_sum_over_workgroups"); n=pt[13]
    print("$t units %d per unit us: staging %.1f relay %.1f offsets+dst+init %.1f first packs %.1f deep+stores %.1f  total %.1f" % (n, pt[8]/n, pt[12]/n, pt[9]/n, pt[10]/n, pt[11]/n, (pt[8]+pt[12]+pt[9]+pt[10]+pt[11])/n), [x["ticks"] for x in d["reference_order"]["per_iteration"]])
except Exception as e:
    print("$t failed", e); print(open("gpurun_out/r6bg/$t.err").read()[-300:])
PY
done
