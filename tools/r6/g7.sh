mkdir -p gpurun_out/r6g
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r6g/bench_line.json 2> gpurun_out/r6g/bench.err ) 2> gpurun_out/r6g/time.txt
tail -3 gpurun_out/r6g/time.txt; cat gpurun_out/r6g/bench_line.json | head -c 4200; echo; grep "leg:" gpurun_out/r6g/bench.err
cp bench_full.json gpurun_out/r6g/bench_full.json
python - <<'PY'
import json
f=json.load(open("gpurun_out/r6g/bench_full.json"))
print(json.dumps(f.get("reference_order"), indent=1)[:3000])
PY
