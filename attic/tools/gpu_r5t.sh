#!/bin/bash
# round 5, GPU call T: reference-order numerics on 8 / 6 tick streams (MAX_TS raised to 8) against the default 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for ns in 8 6 4; do
  echo "RO_STREAMS=$ns"
  RO_STREAMS=$ns timeout 600 python tools/ro_probe.py 256 4 4 > gpurun_out/r5t_$ns.json 2> gpurun_out/r5t_$ns.err
  grep ro_probe gpurun_out/r5t_$ns.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r5t_$ns.json"))
print(json.dumps({k: d[k] for k in ("solves_per_s_after_first_iteration", "vs_oracle_twin")}))
PY
done
