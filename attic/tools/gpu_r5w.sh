#!/bin/bash
# round 5, GPU call W: reference-order column pass A/B: work-unit size (MLX_CUNIT) and the relay threshold (RO_LONG_T builds)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
one() { # label, env...
  lab=$1; shift
  env RO_ONLY=1 "$@" timeout 300 python tools/ro_probe.py 256 3 1 > gpurun_out/r5w.json 2> gpurun_out/r5w.err
  python - "$lab" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5w.json"))
print(sys.argv[1], [x["solves_per_s"] for x in d["reference_order"]["per_iteration"]], d["vs_oracle_twin"]["bit_identical_beta_and_uplusx"], "/", d["vs_oracle_twin"]["solves"])
PY
}
one default A=1
one cunit131072 MLX_CUNIT=131072
one cunit524288 MLX_CUNIT=524288
one cunit65536 MLX_CUNIT=65536
one longt16 MLX_LIB_PATH=$GRAFT_REPO_ROOT/tools/abl/libmlease_hip_lt16.so
one longt32 MLX_LIB_PATH=$GRAFT_REPO_ROOT/tools/abl/libmlease_hip_lt32.so
