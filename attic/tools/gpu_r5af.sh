#!/bin/bash
# round 5, GPU call AF: reference-order step with smaller chunks (66 / 33 KB of LDS instead of 132) so that pass workgroups of the other
# tick streams can share a CU with a step workgroup, with and without smaller pass footprints (MLX_SLW / MLX_RBMAX)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/abl
one() { # label, env...
  lab=$1; shift
  env RO_ONLY=1 "$@" timeout 300 python tools/ro_probe.py 256 3 1 > gpurun_out/r5af.json 2> gpurun_out/r5af.err || tail -2 gpurun_out/r5af.err
  python - "$lab" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5af.json"))
print(sys.argv[1], [x["solves_per_s"] for x in d["reference_order"]["per_iteration"]], d["vs_oracle_twin"]["bit_identical_beta_and_uplusx"], "/", d["vs_oracle_twin"]["solves"])
PY
}
one ch1024 A=1
one ch512 MLX_LIB_PATH=$A/libmlease_hip_ch512.so
one ch256 MLX_LIB_PATH=$A/libmlease_hip_ch256.so
one ch512_small_passes MLX_LIB_PATH=$A/libmlease_hip_ch512.so MLX_SLW=11264 MLX_RBMAX=11264
one ch256_mid_passes MLX_LIB_PATH=$A/libmlease_hip_ch256.so MLX_SLW=15360 MLX_RBMAX=15360
one ch1024_small_passes MLX_SLW=11264 MLX_RBMAX=11264
