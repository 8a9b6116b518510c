#!/bin/bash
# Timing-experiment builds of the HIP library (never used by tests or the product): -DMLX_ABLATE=1 drops the LDS gathers of
# the sparse passes, =2 the index loads, =3 both; -DLSU=n changes the packs in flight; -DSTEP_MINW=n: minimum waves per SIMD of the step
# phases; -DSTEP_HEAD_A=64: head columns of phase A's grid (16 lanes per column in k_step_head). Output: tools/abl/libmlease_hip_<tag>.so
set -e
cd "$(dirname "$0")/../ml-ease_amd/csrc"
mkdir -p ../../tools/abl
build() { # tag, flags
  /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value $2 -c mlx_kernels.hip -o ../../tools/abl/k_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/libmlease_hip_$1.so ../../tools/abl/k_$1.o mlx_api.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  rm -f ../../tools/abl/k_$1.o
}
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  build "$tag" "$flags" &
done
wait
ls -la ../../tools/abl/
