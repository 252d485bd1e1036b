#!/bin/bash
# Build A/B variants of the fused flow kernel into build_variants/ (git-ignored, shipped by gpurun).
#   [SRC=fm_optim.hip] tools/build_variants.sh name1:"-DFLAG ..." name2:"..."
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -shared $flags \
    flowmap_amd/csrc/${SRC:-fm_flow.hip} -o "build_variants/libfm_${name}.so" &
done
wait
ls -la build_variants
