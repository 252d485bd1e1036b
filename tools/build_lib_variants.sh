#!/bin/bash
# Whole-library build variants for A/B runs on the GPU box (cross-compiled here: no hipcc time there).
#   tools/build_lib_variants.sh name1:"fm_track.hip:-DFM_TRACK_PG=1" name2:"fm_procrustes.hip:-DFOO=1 -DBAR=2" ...
# -> build_variants/libflowmap_hip_<name>.so = the shipped objects with ONE source rebuilt with the extra flags.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants/obj
python -m flowmap_amd.build > /dev/null   # the shipped objects (flowmap_amd/csrc/build/*.o)
for spec in "$@"; do
  name="${spec%%:*}"; rest="${spec#*:}"; src="${rest%%:*}"; flags="${rest#*:}"
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function $flags -c flowmap_amd/csrc/$src -o build_variants/obj/${name}_${src%.hip}.o
    objs=""
    for o in flowmap_amd/csrc/build/*.o; do
      if [ "$(basename $o)" = "${src%.hip}.o" ]; then objs="$objs build_variants/obj/${name}_${src%.hip}.o"; else objs="$objs $o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o build_variants/libflowmap_hip_${name}.so
    echo "built build_variants/libflowmap_hip_${name}.so ($src $flags)"
  ) &
done
wait
