#!/bin/bash
# A/B of PRE-BUILT whole-library variants (tools/build_lib_variants.sh) against the shipped library on the GPU box.
#   gpurun -- 'NAMES="pg1 pg1w2" TESTS="track or step" LINES="c2:--config+c2 r180:--height+180+--width+240+--tracking" bash tools/ab_lib_variants.sh'
# Per variant: a parity subset of the GPU suite (TESTS = pytest -k expression; empty = none), then every bench line twice; the shipped library's
# lines before and after.  KERNEL = substring of the kernel whose HIP-event time the line reports next to ms/step (default: the flow kernel).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${TAG:-ab}; out=gpurun_out/$tag; mkdir -p $out
line() {  # name, bench args ("+" for spaces)
  python bench.py ${2//+/ } --cpu-frames 0 --sustained-steps ${SUSTAINED:-0} --ate off --default-resolution off 2> $out/last.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
t = d.get('roofline_tracking', {}).get('kernel_ms')
s = (d.get('sustained') or {}).get('ms_per_step')
print('$1', 'ms_per_step', round(d['ms_per_step'], 4), 'flow_kernel_ms', round(d['roofline']['kernel_ms'], 4), 'track_kernel_ms', None if t is None else round(t, 4), 'sustained', None if s is None else round(s, 4))" | tee -a $out/lines.txt
}
run_lines() { for spec in $LINES; do line "$1_${spec%%:*}" "${spec#*:}"; done; }
cp flowmap_amd/libflowmap_hip.so /tmp/shipped.so
run_lines shipped
for name in $NAMES; do
  cp build_variants/libflowmap_hip_$name.so flowmap_amd/libflowmap_hip.so
  if [ -n "$TESTS" ]; then (FLOWMAP_SKIP_FULL_SIZE=1 timeout 600 python -m pytest tests -m gpu -q -x -k "$TESTS" 2>&1 | tail -3) | sed "s/^/$name: /" | tee -a $out/pytest.txt; fi
  run_lines $name; run_lines $name
done
cp /tmp/shipped.so flowmap_amd/libflowmap_hip.so
run_lines shipped
