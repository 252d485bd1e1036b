#!/bin/bash
# round 4: does the 50 ms host pause between warm-up and the timed region (the GC freeze) cause the transient of the driver's 20-step window?
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04o; mkdir -p $out
export TMPDIR=/tmp
for i in 1 2; do for gc in first early late; do
  FLOWMAP_BENCH_GC=$gc timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > $out/bench_driver_${gc}_$i.json 2> $out/bench_driver_${gc}_$i.err
  python3 - $out/bench_driver_${gc}_$i.json $gc$i <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); k = r["roofline"]["kernel_ms_per_launch"]
print(sys.argv[2], "ms/step", round(r["ms_per_step"], 4), "kernel avg", round(r["roofline"]["kernel_ms"], 4), "frac", round(r["roofline"]["frac"], 3), "sustained", round(r["sustained"]["ms_per_step"], 4), "per launch", k)
PY
done; done
