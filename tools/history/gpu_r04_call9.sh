#!/bin/bash
# round 4, ninth GPU call: the ghost halo under hipGraph capture (its inputs now copied into persistent storage by one launch per step), the proxy rows
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04i; mkdir -p $out
export TMPDIR=/tmp
proxy=$out/strong_scaling_proxy_ghost.jsonl; : > $proxy
for k in 8 4 2; do
  for mode in "--graph compute --halo ghost" "--graph whole --halo ghost" "--graph compute --halo early" "--graph compute"; do
    timeout 300 python -X faulthandler bench.py --cpu-frames 0 --steps 200 --warmup 20 --sustained-steps 0 --share $k $mode >> $proxy 2>> $out/proxy.err || { echo "{\"failed\": \"--share $k $mode\"}" >> $proxy; grep -A12 "Fatal Python" $out/proxy.err | tail -14; }
  done
done
python - "$proxy" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
print("K  mode           frames  ms/step  flow-kernel ms")
for r in rows:
    if "failed" in r:
        print("FAILED", r["failed"]); continue
    k = r.get("proxy", {}).get("share_of", 1)
    mode = "whole" if "whole step replayed" in r["config"]["workload"] else "compute" if "collectives issued eagerly" in r["config"]["workload"] else "eager"
    h = str(r["config"].get("halo_exchange", ""))
    mode += "+early" if h.startswith("early") else "+ghost" if h.startswith("ghost") else ""
    print(f"{k:<2d} {mode:14s} {r['config']['frames_per_gpu']:>6d}  {r['ms_per_step']:.4f}   {r['roofline']['kernel_ms']:.4f}")
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_share8 -o s -- python3 $REPO/bench.py --steps 20 --warmup 5 --cpu-frames 0 --sustained-steps 0 --share 8 --graph compute --halo ghost) > $out/prof_share8.log 2>&1
python3 tools/export_profile.py $out/prof_share8 > $out/share8_ghost_rocprofv3_summary.csv 2>> $out/prof_share8.log; rm -rf $out/prof_share8; head -16 $out/share8_ghost_rocprofv3_summary.csv | cut -c1-150
