#!/bin/bash
# Per-rank step time of a K-GPU strong-scaling run of the headline workload (150 frames @ 720p, flow loss) measured on ONE GPU:
# `bench.py --share K` runs an interior rank's share (its pairs + halo frames, every collective on a one-rank RCCL
# communicator, the halo exchange replaced by its local copies/adds), eager and replayed as one hipGraph.
#   bash tools/scaling_proxy.sh [out.jsonl] [extra bench args...]      (through gpurun; K = 1 is the unsharded step)
cd "${GRAFT_REPO_ROOT:-.}"
out=${1:-gpurun_out/strong_scaling_proxy.jsonl}; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for k in 1 2 4 8; do
  for mode in "--graph off" "--graph compute" "--graph whole" "--graph off --halo early" "--graph compute --halo early" "--graph whole --halo early" "--graph off --halo ghost" "--graph compute --halo ghost" "--graph whole --halo ghost"; do
    share=""; [ "$k" -gt 1 ] && share="--share $k"
    [ "$k" -eq 1 ] && [ "$mode" != "--graph off" ] && [ "$mode" != "--graph whole" ] && continue   # (one GPU: no collectives, no halo)
    timeout 300 python bench.py --cpu-frames 0 --steps 200 --warmup 20 $share $mode "$@" >> "$out" 2>> "${out%.jsonl}.err" || echo "{\"failed\": \"--share $k $mode\"}" >> "$out"
  done
done
python - "$out" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
print("K  mode           frames  ms/step  flow-kernel ms  launches")
for r in rows:
    if "failed" in r:
        print("FAILED", r["failed"]); continue
    k = r.get("proxy", {}).get("share_of", 1)
    mode = "whole" if "whole step replayed" in r["config"]["workload"] else "compute" if "collectives issued eagerly" in r["config"]["workload"] else "eager"
    mode += "+early" if str(r["config"].get("halo_exchange", "")).startswith("early") else "+ghost" if str(r["config"].get("halo_exchange", "")).startswith("ghost") else ""
    print(f"{k:<2d} {mode:13s} {r['config']['frames_per_gpu']:>6d}  {r['ms_per_step']:.4f}   {r['roofline']['kernel_ms']:.4f}          {r['roofline'].get('launches_per_step')}")
PY
