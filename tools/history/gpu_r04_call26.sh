#!/bin/bash
# round 4: the ghost halo with its forward end in one launch (fm_halo_ghost_begin: poses packed + compact baseline) and the losses
# seeding their own backward: halo kernels on the GPU, the one-GPU proxy's rows again (K = 1, 2, 4, 8; one-shot / ghost; compute / whole), the
# kernel table of one rank's share of 8
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04x; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "halo or ghost or shard" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
: > $out/proxy.jsonl
for k in 1 2 4 8; do
  for mode in "--graph off" "--graph compute" "--graph whole" "--graph compute --halo ghost" "--graph whole --halo ghost"; do
    share=""; [ "$k" -gt 1 ] && share="--share $k"
    [ "$k" -eq 1 ] && [ "$mode" != "--graph off" ] && [ "$mode" != "--graph whole" ] && continue
    timeout 300 python3 bench.py --cpu-frames 0 --steps 200 --warmup 20 $share $mode >> $out/proxy.jsonl 2>> $out/proxy.err || echo "{\"failed\": \"--share $k $mode\"}" >> $out/proxy.jsonl
  done
done
python3 - $out/proxy.jsonl <<'PY'
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
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_share8 -o s -- python3 $REPO/bench.py --cpu-frames 0 --steps 30 --warmup 5 --share 8 --graph off --halo ghost --sustained-steps 0) > $out/prof_share8.log 2>&1
python3 tools/export_profile.py $out/prof_share8 > $out/share8_ghost_rocprofv3_summary.csv 2>&1; rm -rf $out/prof_share8
head -16 $out/share8_ghost_rocprofv3_summary.csv | cut -c1-150
