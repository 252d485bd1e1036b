#!/bin/bash
# round 4, seventh GPU call: the ghost halo (parity of fm_flow_ghost_terms, the one-GPU proxy of a K-rank share with it), the ATE leg against the
# imported reference's record, the GPU suite's fast part.
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04g; mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -rf ) > $out/pytest_parity.log 2>&1; tail -4 $out/pytest_parity.log
proxy=$out/strong_scaling_proxy_ghost.jsonl; : > $proxy
timeout 300 python bench.py --cpu-frames 0 --steps 200 --warmup 20 --sustained-steps 0 >> $proxy 2>> $out/proxy.err
for k in 4 8; do
  for mode in "--graph compute" "--graph compute --halo early" "--graph compute --halo ghost" "--graph whole --halo ghost" "--graph off --halo ghost"; do
    timeout 300 python bench.py --cpu-frames 0 --steps 200 --warmup 20 --sustained-steps 0 --share $k $mode >> $proxy 2>> $out/proxy.err || echo "{\"failed\": \"--share $k $mode\"}" >> $proxy
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
timeout 900 python tests/tools/ate_full_chain.py --leg ours --reference tests/golden/ate_150x360x640_imported_reference.json --out $out/ate_150x360x640_vs_imported_reference.json > $out/ate.log 2>&1; tail -3 $out/ate.log | cut -c1-600
