#!/bin/bash
# round 4, end of round: the other workloads of DESIGN §6 on the final build — C2 kernel table, the reference's default resolution (eager and replayed), dense Procrustes
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04e2; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_c2 -o s -- python3 $REPO/bench.py --config c2 --steps 20 --warmup 5 --cpu-frames 0 --sustained-steps 0) > $out/prof_c2.log 2>&1
python3 tools/export_profile.py $out/prof_c2 > $out/c2_rocprofv3_summary.csv 2>&1; rm -rf $out/prof_c2
head -24 $out/c2_rocprofv3_summary.csv | cut -c1-150
for mode in "" "--graph whole" "--optimizer fused" "--optimizer fused --graph whole"; do
  name=$(echo "180x240 $mode" | tr ' ' '_' | tr -d '-')
  timeout 300 python3 bench.py --config c2 --height 180 --width 240 --steps 200 --warmup 20 --cpu-frames 0 $mode > $out/$name.json 2> $out/$name.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$name.json') if l.startswith('{')][-1]); print('$name', 'ms/step %.4f'%d['ms_per_step'], 'launches', d['roofline'].get('launches_per_step'))"
done
timeout 300 python3 bench.py --points 0 --steps 50 --warmup 10 --cpu-frames 0 > $out/dense.json 2> $out/dense.err
python3 -c "
import json
d=json.loads([l for l in open('$out/dense.json') if l.startswith('{')][-1]); print('dense ms/step %.4f'%d['ms_per_step'], 'launches', d['roofline'].get('launches_per_step'))"
