#!/bin/bash
# round 4: does the tap exchange pay at the reference's default 180x240 once the step is replayed as a hipGraph (no host bookkeeping left)?
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04o; mkdir -p $out
for opt in "" "--optimizer fused" "--optimizer in_pass"; do
  for graph in "" "--graph whole"; do
    for mb in default 0; do
      env=""; [ $mb = 0 ] && env="FLOWMAP_TAP_EXCHANGE_MIN_BYTES=0"
      name=$(echo "c2_180 $opt $graph min_$mb" | tr ' ' '_' | tr -d '-')
      env $env timeout 300 python3 bench.py --config c2 --height 180 --width 240 --steps 300 --warmup 30 --cpu-frames 0 --sustained-steps 0 $opt $graph > $out/$name.json 2> $out/$name.err
      python3 -c "
import json
try:
    d=json.loads([l for l in open('$out/$name.json') if l.startswith('{')][-1]); t=d.get('roofline_tracking',{}).get('tap_exchange')
    print('$name', 'ms/step %.4f'%d['ms_per_step'], t)
except Exception as e:
    print('$name FAILED', e)"
    done
  done
done
