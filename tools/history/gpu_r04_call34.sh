#!/bin/bash
# round 4: eager steps at small sizes, tap exchange off (default below 128 MB) vs forced on, alternating
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04n; mkdir -p $out
for size in "180 240" "90 120" "270 480"; do
  set -- $size
  for opt in "" "--optimizer fused"; do
    for run in 1 2 3; do
      for mb in default 0; do
        env=""; [ $mb = 0 ] && env="FLOWMAP_TAP_EXCHANGE_MIN_BYTES=0"
        name=$(echo "c2_$1x$2 $opt min_$mb $run" | tr ' ' '_' | tr -d '-')
        env $env timeout 300 python3 bench.py --config c2 --height $1 --width $2 --steps 300 --warmup 30 --cpu-frames 0 --sustained-steps 0 $opt > $out/$name.json 2> $out/$name.err
        python3 -c "
import json
try:
    d=json.loads([l for l in open('$out/$name.json') if l.startswith('{')][-1]); print('$name', 'ms/step %.4f'%d['ms_per_step'])
except Exception as e:
    print('$name FAILED', e)"
      done
    done
  done
done
