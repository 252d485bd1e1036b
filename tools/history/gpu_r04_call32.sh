#!/bin/bash
# round 4: RootLoss without Python dispatch (torch function off, the loss re-typed in place): host-bound 180x240 eager, seeded vs plain, alternating; C1 driver command
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04r; mkdir -p $out
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_install_standin.py -q -m gpu -k "seed or stray or standin or tap_exchange" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for run in 1 2 3; do
  for mode in seeded plain; do
    env=""; [ $mode = plain ] && env="FLOWMAP_PLAIN_LOSS=1"
    env $env timeout 300 python3 bench.py --config c2 --height 180 --width 240 --steps 300 --warmup 30 --cpu-frames 0 --sustained-steps 0 > $out/s_${mode}_$run.json 2> $out/s_${mode}_$run.err
    python3 -c "
import json
d=json.loads([l for l in open('$out/s_${mode}_$run.json') if l.startswith('{')][-1]); print('180x240 $mode $run ms/step %.4f'%d['ms_per_step'], 'launches', d['roofline'].get('launches_per_step'))"
  done
done
for mode in seeded plain; do
  env=""; [ $mode = plain ] && env="FLOWMAP_PLAIN_LOSS=1"
  env $env timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > $out/c1_$mode.json 2> $out/c1_$mode.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/c1_$mode.json') if l.startswith('{')][-1]); print('c1 $mode ms/step %.4f'%d['ms_per_step'], 'launches', d['roofline'].get('launches_per_step'))"
done
