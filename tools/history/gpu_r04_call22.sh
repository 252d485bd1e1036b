#!/bin/bash
# round 4: the losses seed their own backward (RootLoss: no ones_like fill, no is-the-seed-one launch) — new GPU tests, the driver's command with
# and without it (FLOWMAP_PLAIN_LOSS=1), C2, and install() on the stand-in package with the HIP library
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
timeout 900 python3 -m pytest tests/test_install_standin.py tests/test_gpu_parity.py -q -m gpu -k "standin or seed or stray or tap_exchange or in_pass" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for run in 1 2; do
  for mode in seeded plain; do
    env=""; [ $mode = plain ] && env="FLOWMAP_PLAIN_LOSS=1"
    env $env timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > $out/driver_${mode}_$run.json 2> $out/driver_${mode}_$run.err
    python3 - $out/driver_${mode}_$run.json $mode$run <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line); r = d["roofline"]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "kernel avg %.4f" % (r.get("kernel_ms") or 0), "frac %.3f" % r["frac"], "launches", r.get("launches_per_step"), "step_frac %.3f" % r.get("step_frac", 0))
PY
  done
done
for mode in seeded plain; do
  env=""; [ $mode = plain ] && env="FLOWMAP_PLAIN_LOSS=1"
  env $env timeout 300 python3 bench.py --config c2 --steps 100 --warmup 20 --cpu-frames 0 > $out/c2_$mode.json 2> $out/c2_$mode.err
  python3 -c "
import json,sys
d=json.loads([l for l in open('$out/c2_$mode.json') if l.startswith('{')][-1]); print('c2 $mode ms/step %.4f'%d['ms_per_step'], 'launches', d['roofline'].get('launches_per_step'))"
done
