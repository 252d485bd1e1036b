#!/bin/bash
# round 4: the tap image sampled branch-free in track_pairs' prologue — parity (tap exchange tests, multi-step fuzz), C2 lines, phase clocks
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04t; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -m gpu -k "tap or exchange or tracking or in_pass" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 600 python3 tests/tools/extended_fuzz.py --device cuda:0 --count 120 --seed 41 --steps 3 --tracks > $out/fuzz.txt 2>&1; tail -2 $out/fuzz.txt
for i in 1 2; do
  timeout 300 python3 bench.py --config c2 --steps 100 --warmup 20 --cpu-frames 0 > $out/c2_$i.json 2> $out/c2_$i.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/c2_$i.json') if l.startswith('{')][-1]); t=d.get('roofline_tracking',{})
print('c2 run $i ms/step %.4f'%d['ms_per_step'], 'flow kernel %.4f'%d['roofline']['kernel_ms'], 'tracking', {k:t[k] for k in t if k in ('kernel_ms','frac','call_ms')})"
done
timeout 300 python3 bench.py --config c2 --steps 60 --warmup 20 --cpu-frames 0 --optimizer in_pass > $out/c2_in_pass.json 2> $out/c2_in_pass.err
python3 -c "
import json
d=json.loads([l for l in open('$out/c2_in_pass.json') if l.startswith('{')][-1]); print('c2 in-pass adam ms/step %.4f'%d['ms_per_step'])"
timeout 600 python3 tools/track_clocks.py > $out/clocks.txt 2>&1; grep -v "^{" $out/clocks.txt | grep -A 8 "^1980 waves" | head -12
