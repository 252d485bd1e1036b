#!/bin/bash
# round 4, second GPU call: the tap exchange between the fused flow loss and the fused tracking loss — parity (case_tap_exchange, C2 at full
# size on its third step), C2 and 180x240 bench lines with and without it, the kernel table of a C2 step.
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04b; mkdir -p $out
export TMPDIR=/tmp
export FLOWMAP_PARITY_RECORD=$PWD/$out/full_size_parity.jsonl
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rf ) > $out/pytest_parity.log 2>&1; tail -6 $out/pytest_parity.log
( time timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -rf -k "c2 or c1" ) > $out/pytest_full_c2.log 2>&1; tail -6 $out/pytest_full_c2.log
b() { name=$1; shift; timeout 400 python3 bench.py --steps 100 --warmup 20 --cpu-frames 0 --sustained-steps 0 "$@" > $out/bench_$name.json 2> $out/bench_$name.err; python3 - "$out/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    t = r.get("roofline_tracking", {})
    print(sys.argv[2], "ms/step", round(r["ms_per_step"], 4), "flow kernel", round(r["roofline"]["kernel_ms"], 4), "frac", round(r["roofline"]["frac"], 3),
          "track call ms", round(t.get("kernel_ms", 0), 4), "track frac", round(t.get("frac", 0), 3), t.get("tap_exchange"), "launches", r["roofline"]["launches_per_step"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
b c2_exchange --config c2
b c2_round3 --config c2 --no-tap-exchange
b c2_exchange_torch_adam --config c2 --optimizer torch
b c2_round3_torch_adam --config c2 --optimizer torch --no-tap-exchange
b c1 --config c1
b small_exchange --config c2 --height 180 --width 240
b small_round3 --config c2 --height 180 --width 240 --no-tap-exchange
for v in exchange round3; do
  extra=""; [ $v = round3 ] && extra="--no-tap-exchange"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_c2_$v -o c2 -- python3 $REPO/bench.py --config c2 --steps 20 --warmup 5 --cpu-frames 0 --sustained-steps 0 $extra) > $out/prof_c2_$v.log 2>&1
  f=$(find $out/prof_c2_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_${v}_kernel_stats.csv && head -14 $f | cut -c1-150
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_small -o s -- python3 $REPO/bench.py --config c2 --height 180 --width 240 --steps 20 --warmup 5 --cpu-frames 0 --sustained-steps 0) > $out/prof_small.log 2>&1
f=$(find $out/prof_small -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/small_exchange_kernel_stats.csv && head -10 $f | cut -c1-150
rm -rf $out/prof_c2_exchange $out/prof_c2_round3 $out/prof_small
