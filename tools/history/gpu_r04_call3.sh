#!/bin/bash
# round 4, third GPU call: the tap exchange with its loads requested ahead (flow pass), where a wave of track_pairs spends its time with
# and without the compact tap image (phase clocks), kernel tables of a C2 step.
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04d; mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rf ) > $out/pytest_parity.log 2>&1; tail -4 $out/pytest_parity.log
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
b small_exchange --config c2 --height 180 --width 240
b small_round3 --config c2 --height 180 --width 240 --no-tap-exchange
for v in exchange round3; do
  extra=""; [ $v = round3 ] && extra="--no-tap-exchange"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_c2_$v -o c2 -- python3 $REPO/bench.py --config c2 --steps 20 --warmup 5 --cpu-frames 0 --sustained-steps 0 $extra) > $out/prof_c2_$v.log 2>&1
  python3 tools/export_profile.py $out/prof_c2_$v > $out/c2_${v}_rocprofv3_summary.csv 2>> $out/prof_c2_$v.log; rm -rf $out/prof_c2_$v; head -16 $out/c2_${v}_rocprofv3_summary.csv | cut -c1-160
done
cp -r flowmap_amd/libflowmap_hip.so /tmp/libflowmap_hip.so.keep
timeout 600 python3 tools/track_clocks.py > $out/track_clocks_exchange.txt 2> $out/track_clocks_exchange.err; tail -9 $out/track_clocks_exchange.txt
FLOWMAP_NO_TAP_EXCHANGE=1 timeout 600 python3 tools/track_clocks.py > $out/track_clocks_round3.txt 2> $out/track_clocks_round3.err; tail -9 $out/track_clocks_round3.txt
