#!/bin/bash
# round 4, eighth GPU call: why the ghost halo crashes under hipGraph capture (faulthandler), C2 with the Adam step (the in-pass update now engages with the
# tap exchange), the new GPU tests
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04h; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -X faulthandler bench.py --cpu-frames 0 --steps 20 --warmup 5 --sustained-steps 0 --share 8 --graph compute --halo ghost > $out/ghost_graph.json 2> $out/ghost_graph.err; echo "ghost+graph exit $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $out/ghost_graph.err | tail -40 | cut -c1-200
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rf -k "tap or ghost or in_pass or adam" ) > $out/pytest_parity.log 2>&1; tail -4 $out/pytest_parity.log
b() { name=$1; shift; timeout 400 python3 bench.py --steps 100 --warmup 20 --cpu-frames 0 --sustained-steps 0 "$@" > $out/bench_$name.json 2> $out/bench_$name.err; python3 - "$out/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    t = r.get("roofline_tracking", {})
    print(sys.argv[2], "ms/step", round(r["ms_per_step"], 4), "flow kernel", round(r["roofline"]["kernel_ms"], 4), "frac", round(r["roofline"]["frac"], 3),
          "track call ms", round(t.get("kernel_ms", 0), 4), t.get("tap_exchange"), "launches", r["roofline"]["launches_per_step"], r["config"]["workload"][-120:])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
b c2_in_pass_exchange --config c2 --optimizer in_pass
b c2_in_pass_round3 --config c2 --optimizer in_pass --no-tap-exchange
b c2_fused_exchange --config c2 --optimizer fused
b c2_fused_round3 --config c2 --optimizer fused --no-tap-exchange
b c2_torch_exchange --config c2 --optimizer torch
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_c2_in_pass -o c2 -- python3 $REPO/bench.py --config c2 --optimizer in_pass --steps 20 --warmup 5 --cpu-frames 0 --sustained-steps 0) > $out/prof_c2_in_pass.log 2>&1
python3 tools/export_profile.py $out/prof_c2_in_pass > $out/c2_in_pass_exchange_rocprofv3_summary.csv 2>> $out/prof_c2_in_pass.log; rm -rf $out/prof_c2_in_pass; head -14 $out/c2_in_pass_exchange_rocprofv3_summary.csv | cut -c1-160
