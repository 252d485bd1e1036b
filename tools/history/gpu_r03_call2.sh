#!/bin/bash
# round 3: fast GPU tests, headline bench (no CPU leg), strong-scaling proxy (eager / compute-graph / whole-graph), kernel tables.
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/${1:-r03b}; mkdir -p $out
export TMPDIR=/tmp
( time FLOWMAP_SKIP_FULL_SIZE=1 timeout 900 python -m pytest tests -m gpu -q -x -rf --durations=5 ) > $out/pytest.log 2>&1; tail -12 $out/pytest.log
timeout 600 python bench.py --cpu-frames 0 > $out/bench_c1.json 2> $out/bench_c1.err; cut -c1-700 $out/bench_c1.json; tail -3 $out/bench_c1.err
bash tools/scaling_proxy.sh $out/strong_scaling_proxy.jsonl
tail -5 $out/strong_scaling_proxy.err
prof() {  # prof <name> <bench args...>
  local name=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_$name -o stats -- python $REPO/bench.py --steps 20 --warmup 3 --cpu-frames 0 "$@" > /dev/null 2> $REPO/$out/prof_$name.err)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --cpu-frames 0 $*   (calls = 3 set-up + 3 warm-up + 20 timed steps)"; python tools/export_profile.py $out/prof_$name; } > $out/r03_${name}_rocprofv3_summary.csv 2>> $out/prof_$name.err; rm -rf $out/prof_$name
  head -14 $out/r03_${name}_rocprofv3_summary.csv
}
prof c1_bench
prof share8 --share 8
