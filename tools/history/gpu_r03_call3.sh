#!/bin/bash
# round 3: microbenches of the Procrustes kernels, fast GPU tests, headline bench, proxy, kernel tables; then (optional) full-size parity + ATE
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/${1:-r03c}; mkdir -p $out
export TMPDIR=/tmp
python tools/phase_clocks.py 2>&1 | tail -2 | tee $out/phase_clocks_150.txt
python tools/phase_clocks.py 20 720 1280 2>&1 | tail -2 | tee $out/phase_clocks_20.txt
python tools/fit_microbench.py 2>&1 | tail -1 | tee $out/fit_microbench_150.json
python tools/fit_microbench.py 20 720 1280 2>&1 | tail -1 | tee $out/fit_microbench_20.json
( time FLOWMAP_SKIP_FULL_SIZE=1 timeout 900 python -m pytest tests -m gpu -q -x -rf ) > $out/pytest.log 2>&1; tail -6 $out/pytest.log
timeout 600 python bench.py --cpu-frames 0 > $out/bench_c1.json 2> $out/bench_c1.err; cut -c1-500 $out/bench_c1.json; tail -2 $out/bench_c1.err
bash tools/scaling_proxy.sh $out/strong_scaling_proxy.jsonl
prof() {  # prof <name> <bench args...>
  local name=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_$name -o stats -- python $REPO/bench.py --steps 20 --warmup 3 --cpu-frames 0 "$@" > /dev/null 2> $REPO/$out/prof_$name.err)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --cpu-frames 0 $*   (calls = 3 set-up + 3 warm-up + 20 timed steps)"; python tools/export_profile.py $out/prof_$name; } > $out/r03_${name}_rocprofv3_summary.csv 2>> $out/prof_$name.err; rm -rf $out/prof_$name
  head -13 $out/r03_${name}_rocprofv3_summary.csv
}
prof c1_bench
prof share8 --share 8
if [ "$2" = "full" ]; then
  export FLOWMAP_PARITY_RECORD=$REPO/$out/r03_full_size_parity.jsonl
  ( time timeout 1500 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -rf ) > $out/pytest_full.log 2>&1; tail -8 $out/pytest_full.log
  cat $FLOWMAP_PARITY_RECORD
fi
if [ -f tests/golden/ate_150x360x640_reference.json ]; then
  python tests/tools/ate_full_chain.py --leg ours 2> $out/ate.err | tail -1 > $out/r03_ate_150x360x640.json; cat $out/r03_ate_150x360x640.json; tail -2 $out/ate.err
  python tests/tools/ate_full_chain.py --leg ours --in-pass 2>> $out/ate.err | tail -1 > $out/r03_ate_150x360x640_in_pass.json; cat $out/r03_ate_150x360x640_in_pass.json
fi
