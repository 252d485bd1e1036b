#!/bin/bash
# round 3, first GPU call: fast GPU tests, the headline bench line, the strong-scaling proxy, the C1 kernel table.
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r03a; mkdir -p $out
export TMPDIR=/tmp
( time FLOWMAP_SKIP_FULL_SIZE=1 timeout 900 python -m pytest tests -m gpu -q -x -rf --durations=8 ) > $out/pytest.log 2>&1; tail -15 $out/pytest.log
timeout 600 python bench.py > $out/bench_c1.json 2> $out/bench_c1.err; cat $out/bench_c1.json; tail -3 $out/bench_c1.err
FLOWMAP_THREE_LAUNCH_BWD=1 timeout 300 python bench.py --cpu-frames 0 > $out/bench_c1_three_launch.json 2> $out/bench_c1_three_launch.err; cut -c1-400 $out/bench_c1_three_launch.json
bash tools/scaling_proxy.sh $out/strong_scaling_proxy.jsonl
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_c1 -o stats -- python $REPO/bench.py --steps 20 --warmup 3 --cpu-frames 0 > /dev/null 2> $REPO/$out/prof_c1.err)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --cpu-frames 0   (calls = 3 set-up + 3 warm-up + 20 timed steps)"; python tools/export_profile.py $out/prof_c1; } > $out/r03_c1_bench_rocprofv3_summary.csv 2>> $out/prof_c1.err; rm -rf $out/prof_c1
head -16 $out/r03_c1_bench_rocprofv3_summary.csv
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_s8 -o stats -- python $REPO/bench.py --steps 20 --warmup 3 --cpu-frames 0 --share 8 > /dev/null 2> $REPO/$out/prof_s8.err)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --cpu-frames 0 --share 8"; python tools/export_profile.py $out/prof_s8; } > $out/r03_share8_rocprofv3_summary.csv 2>> $out/prof_s8.err; rm -rf $out/prof_s8
head -20 $out/r03_share8_rocprofv3_summary.csv
