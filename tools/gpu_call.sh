#!/bin/bash
# One gpurun call: environment facts, the GPU test suite, benches.  Usage (from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <tag> [what...]'     what: info tests bench dense track
tag=${1:-call}; shift
what=${*:-info tests bench}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
REPO=$PWD
export FLOWMAP_PARITY_RECORD=$PWD/$out/full_size_parity.jsonl
for w in $what; do
  case $w in
    info) { free -g; nproc; lscpu | grep -i "model name"; rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $out/info.txt 2>&1 ;;
    tests) ( time python -m pytest tests -m gpu -q -rf --durations=8 ) > $out/pytest.log 2>&1; tail -30 $out/pytest.log ;;
    fasttests) ( time FLOWMAP_SKIP_FULL_SIZE=1 python -m pytest tests -m gpu -q -rf --durations=8 ) > $out/pytest.log 2>&1; tail -30 $out/pytest.log ;;
    bench) python bench.py > $out/bench_c1.json 2> $out/bench_c1.err; cat $out/bench_c1.json ;;
    dense) python bench.py --points 0 --cpu-frames 0 > $out/bench_dense.json 2> $out/bench_dense.err; cat $out/bench_dense.json ;;
    track) python bench.py --tracking --cpu-frames 0 > $out/bench_c2.json 2> $out/bench_c2.err; cat $out/bench_c2.json ;;
    run-*)  # run-<name>:<bench args with + for spaces>   -> one bench line
      name=${w#run-}; args=${name#*:}; name=${name%%:*}; args=${args//+/ }
      python bench.py --cpu-frames 0 $args > $out/bench_$name.json 2> $out/bench_$name.err; cat $out/bench_$name.json; tail -3 $out/bench_$name.err ;;
    prof-*)  # prof-<name>:<bench args with + for spaces>, e.g. prof-dense:--points+0   -> rocprofv3 kernel stats
      name=${w#prof-}; args=${name#*:}; name=${name%%:*}; args=${args//+/ }
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_$name -o stats -- python $REPO/bench.py --steps 10 --warmup 3 --cpu-frames 0 $args > /dev/null 2> $REPO/$out/prof_$name.err)
      python tools/export_profile.py $out/prof_$name > $out/${name}_rocprofv3_summary.csv 2>> $out/prof_$name.err; rm -rf $out/prof_$name
      head -14 $out/${name}_rocprofv3_summary.csv ;;
  esac
done
