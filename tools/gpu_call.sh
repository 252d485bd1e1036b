#!/bin/bash
# One gpurun call: environment facts, the GPU test suite, benches.  Usage (from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <tag> [what...]'     what: info tests bench dense track
tag=${1:-call}; shift
what=${*:-info tests bench}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
REPO=$PWD
export FLOWMAP_PARITY_RECORD=$PWD/$out/full_size_parity.jsonl
export FLOWMAP_FOCAL_LOG=$PWD/$out/focal_gate_ratios.txt   # tests/helpers.focal_close: err / (2^-24 x sum of |terms|) of every dL/dfocal comparison
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
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_$name -o stats -- python $REPO/bench.py --steps ${PROF_STEPS:-20} --warmup ${PROF_WARMUP:-5} --cpu-frames 0 --sustained-steps 0 --ate off $args > /dev/null 2> $REPO/$out/prof_$name.err)
      python tools/export_profile.py $out/prof_$name > $out/${name}_rocprofv3_summary.csv 2>> $out/prof_$name.err; rm -rf $out/prof_$name
      head -14 $out/${name}_rocprofv3_summary.csv ;;
    sq-*)  # sq-<name>:<bench args>   -> VALU / SALU / LDS instructions per wave, cycles, VALU share of the issue slots per fm:: kernel (a PMC pass of its own)
      name=${w#sq-}; args=${name#*:}; name=${name%%:*}; args=${args//+/ }
      BENCH_ARGS="$args" bash tools/gpu_sq_counters.sh > $out/${name}_sq_counters.csv 2> $out/sq_$name.err; cat $out/${name}_sq_counters.csv ;;
    pmc-*)  # pmc-<name>:<bench args>  -> HBM bytes per launch of every fm:: kernel: FETCH_SIZE and WRITE_SIZE in separate passes (the guide's recipe)
      name=${w#pmc-}; args=${name#*:}; name=${name%%:*}; args=${args//+/ }
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && rm -rf /tmp/pmc_$c && timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o pmc -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-frames 0 --sustained-steps 0 --ate off $args > /dev/null 2> $REPO/$out/pmc_${name}_$c.err)
      done
      python tools/export_profile.py --pmc-only /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $out/${name}_pmc.txt 2>> $out/pmc_${name}_FETCH_SIZE.err; head -30 $out/${name}_pmc.txt ;;
    hostprof-*)  # hostprof-<name>:<args of tools/host_profile_default.py>
      name=${w#hostprof-}; args=${name#*:}; name=${name%%:*}; args=${args//+/ }
      python tools/host_profile_default.py $args > $out/host_profile_$name.txt 2>&1; head -12 $out/host_profile_$name.txt ;;
  esac
done
