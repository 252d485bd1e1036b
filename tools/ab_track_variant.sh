#!/bin/bash
# A/B of a PRE-BUILT variant of the library (build_variants/libflowmap_hip_<name>.so, cross-compiled in the build container: no hipcc time on the
# GPU box) against the shipped one: the tracking tests and the randomised sweep on the variant, C2 and 180x240 lines of both, the variant's kernel table.
#   gpurun -- 'NAME=mfma bash tools/ab_track_variant.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
name=${NAME:-mfma}; out=gpurun_out/ab_track_$name; mkdir -p $out
line() { python bench.py $2 --cpu-frames 0 --sustained-steps 0 --ate off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline_tracking']['kernel_ms'],4))" | tee -a $out/lines.txt; }
line shipped_c2 "--config c2"; line shipped_180 "--height 180 --width 240 --tracking"
cp flowmap_amd/libflowmap_hip.so /tmp/shipped.so
cp build_variants/libflowmap_hip_$name.so flowmap_amd/libflowmap_hip.so
(FLOWMAP_SKIP_FULL_SIZE=1 python -m pytest tests -m gpu -q -k "track or step or fuzz" 2>&1 | tail -4) | tee $out/pytest_track.txt
(python -m pytest tests/test_gpu_full_size.py -m gpu -q -k "c2" 2>&1 | tail -3) | tee $out/pytest_c2_full.txt
(python tests/tools/extended_fuzz.py --device cuda:0 --count 120 --seed 23 --steps 3 --tracks 2>&1 | tail -3) | tee $out/fuzz.txt
line ${name}_c2 "--config c2"; line ${name}_180 "--height 180 --width 240 --tracking"
line ${name}_c2 "--config c2"; line ${name}_180 "--height 180 --width 240 --tracking"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o stats -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps 20 --warmup 5 --cpu-frames 0 --sustained-steps 0 --ate off > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err)
python tools/export_profile.py $out/prof > $out/c2_${name}_rocprofv3_summary.csv 2>> $out/prof.err; rm -rf $out/prof; head -8 $out/c2_${name}_rocprofv3_summary.csv
cp /tmp/shipped.so flowmap_amd/libflowmap_hip.so
line shipped_c2 "--config c2"; line shipped_180 "--height 180 --width 240 --tracking"
