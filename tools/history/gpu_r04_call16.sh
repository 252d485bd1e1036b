#!/bin/bash
# round 4: HBM traffic of the flow pass's TAPS instances (C2, with and without the in-pass Adam update): rocprofv3 PMC passes, FETCH_SIZE and WRITE_SIZE separately
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04p; mkdir -p $out
export TMPDIR=/tmp
for mode in plain in_pass; do
  extra=""; [ $mode = in_pass ] && extra="--optimizer in_pass"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $REPO/$out/pmc_${mode}_$ctr -o p -- python3 $REPO/bench.py --config c2 --steps 20 --warmup 3 --cpu-frames 0 --sustained-steps 0 $extra) > $out/pmc_${mode}_$ctr.log 2>&1
  done
  python3 tools/export_profile.py $out/pmc_${mode}_FETCH_SIZE $out/pmc_${mode}_FETCH_SIZE $out/pmc_${mode}_WRITE_SIZE > $out/pmc_${mode}_summary.txt 2>&1
  grep -i "flow_fused\|FETCH\|WRITE\|track_pairs\|tap_grad" $out/pmc_${mode}_summary.txt | head -20 | cut -c1-200
  rm -rf $out/pmc_${mode}_FETCH_SIZE $out/pmc_${mode}_WRITE_SIZE
done
