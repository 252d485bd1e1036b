#!/bin/bash
# rocprofv3 passes for the bench workload (run through gpurun).  Kernel timing and PMC
# counters are collected in SEPARATE runs (never --pmc together with trace domains other
# than --kernel-trace).  Results: gpurun_out/prof_{stats,fetch,write}/
set -u
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps ${STEPS:-20} --warmup 2 --cpu-frames 0 ${BENCH_ARGS:-}"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_stats" -o stats -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/prof_stats.log" 2>&1
echo "stats exit $?" >> "$REPO/gpurun_out/prof_stats.log"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$REPO/gpurun_out/prof_fetch" -o fetch -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/prof_fetch.log" 2>&1
echo "fetch exit $?" >> "$REPO/gpurun_out/prof_fetch.log"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$REPO/gpurun_out/prof_write" -o write -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/prof_write.log" 2>&1
echo "write exit $?" >> "$REPO/gpurun_out/prof_write.log"
cd "$REPO"
python tools/export_profile.py gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/profile_summary.txt 2>&1
tail -30 gpurun_out/profile_summary.txt
