#!/bin/bash
# round 4: the dense Procrustes backward with four pixels per thread (16-byte loads / stores) against round 3's one pixel per thread
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04m; mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rf -k "dense" ) > $out/pytest_dense.log 2>&1; tail -4 $out/pytest_dense.log
timeout 900 python3 tools/dense_microbench.py 150 gentle,smooth,iid > $out/dense_microbench.txt 2> $out/dense_microbench.err; grep -v "^{" $out/dense_microbench.txt | cut -c1-330; tail -3 $out/dense_microbench.err
timeout 400 python3 bench.py --points 0 --cpu-frames 0 --steps 50 --warmup 10 --sustained-steps 0 > $out/bench_dense.json 2> $out/bench_dense.err; python3 -c "
import json; r=json.load(open('$out/bench_dense.json')); print('dense step ms', round(r['ms_per_step'],4))"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_dense -o d -- python3 $REPO/bench.py --points 0 --steps 20 --warmup 5 --cpu-frames 0 --sustained-steps 0) > $out/prof_dense.log 2>&1
python3 tools/export_profile.py $out/prof_dense > $out/dense_rocprofv3_summary.csv 2>> $out/prof_dense.log; rm -rf $out/prof_dense; head -8 $out/dense_rocprofv3_summary.csv | cut -c1-160
