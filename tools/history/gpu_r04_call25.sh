#!/bin/bash
# round 4: phase clocks of the sparse Procrustes fit (one block per pair) at C1 and at one rank's share of 8 GPUs
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04y; mkdir -p $out
SRC=fm_procrustes.hip bash tools/build_variants.sh clocks:-DFM_PHASE_CLOCKS > $out/build.log 2>&1
timeout 300 python3 tools/phase_clocks_fit.py > $out/fit_phase_clocks_c1.txt 2>&1; cat $out/fit_phase_clocks_c1.txt | tail -6
timeout 300 python3 tools/phase_clocks_fit.py 20 720 1280 > $out/fit_phase_clocks_share8.txt 2>&1; cat $out/fit_phase_clocks_share8.txt | tail -6
