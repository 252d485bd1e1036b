#!/bin/bash
# A/B of fm_track.hip build flags on the GPU box (FM_TRACK_PG: points per lane of track_pairs; FM_TRACK_AHEAD: target frames in flight):
# C2 step and the tracking kernels' time with the shipped build first, then a rebuild per entry of VARIANTS (the box has hipcc; nothing is
# written back).    VARIANTS="-DFM_TRACK_AHEAD=8 -DFM_TRACK_PG=1" bash tools/ab_track.sh
set -e
cd $GRAFT_REPO_ROOT
line() { python bench.py --config c2 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline_tracking']['kernel_ms'])"; }
FLOWMAP_SKIP_FULL_SIZE=1 python -m pytest tests -m gpu -q -k "track" 2>&1 | tail -1
line shipped; line shipped
for v in ${VARIANTS:-}; do
python - <<PY
import flowmap_amd.build as b
b.FILE_FLAGS["fm_track.hip"] = "$v".split(",")
b.build_library(force=True, verbose=False)
PY
line "$v"; line "$v"
done
