set -e
cd $GRAFT_REPO_ROOT
python bench.py --config c2 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PG2', d['ms_per_step'], d['roofline_tracking']['kernel_ms'])"
python bench.py --config c2 --cpu-frames 0 --height 180 --width 240 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PG2 small', d['ms_per_step'], d['roofline_tracking']['kernel_ms'])"
FLOWMAP_SKIP_FULL_SIZE=1 python -m pytest tests -m gpu -q -k "track" 2>&1 | tail -2
python - <<'PY'
import flowmap_amd.build as b
b.FILE_FLAGS["fm_track.hip"] = ["-DFM_TRACK_PG=1"]
b.build_library(force=True, verbose=False)
PY
python bench.py --config c2 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PG1', d['ms_per_step'], d['roofline_tracking']['kernel_ms'])"
python bench.py --config c2 --cpu-frames 0 --height 180 --width 240 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PG1 small', d['ms_per_step'], d['roofline_tracking']['kernel_ms'])"
