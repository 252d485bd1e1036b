cd $GRAFT_REPO_ROOT
for flag in True False; do
for args in "" "--height 180 --width 240" "--height 180 --width 240 --tracking --optimizer fused"; do
python - --cpu-frames 0 $args <<PY 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain=$flag', '$args', round(d['ms_per_step'],4))"
import sys, runpy
import flowmap_amd._ops as o
o.use_fit_chain = $flag
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')
PY
done; done
