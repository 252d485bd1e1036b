cd /tmp && export TMPDIR=/tmp
cat > /tmp/run_nochain.py <<PY
import sys, runpy
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import flowmap_amd._ops as o
o.use_fit_chain = False
sys.argv = ['bench.py', '--cpu-frames', '0', '--steps', '20', '--warmup', '3', '--height', '180', '--width', '240', '--tracking', '--optimizer', 'fused']
runpy.run_path("$GRAFT_REPO_ROOT/bench.py", run_name='__main__')
PY
rocprofv3 --kernel-trace --stats -d /tmp/pnc -o stats -- python /tmp/run_nochain.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/export_profile.py /tmp/pnc | head -34 | cut -c1-120
