#!/bin/bash
# One iteration of flow-kernel tuning on the GPU box: the GPU parity tests (without the full-size module unless FULL=1), the C1 bench
# line, and the VALU count of the fused flow kernel (a PMC pass of its own).
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/flow
if [ "${FULL:-0}" = "1" ]; then timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4; else FLOWMAP_SKIP_FULL_SIZE=1 timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -4; fi
for i in 1 2; do timeout 200 python bench.py --cpu-frames 0 2>&1 | tail -1 > gpurun_out/flow/bench_c1_$i.json; python -c "import json;d=json.load(open('gpurun_out/flow/bench_c1_$i.json'));print('C1 step ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'step_frac', d['roofline']['step_frac'])"; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d /tmp/prof_sq -o sq -- python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-frames 0 > /tmp/prof_sq.log 2>&1
cd "$REPO"
python - <<'PY'
import glob, sqlite3
hits = glob.glob("/tmp/prof_sq/**/*.db", recursive=True)
con = sqlite3.connect(hits[0])
for k in ("flow_fused_kernel<4, 0, true, true",):
    row = {r[0]: r[1] for r in con.execute(f"select counter_name, avg(value) from counters_collection where kernel_name like '%{k}%' group by counter_name")}
    print(k, {n: round(v / 2.16e6, 1) for n, v in row.items() if n != "GRBM_GUI_ACTIVE"}, "per wave-pixel; GUI cycles/8", round(row.get("GRBM_GUI_ACTIVE", 0) / 8))
PY
