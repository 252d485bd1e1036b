#!/bin/bash
# One iteration of dense-Procrustes tuning on the GPU box: dense parity tests, the microbench, the dense bench line and the VALU count
# of the dense kernels (a PMC pass of its own).
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/dense
FLOWMAP_SKIP_FULL_SIZE=1 timeout 300 python -m pytest tests -x -q -m gpu -k "dense" 2>&1 | tail -3
timeout 300 python tools/dense_microbench.py 150 ${FLOWS:-gentle} 2>&1 | grep -v "amdgpu.ids\|^{" | tee gpurun_out/dense/microbench.txt
timeout 120 python bench.py --points 0 --cpu-frames 0 --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/dense/bench_dense.json
python -c "import json;d=json.load(open('gpurun_out/dense/bench_dense.json'));print('dense step ms', d['ms_per_step'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d /tmp/prof_sq -o sq -- python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-frames 0 --points 0 > /tmp/prof_sq.log 2>&1
cd "$REPO"
python - <<'PY'
import glob, sqlite3
hits = glob.glob("/tmp/prof_sq/**/*.db", recursive=True)
con = sqlite3.connect(hits[0])
for k in ("dense_bwd_fused", "moments_dense", "flow_fused_kernel<4, 0, true, true"):
    row = {r[0]: r[1] for r in con.execute(f"select counter_name, avg(value) from counters_collection where kernel_name like '%{k}%' group by counter_name")}
    print(k, {n: round(v / 2.1456e6, 1) for n, v in row.items() if n != "GRBM_GUI_ACTIVE"}, "per wave-pixel; GUI cycles/8", round(row.get("GRBM_GUI_ACTIVE", 0) / 8))
PY
