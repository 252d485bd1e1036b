#!/bin/bash
# SQ-level PMC pass on the bench workload (separate from timing runs).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d "$REPO/gpurun_out/prof_sq" -o sq -- python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-frames 0 ${BENCH_ARGS:-} > "$REPO/gpurun_out/prof_sq.log" 2>&1
echo "sq exit $?" >> "$REPO/gpurun_out/prof_sq.log"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d "$REPO/gpurun_out/prof_sq2" -o sq2 -- python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-frames 0 ${BENCH_ARGS:-} > "$REPO/gpurun_out/prof_sq2.log" 2>&1
echo "sq2 exit $?" >> "$REPO/gpurun_out/prof_sq2.log"
cd "$REPO"
python - <<'PY'
import glob, os, sqlite3
kernel = os.environ.get("KERNEL", "flow_fused_kernel")
for d in ("gpurun_out/prof_sq", "gpurun_out/prof_sq2"):
    hits = glob.glob(d + "/**/*.db", recursive=True)
    if not hits:
        print(d, "no db"); continue
    con = sqlite3.connect(hits[0])
    q = f"select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%{kernel}%' group by counter_name"
    for r in con.execute(q):
        print(d, r)
PY
