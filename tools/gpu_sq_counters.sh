#!/bin/bash
# VALU / SALU / LDS instruction counts and the launch's cycles for every fm:: kernel of a bench workload (a PMC pass of its own):
#   BENCH_ARGS="--tracking" bash tools/gpu_sq_counters.sh
# valu_share = SQ_INSTS_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the fraction of the launch's VALU issue slots in use.
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_sq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES -d /tmp/prof_sq -o sq -- python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-frames 0 --ate off --sustained-steps 0 ${BENCH_ARGS:-} > /tmp/prof_sq.log 2>&1
cd "$REPO"
python - <<'PY'
import glob, sqlite3
hits = glob.glob("/tmp/prof_sq/**/*.db", recursive=True)
con = sqlite3.connect(hits[0])
rows = {}
for name, counter, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%fm::%' group by kernel_name, counter_name"):
    rows.setdefault(name, {})[counter] = avg
    rows[name]["launches"] = n
print("kernel,launches,waves,valu_per_wave,salu_per_wave,lds_per_wave,cycles,valu_share")
for name, r in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    cyc = r.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc < 5000:
        continue
    w = max(r.get("SQ_WAVES", 1), 1)
    print(f'"{name[:100]}",{r["launches"]},{w:.0f},{r.get("SQ_INSTS_VALU",0)/w:.0f},{r.get("SQ_INSTS_SALU",0)/w:.0f},{r.get("SQ_INSTS_LDS",0)/w:.0f},{cyc:.0f},{r.get("SQ_INSTS_VALU",0)*4/(cyc*1024):.2f}')
PY
