#!/bin/bash
# round 4, end-of-round check in one gpurun call: the whole GPU suite + smoke, the driver's command twice (bench lines), its rocprofv3
# kernel table, and the HBM traffic of the flow kernel on that very command (PMC passes: FETCH_SIZE and WRITE_SIZE separately)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04z; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke exit $?" >> $out/smoke.log
for run in 1 2; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$run.json 2> $out/bench_driver_$run.err
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/$out/prof_c1 -o s -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0) > $out/prof_c1.log 2>&1
python3 tools/export_profile.py $out/prof_c1 > $out/c1_bench_rocprofv3_summary.csv 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $REPO/$out/pmc_$ctr -o p -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0) > $out/pmc_$ctr.log 2>&1
done
python3 tools/export_profile.py $out/pmc_FETCH_SIZE $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_summary.txt 2>&1
rm -rf $out/prof_c1 $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
tail -4 $out/pytest_gpu.log; tail -2 $out/smoke.log
python3 - <<'PY'
import json
for run in (1, 2):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r04z/bench_driver_{run}.json") if l.startswith("{")][-1]); r = d["roofline"]
        print(run, "ms/step %.4f value %.1f" % (d["ms_per_step"], d["value"]), "kernel %.4f frac %.3f launches %s step_frac %.3f" % (r["kernel_ms"], r["frac"], r["launches_per_step"], r["step_frac"]),
              "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("sample", "")[:60])
    except Exception as e:
        print(run, "no bench line:", e)
PY
head -12 $out/c1_bench_rocprofv3_summary.csv | cut -c1-160
grep -i "flow_fused\|FETCH\|WRITE" $out/pmc_summary.txt | head -8 | cut -c1-200
