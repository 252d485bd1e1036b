#!/bin/bash
# Regenerate every artefact kept under profiles/ (run through gpurun; results land in
# gpurun_out/profiles_new/, copy them into profiles/ afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/profiles_new
mkdir -p "$OUT"
python bench.py --steps 30 --warmup 5 > "$OUT/r01_bench_c1.json" 2> "$OUT/bench_c1.err"
python bench.py --steps 30 --warmup 5 --cpu-frames 0 --tracking > "$OUT/r01_bench_c2_tracking.json" 2>/dev/null
{ for a in "--intrinsics softmin" "--optimizer fused" "--optimizer torch" "--points 0" "--points 0 --smooth-flows" "--tracking --intrinsics softmin --optimizer fused"; do
    python bench.py --steps 20 --warmup 5 --cpu-frames 0 $a 2>/dev/null; done; } > "$OUT/r01_bench_variants.jsonl"
STEPS=20 bash tools/gpu_profile.sh > /dev/null 2>&1
cp gpurun_out/profile_summary.txt "$OUT/r01_c1_bench_rocprofv3_summary.csv"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_track" -o stats -- python "$REPO/bench.py" --steps 20 --warmup 2 --cpu-frames 0 --tracking > /dev/null 2>&1)
python tools/export_profile.py gpurun_out/prof_track > "$OUT/r01_c2_tracking_rocprofv3_summary.csv"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_dense" -o stats -- python "$REPO/bench.py" --steps 6 --warmup 2 --cpu-frames 0 --points 0 > /dev/null 2>&1)
python tools/export_profile.py gpurun_out/prof_dense > "$OUT/r01_dense_procrustes_rocprofv3_summary.csv"
python - <<'PY' > "$OUT/r01_flow_kernel_traffic.json"
import glob, json, sqlite3
def avg(d, counter):
    con = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0])
    n, v = con.execute("select count(*), avg(value) from counters_collection where kernel_name like '%flow_fused_kernel%' and counter_name = ?", (counter,)).fetchone()
    return n, v
nf, fetch = avg("gpurun_out/prof_fetch", "FETCH_SIZE")
nw, write = avg("gpurun_out/prof_write", "WRITE_SIZE")
f, h, w = 150, 720, 1280
algo = h * w * (8 * f + 24 * (f - 1))
rd, wr = fetch * 1024 * 2, write * 1024
print(json.dumps({
    "round": 1,
    "kernel": "fm::flow_fused_kernel<4, 0, true, true> (affine projection form, packed constants, non-temporal accesses)",
    "workload": {"frames": f, "height": h, "width": w},
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), python bench.py --steps 20 --warmup 2 --cpu-frames 0; {nf} / {nw} dispatches; see r01_c1_bench_rocprofv3_summary.csv",
    "fetch_size_kb_raw_avg": fetch, "write_size_kb_raw_avg": write,
    "fetch_correction": "x2: on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads (MI355X_MICROARCH.md HBM section); confirmed earlier in this round on torch's sigmoid kernel (reads 549.2 MB, FETCH_SIZE 268220 KB) and on fm::sum2_kernel (reads 1098.4 MB, reports 549.3 MB)",
    "write_correction": "x1: confirmed on torch's sigmoid kernel (writes 549.2 MB, WRITE_SIZE 536400 KB)",
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
    "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": round((rd + wr) / algo, 4)}, indent=2))
PY
python tools/probes/bw_probe.py > "$OUT/r01_hbm_stream_probe.json" 2>/dev/null
python tools/adam_microbench.py > "$OUT/r01_adam_microbench.json" 2>/dev/null
python tests/tools/preprocess_bench.py > "$OUT/r01_preprocess_bench.jsonl" 2>/dev/null
python tests/tools/ate_check.py --device cuda > "$OUT/ate.log" 2>&1; tail -1 "$OUT/ate.log" > "$OUT/r01_ate_c0_16x256x256.json"
python tests/tools/ate_check.py --device cuda --height 192 --width 256 --tracking > "$OUT/ate_tracking.log" 2>&1; tail -1 "$OUT/ate_tracking.log" > "$OUT/r01_ate_16x192x256_flow_tracking.json"
rm -rf gpurun_out/prof_track gpurun_out/prof_dense gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_stats
ls -la "$OUT"; cat "$OUT/r01_bench_c1.json"; cat "$OUT/r01_flow_kernel_traffic.json" | tail -8
