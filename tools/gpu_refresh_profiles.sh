#!/bin/bash
# Regenerate every artefact kept under profiles/ for one round (run through gpurun; results land in
# gpurun_out/profiles_new/, copy them into profiles/ afterwards).   ROUND=r02 bash tools/gpu_refresh_profiles.sh [part...]
# parts: bench variants stats pmc ate misc proxy     (default: all)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
R=${ROUND:-r03}
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/profiles_new
mkdir -p "$OUT"
parts=${*:-bench variants stats pmc ate misc proxy}
stats() {  # stats <name> <bench args...>: rocprofv3 kernel trace of a short bench run -> per-kernel table
  local name=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_$name" -o stats -- python "$REPO/bench.py" --steps 20 --warmup 3 --cpu-frames 0 "$@" > /dev/null 2> "$OUT/prof_$name.err")
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --cpu-frames 0 $*   (calls = 3 set-up + 3 warm-up + 20 timed steps)"; python tools/export_profile.py "gpurun_out/prof_$name"; } > "$OUT/${R}_${name}_rocprofv3_summary.csv" 2>> "$OUT/prof_$name.err"
  rm -rf "gpurun_out/prof_$name"
}
for part in $parts; do
case $part in
bench)
  python bench.py > "$OUT/${R}_bench_c1.json" 2> "$OUT/bench_c1.err"   # (with the CPU baseline on the whole workload: ~70 s of host time)
  python bench.py --config c2 > "$OUT/${R}_bench_c2_tracking.json" 2> "$OUT/bench_c2.err" ;;
variants)
  { for a in "--config c3" "--config c4" "--inputs iid" "--intrinsics softmin" "--optimizer fused" "--optimizer in_pass" "--optimizer torch" \
             "--config c2 --optimizer fused" "--config c2 --optimizer in_pass" "--intrinsics softmin --optimizer in_pass" "--points 0" \
             "--config c2 --intrinsics softmin --optimizer fused" \
             "--height 180 --width 240" "--height 180 --width 240 --optimizer fused" "--height 180 --width 240 --optimizer fused --graph" \
             "--height 180 --width 240 --intrinsics softmin" "--height 180 --width 240 --tracking --optimizer fused" \
             "--height 180 --width 240 --tracking --optimizer fused --graph"; do
      python bench.py --cpu-frames 0 $a 2>> "$OUT/variants.err"; done; } > "$OUT/${R}_bench_variants.jsonl"
  ;;
proxy)
  bash tools/scaling_proxy.sh "$OUT/${R}_strong_scaling_proxy.jsonl" > "$OUT/${R}_strong_scaling_proxy_table.txt" 2>&1; cat "$OUT/${R}_strong_scaling_proxy_table.txt" ;;
stats)
  stats c1_bench
  stats c2_tracking --config c2
  stats dense_procrustes --points 0
  stats c1_adam_in_pass --optimizer in_pass
  stats default_resolution_180x240_tracking_adam --height 180 --width 240 --tracking --optimizer fused
  stats softmin_sweep --intrinsics softmin
  stats share8 --share 8 ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    # (i.i.d. inputs and a kernel filter: with the scene synthesis' 126 000 torch launches in the counter pass rocprofv3 crashed)
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "flow_fused_kernel" -d "$REPO/gpurun_out/prof_$c" -o pmc -- python "$REPO/bench.py" --steps 20 --warmup 3 --cpu-frames 0 --inputs iid > /dev/null 2> "$OUT/prof_$c.err")
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "flow_fused_kernel" -d "$REPO/gpurun_out/prof_adam_$c" -o pmc -- python "$REPO/bench.py" --steps 20 --warmup 3 --cpu-frames 0 --inputs iid --optimizer in_pass > /dev/null 2> "$OUT/prof_adam_$c.err")
  done
  python - "$R" <<'PY' > "$OUT/${R}_flow_kernel_traffic.json"
import glob, json, sqlite3, sys
def avg(d, counter, like):
    con = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0])
    return con.execute("select count(*), avg(value) from counters_collection where kernel_name like ? and counter_name = ?", (like, counter)).fetchone()
f, h, w = 150, 720, 1280
out = {"round": int(sys.argv[1][1:]), "workload": {"frames": f, "height": h, "width": w, "inputs": "iid"},
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) --kernel-include-regex flow_fused_kernel, python bench.py --steps 20 --warmup 3 --cpu-frames 0 --inputs iid [--optimizer in_pass]",
       "fetch_correction": "x2: on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads (MI355X_MICROARCH.md HBM section); confirmed in round 1 on torch's sigmoid kernel (reads 549.2 MB, FETCH_SIZE 268220 KB) and on fm::sum2_kernel (reads 1098.4 MB, reports 549.3 MB)",
       "write_correction": "x1: confirmed on torch's sigmoid kernel (writes 549.2 MB, WRITE_SIZE 536400 KB)"}
for key, prefix, like, per_px in (("flow_fused_kernel", "gpurun_out/prof_", "%flow_fused_kernel<4, 0, true, true, false>%", (8, 24)),
                                   ("flow_fused_kernel_adam", "gpurun_out/prof_adam_", "%flow_fused_kernel<4, 0, true, true, true>%", (24, 24))):
    nf, fetch = avg(prefix + "FETCH_SIZE", "FETCH_SIZE", like)
    nw, write = avg(prefix + "WRITE_SIZE", "WRITE_SIZE", like)
    algo = h * w * (per_px[0] * f + per_px[1] * (f - 1))
    rd, wr = fetch * 1024 * 2, write * 1024
    entry = {"kernel": like.strip("%"), "dispatches": [nf, nw], "fetch_size_kb_raw_avg": fetch, "write_size_kb_raw_avg": write,
             "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
             "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": round((rd + wr) / algo, 4)}
    if key == "flow_fused_kernel":
        out.update(entry)  # (bench.py reads these keys)
    else:
        out[key] = entry
print(json.dumps(out, indent=2))
PY
  rm -rf gpurun_out/prof_FETCH_SIZE gpurun_out/prof_WRITE_SIZE gpurun_out/prof_adam_FETCH_SIZE gpurun_out/prof_adam_WRITE_SIZE ;;
ate)
  python tests/tools/ate_check.py --device cuda --in-pass 2> "$OUT/ate.err" | tail -1 > "$OUT/${R}_ate_c0_16x256x256.json"
  python tests/tools/ate_check.py --device cuda --height 192 --width 256 --tracking --in-pass 2>> "$OUT/ate.err" | tail -1 > "$OUT/${R}_ate_16x192x256_flow_tracking.json"
  python tests/tools/ate_check.py --device cuda --frames 32 --height 360 --width 640 --tracking --in-pass 2>> "$OUT/ate.err" | tail -1 > "$OUT/${R}_ate_32x360x640_flow_tracking.json"
  for v in "" "_nosoftmin" "_nosoftmin_lr3e-4"; do
    [ -f tests/golden/ate_150x360x640${v}_reference.json ] && python tests/tools/ate_full_chain.py --leg ours --reference tests/golden/ate_150x360x640${v}_reference.json 2>> "$OUT/ate.err" | tail -1 > "$OUT/${R}_ate_150x360x640${v}.json"
  done
  [ -f tests/golden/ate_150x360x640_reference_perturbed.json ] && python tests/tools/ate_full_chain.py --leg compare --reference tests/golden/ate_150x360x640_reference.json --other tests/golden/ate_150x360x640_reference_perturbed.json > "$OUT/${R}_ate_150x360x640_reference_sensitivity.json" ;;
misc)
  python tools/dense_microbench.py > "$OUT/${R}_dense_microbench.txt" 2>&1 ;;
esac
done
ls -la "$OUT"; cat "$OUT/${R}_bench_c1.json" 2>/dev/null
