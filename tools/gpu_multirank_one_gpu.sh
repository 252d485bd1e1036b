#!/bin/bash
# The driver's N-rank command at the metric's size on a ONE-GPU box, over RCCL: the ranks share cuda:0, every rank declares a host of its own (NCCL_HOSTID) and they meet
# over RCCL's socket transport on the loopback interface (tools/probes/rccl_one_gpu_probe.py; RCCL refuses two ranks of one host on one device).  FUNCTIONAL record: every
# rank runs this package's kernels on its shard of the 150-frame video, the halo travels point to point through RCCL, forward + backward replay as hipGraphs; the loss must
# be the whole video's.  Timing is meaningless (N processes time-slice one GPU, the transport is sockets).  One line over gloo for comparison.
#   gpurun -- 'bash tools/gpu_multirank_one_gpu.sh'
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_multirank; mkdir -p $out
export FLOWMAP_BENCH_NO_PROFILER=1
common="--steps 5 --warmup 2 --cpu-frames 0 --ate off --default-resolution off --sustained-steps 0"
run() {  # name, args
  timeout 900 python bench.py $2 $common > $out/$1.json 2> $out/$1.err
  python - "$1" "$out/$1.json" <<'PY'
import json, sys
name, path = sys.argv[1:3]
try:
    d = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    print(name, "n_gpus", d["n_gpus"], "ranks", d["rccl_ranks"], d["collective_backend"], "frames/rank", d["config"]["frames_per_gpu"], "halo", str(d["config"]["halo_exchange"])[:8],
          "graphs" if "hipGraph" in d["config"]["workload"] else "eager", "loss", repr(d["config"]["loss"]), flush=True)
except Exception as exc:
    print(name, "FAILED", repr(exc)[:200], flush=True)
PY
}
run n1 "--config c1"
for n in 2 4 8; do run n${n}_default "--gpus $n --one-gpu --config c1"; done   # the driver's command shape: ghost halo, compute graphs
run n8_early_graphs "--gpus 8 --one-gpu --config c1 --halo early"
run n4_oneshot_eager "--gpus 4 --one-gpu --config c1 --halo oneshot --graph off"
run n8_default_gloo "--gpus 8 --backend gloo --one-gpu --config c1"
run n1_c2 "--config c2"
run n4_c2_default "--gpus 4 --one-gpu --config c2"
run n8_c2_early "--gpus 8 --one-gpu --config c2 --halo early"
# BASELINE configs[3]: 65 frames @ 1080x1920 frame-sharded over 4 ranks (pairs 0-15 | 16-31 | 32-47 | 48-63)
run n1_c3 "--config c3"
run n4_c3_default "--gpus 4 --one-gpu --config c3"
