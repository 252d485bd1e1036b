#!/bin/bash
# round 4, first GPU call: the driver's exact command three times (is a 20-step run on a clock ramp?), configs[4] WHOLE on one GPU
# (test + bench line), the MFMA variants of the dense moments kernel, the reference op sequence on stock PyTorch-ROCm at 150 frames.
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04a; mkdir -p $out
export TMPDIR=/tmp
export FLOWMAP_PARITY_RECORD=$PWD/$out/full_size_parity.jsonl
{ free -g | head -2; nproc; rocm-smi --showmeminfo vram 2>/dev/null | head -6; rocm-smi --showclocks 2>/dev/null | head -12; } > $out/info.txt 2>&1
# 1. the driver's command, as the driver issues it
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_1.json 2> $out/bench_driver_1.err; cut -c1-300 $out/bench_driver_1.json; tail -2 $out/bench_driver_1.err
for i in 2 3; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err; cut -c1-200 $out/bench_driver_$i.json; done
timeout 300 python3 bench.py --gpus 1 --steps 100 --warmup 20 --cpu-frames 0 > $out/bench_100.json 2> $out/bench_100.err; cut -c1-200 $out/bench_100.json
python3 - <<'PY' > $out/ramp.txt
import json
for name in ("bench_driver_1", "bench_driver_2", "bench_driver_3", "bench_100"):
    try:
        r = json.load(open(f"gpurun_out/r04a/{name}.json"))
        k = r["roofline"]["kernel_ms_per_launch"]
        print(name, "ms/step", round(r["ms_per_step"], 4), "kernel avg", round(r["roofline"]["kernel_ms"], 4), "first5/last5", r["roofline"]["kernel_ms_first5_last5"], "per launch", k[:20], "...", k[-5:])
    except Exception as e:
        print(name, "failed", e)
PY
cat $out/ramp.txt
# 2. configs[4] whole
( time timeout 1200 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k c4_whole -rf ) > $out/pytest_c4_whole.log 2>&1; tail -12 $out/pytest_c4_whole.log
timeout 900 python3 bench.py --config c4 --whole --steps 20 --warmup 5 --cpu-iters 1 > $out/bench_c4_whole.json 2> $out/bench_c4_whole.err; cut -c1-400 $out/bench_c4_whole.json; tail -3 $out/bench_c4_whole.err
# 3. the dense moments kernel: VALU against two MFMA formulations, 32x64 and 64x64 tiles
timeout 600 python3 tools/dense_microbench.py 150 gentle > $out/dense_microbench_mfma.txt 2> $out/dense_microbench_mfma.err; cat $out/dense_microbench_mfma.txt | grep -v "^{"; tail -3 $out/dense_microbench_mfma.err
# 4. the reference's op sequence on stock PyTorch-ROCm (round 1: a GPU memory-access fault at 30 and 150 frames)
for fr in 150 32; do timeout 400 python3 tests/tools/torch_gpu_reference_ops.py --frames $fr --iters 3 > $out/torch_rocm_$fr.json 2> $out/torch_rocm_$fr.err; echo "torch-rocm $fr frames: exit $?"; cat $out/torch_rocm_$fr.json; tail -3 $out/torch_rocm_$fr.err; done
