#!/bin/bash
# Run on the GPU box through gpurun: parity tests, smoke, bench, rocprofv3 kernel stats.
# Everything lands in gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=${STEPS:-10}
{
  echo "== env"; rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" ; nproc; free -g | head -2
  python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))"
} > gpurun_out/env.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit $?" >> gpurun_out/smoke.log
fi
timeout 600 python bench.py --steps $STEPS --warmup 3 ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o stats -- \
     python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-frames 0 ${BENCH_ARGS:-}) > gpurun_out/prof.log 2>&1
  echo "prof exit $?" >> gpurun_out/prof.log
fi
tail -5 gpurun_out/pytest_gpu.log 2>/dev/null; cat gpurun_out/smoke.log 2>/dev/null | tail -3; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
find gpurun_out/prof -name "*stats*" | head
