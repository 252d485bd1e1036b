#!/bin/bash
# GPU parity suite + smoke + the preprocessing bench, in one gpurun call.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 300 python tests/tools/preprocess_bench.py > gpurun_out/preprocess_bench.jsonl 2> gpurun_out/preprocess_bench.err
tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cat gpurun_out/preprocess_bench.jsonl; tail -3 gpurun_out/preprocess_bench.err
