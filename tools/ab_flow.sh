#!/bin/bash
# A/B of fm_flow.hip build flags on the GPU box: the C1 step and its flow kernel with the shipped build, then a rebuild per entry of
# VARIANTS (comma-separated flags per entry), for the plain step and for the in-pass Adam step.  Nothing is written back.
#   VARIANTS="-DFM_FLOW_SCALAR_TERMS -DFM_FLOW_SCALAR_TERMS,-DFM_FLOW_WAVES=4" bash tools/ab_flow.sh
set -e
cd $GRAFT_REPO_ROOT
line() { for args in "" "--optimizer in_pass"; do python bench.py --cpu-frames 0 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '[$args]', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; done; }
line shipped; line shipped
for v in ${VARIANTS:-}; do
python - <<PY
import flowmap_amd.build as b
b.FILE_FLAGS["fm_flow.hip"] = "$v".split(",")
b.build_library(force=True, verbose=False)
PY
line "$v"; line "$v"
done
