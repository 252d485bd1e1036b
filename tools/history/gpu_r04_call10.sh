#!/bin/bash
# round 4, tenth GPU call: how track_pairs' waves are spread over the SIMDs (HW_ID per wave), and what capping the resident waves per CU with LDS does
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04j; mkdir -p $out
export TMPDIR=/tmp
clk() { name=$1; shift; timeout 600 python3 tools/track_clocks.py "$@" > $out/track_clocks_$name.txt 2> $out/track_clocks_$name.err; echo "== $name"; grep -A14 "waves; median" $out/track_clocks_$name.err; python3 - $out/track_clocks_$name.txt <<'PY'
import json, sys
t = open(sys.argv[1]).read()
try:
    r = json.loads(t[t.index('{"metric"'):].splitlines()[0])
    print("   ms/step", round(r["ms_per_step"], 4), "track call ms", round(r["roofline_tracking"]["kernel_ms"], 4))
except Exception as e:
    print("   no bench line", e)
PY
}
clk base
clk pad20k -DFM_TRACK_LDS_PAD=20480
clk pad40k -DFM_TRACK_LDS_PAD=40960
clk pad14k -DFM_TRACK_LDS_PAD=14336
