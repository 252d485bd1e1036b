#!/bin/bash
# round 4, eleventh GPU call: track_pairs with one point per lane (twice the waves: 4 per SIMD) now that its prologue is no longer bound by cold lines
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD
out=gpurun_out/r04k; mkdir -p $out
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
clk pg1 -DFM_TRACK_PG=1
clk pg1_ahead1 -DFM_TRACK_PG=1 -DFM_TRACK_AHEAD=1
clk base
