"""Launcher for tests/test_bench_dryrun.py: injects the host test double, then runs bench.py's main() on CPU tensors
(FLOWMAP_BENCH_DEVICE=cpu, gloo).  Test infrastructure: validates the multi-rank glue of bench.py, measures nothing."""
import os
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ["FLOWMAP_BENCH_DEVICE"] = "cpu"

from flowmap_amd import _lib  # noqa: E402
from helpers import build_host_sim  # noqa: E402

_lib.set_library_for_testing(build_host_sim())
sys.argv = [str(ROOT / "bench.py"), *sys.argv[1:]]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
