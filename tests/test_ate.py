"""'final ATE vs ref' (BASELINE.json metric): the same Adam schedule through the oracle and
through flowmap_amd (host double here) ends at the same trajectory error."""

import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_ate_matches_reference_path_small():
    out = subprocess.run(
        [sys.executable, str(ROOT / "tests" / "tools" / "ate_check.py"), "--device", "cpu", "--frames", "6", "--height", "24", "--width", "32",
         "--steps", "40", "--points", "200", "--threads", "4"],
        check=True, capture_output=True, text=True,
    ).stdout.strip().splitlines()[-1]
    r = json.loads(out)
    assert r["ate_reference_path_cpu"] < 0.7 * r["ate_initial"]  # the optimisation does something
    assert abs(r["ate_reference_path_cpu"] - r["ate_flowmap_amd"]) < 1e-4 * max(r["ate_reference_path_cpu"], 1e-3) + 1e-6
    assert abs(r["final_loss_reference_path"] - r["final_loss_flowmap_amd"]) < 1e-3 * abs(r["final_loss_reference_path"])


def test_ate_matches_reference_path_with_tracking():
    out = subprocess.run(
        [sys.executable, str(ROOT / "tests" / "tools" / "ate_check.py"), "--device", "cpu", "--frames", "6", "--height", "24", "--width", "32",
         "--steps", "30", "--points", "200", "--threads", "4", "--tracking", "--track-grid", "5"],
        check=True, capture_output=True, text=True,
    ).stdout.strip().splitlines()[-1]
    r = json.loads(out)
    assert "tracking" in r["losses"]
    assert abs(r["ate_reference_path_cpu"] - r["ate_flowmap_amd"]) < 1e-4 * max(r["ate_reference_path_cpu"], 1e-3) + 1e-6
    assert abs(r["final_loss_reference_path"] - r["final_loss_flowmap_amd"]) < 1e-3 * abs(r["final_loss_reference_path"])
