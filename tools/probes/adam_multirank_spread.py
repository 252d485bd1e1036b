import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(os.environ.get("GRAFT_REPO_ROOT", "."))
SMALL = ["--frames", "9", "--height", "48", "--width", "64", "--points", "120", "--cpu-frames", "0", "--steps", "3", "--warmup", "1", "--sustained-steps", "0", "--ate", "off", "--default-resolution", "off"]
def bench(argv):
    env = dict(os.environ, OMP_NUM_THREADS="2", FLOWMAP_BENCH_NO_PROFILER="1")
    done = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    return json.loads([l for l in done.stdout.splitlines() if l.startswith("{")][0])["config"]["loss"]
for i in range(8):
    a = bench(["--config", "c1", "--optimizer", "fused", *SMALL])
    b = bench(["--gpus", "2", "--backend", "gloo", "--one-gpu", "--config", "c1", "--optimizer", "fused", "--halo", "oneshot", "--graph", "off", *SMALL])
    print(i, a, b, abs(a - b) / abs(a), flush=True)
