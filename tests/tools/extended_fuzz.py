"""Extended randomised parity sweep (same generator as tests/fuzz_cases.py, other seeds, more
and larger cases).  `python tests/tools/extended_fuzz.py --device cuda:0 --count 150 --seed 11`;
`--device cpu` runs on the host test double."""
import argparse
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import fuzz_cases  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--device", default="cuda:0")
ap.add_argument("--count", type=int, default=150)
ap.add_argument("--seed", type=int, default=11)
args = ap.parse_args()
if args.device == "cpu":
    from flowmap_amd import _lib
    from helpers import build_host_sim

    _lib.set_library_for_testing(build_host_sim())
cases = fuzz_cases.configs(args.seed, args.count)
rng = random.Random(args.seed + 1)
for k in range(args.count // 5):  # some bigger frames (vector paths, several blocks per frame)
    i = args.count + k
    cases.append((i, rng.randint(3, 7), rng.choice([48, 64, 90]), rng.choice([64, 96, 122, 160]), rng.choice([None, 200, 1000]),
                  rng.choice(["huber", "l1", "l2"]), rng.random() < 0.5, rng.random() < 0.7))
failed = []
for cfg in cases:
    try:
        fuzz_cases.run_case(cfg, args.device)
    except AssertionError as exc:
        failed.append((cfg, str(exc)[:200]))
print(f"{len(cases) - len(failed)} / {len(cases)} cases passed")
for cfg, msg in failed:
    print("FAILED", cfg, msg)
sys.exit(1 if failed else 0)
