"""Extended randomised parity sweep (same generator as tests/fuzz_cases.py, other seeds, more
and larger cases).  `python tests/tools/extended_fuzz.py --device cuda:0 --count 150 --seed 11`;
`--device cpu` runs on the host test double.  `--steps 3 --tracks`: every case with a tracking loss, a multiple-of-4 width and the step
repeated three times, so that the tap exchange between the flow pass and the tracking loss (round 4) is what is compared."""
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
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--tracks", action="store_true")
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
if args.tracks:  # (tracking loss on, three frames at least, lazy surfaces — the fused path — and a width the tap plan takes)
    cases = [(i, max(f, 3), h, w + (-w) % 4, p, kind, True, True) for (i, f, h, w, p, kind, _t, _l) in cases]
engaged = 0
failed = []
for cfg in cases:
    try:
        if args.steps > 1:
            from flowmap_amd import _ops

            before = _ops.counters["flow_tap_absorbs"]
        fuzz_cases.run_case(cfg, args.device, steps=args.steps)
        if args.steps > 1:
            engaged += int(_ops.counters["flow_tap_absorbs"] > before)
    except AssertionError as exc:
        failed.append((cfg, str(exc)[:200]))
print(f"{len(cases) - len(failed)} / {len(cases)} cases passed" + (f"; the flow pass absorbed the tracking loss's tap gradients in {engaged} of them" if args.steps > 1 else ""))
for cfg, msg in failed:
    print("FAILED", cfg, msg)
sys.exit(1 if failed else 0)
