"""bench.py's strong-scaling glue (shard the video, tracking windows across shard borders, packed all-reduce, halo
exchange, the one JSON line) run end to end with 1, 2 and 3 ranks over gloo on the host test double: every world size
must report the loss of the whole video.  The GPU runs of the same file are the driver's; nothing is measured here."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
ARGS = ["--frames", "9", "--height", "24", "--width", "32", "--points", "60", "--steps", "2", "--warmup", "1", "--cpu-frames", "0"]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world: int, extra):
    launcher = str(ROOT / "tests" / "tools" / "bench_dryrun.py")
    if world == 1:
        cmd = [sys.executable, launcher, "--gpus", "1", *ARGS, *extra]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), launcher, "--gpus", str(world), *ARGS, *extra]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    if world == 1:  # (--share K makes a one-rank process group on bench.py's default port: its own port, so that these tests can run side by side)
        env["MASTER_PORT"] = str(_free_port())
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, done.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [["--config", "c1"], ["--config", "c2"], ["--config", "c1", "--intrinsics", "softmin"]], ids=["flow", "flow+tracking", "softmin"])
def test_strong_scaling_reports_the_whole_video(extra):
    single = _run(1, extra)
    assert single["n_gpus"] == 1 and single["config"]["frames_per_gpu"] == 9
    # the default line is measured on the drop-in path: the stand-in package's Model / get_losses after install(); the hand-built modules agree
    via = single["via_install"]
    assert via["package"].startswith("stand-in") and via["modules"]["backbone"] == "flowmap_amd.model.backbone.BackboneExplicitDepth"
    assert via["modules"]["model"] == "flowmap.model.model.Model" and via["ms_per_step"] == single["ms_per_step"]
    if "softmin" not in extra:  # (the sweep draws fresh random pixels every step: two runs do not end on the same loss)
        assert abs(via["direct"]["loss"] - single["config"]["loss"]) <= 1e-6 * abs(single["config"]["loss"])
    assert "via_install" not in _run(1, [*extra, "--model", "direct"])
    for world in (2, 3):
        line = _run(world, extra)
        assert line["n_gpus"] == world and line["scaling"] == "strong" and line["steps"] == 2
        assert line["config"]["video_frames"] == 9 and line["config"]["frames_per_gpu"] < 9
        assert abs(line["config"]["loss"] - single["config"]["loss"]) <= 2e-5 * abs(single["config"]["loss"]), (world, line["config"]["loss"], single["config"]["loss"])


@pytest.mark.parametrize("halo", ["oneshot", "early", "ghost"])
def test_every_halo_mode_reports_the_same_loss(halo):
    """bench.py --halo oneshot | early | ghost over three ranks (ghost — the default of a multi-rank strong-scaling run — hands every rank
    the neighbouring pairs' flows when the video is cut): the mode is named in the line, the loss is the whole video's."""
    single = _run(1, ["--config", "c1"])
    line = _run(3, ["--config", "c1", "--halo", halo, "--steps", "3"])
    assert str(line["config"]["halo_exchange"]).startswith({"oneshot": "one shot", "early": "early", "ghost": "ghost"}[halo])
    assert abs(line["config"]["loss"] - single["config"]["loss"]) <= 2e-5 * abs(single["config"]["loss"])
    if halo == "ghost":
        proxy = _run(1, ["--config", "c1", "--share", "3", "--halo", "ghost"])
        assert str(proxy["config"]["halo_exchange"]).startswith("ghost") and proxy["proxy"]["share_of"] == 3


def test_weak_scaling_runs_one_video_per_rank():
    line = _run(2, ["--config", "c1", "--scaling", "weak"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frames_per_gpu"] == 9


@pytest.mark.parametrize("extra", [[], ["--optimizer", "in_pass"], ["--optimizer", "fused"]], ids=["fwd+bwd", "in-pass adam", "fused adam"])
def test_share_proxy_runs_one_ranks_share_with_every_collective(extra):
    """bench.py --share K: rank R's share of a K-rank strong-scaling run in ONE process (a one-member process group, the halo
    exchange replaced by its local copies/adds) — the line names the share and is flagged as a proxy."""
    line = _run(1, ["--config", "c1", "--share", "3", *extra])
    assert line["n_gpus"] == 1 and line["proxy"]["share_of"] == 3 and line["proxy"]["rank"] == 1 and line["proxy"]["wire_time_included"] is False
    a, b = line["proxy"]["pairs"]
    assert (a, b) == (3, 6) and line["config"]["frames_per_gpu"] == 4 and line["config"]["video_frames"] == 9  # 8 pairs over 3 ranks: 3 + 3 + 2
    assert "PROXY" in line["config"]["workload"] and "cpu_baseline" not in line
    edge = _run(1, ["--config", "c1", "--share", "3", "--share-rank", "2", *extra])
    assert edge["proxy"]["pairs"] == [6, 8] and edge["config"]["frames_per_gpu"] == 3


def test_whole_c4_path_releases_the_originals_and_keeps_a_cpu_sample():
    """bench.py --config c4 --whole (all of configs[4] on one GPU, here shrunk): the flows are packed, the originals released — the CPU leg's
    sample copied first — and the line says so; the loss equals the run that keeps the originals."""
    small = ["--frames", "9", "--height", "24", "--width", "32", "--points", "60", "--steps", "2", "--warmup", "1", "--cpu-frames", "4", "--cpu-iters", "1"]
    launcher = str(ROOT / "tests" / "tools" / "bench_dryrun.py")
    lines = []
    for extra in (["--whole"], []):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env["OMP_NUM_THREADS"] = "2"
        done = subprocess.run([sys.executable, launcher, "--gpus", "1", "--config", "c4", *small, *extra], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert done.returncode == 0, done.stderr[-3000:]
        if extra:
            assert "released" in done.stderr
        lines.append(json.loads([ln for ln in done.stdout.splitlines() if ln.startswith("{")][0]))
    whole, kept = lines
    assert "WHOLE" in whole["config"]["workload"] and whole["config"]["video_frames"] == 9
    assert whole["cpu_baseline"]["frames"] == 4 and whole["cpu_baseline"]["whole_workload"] is False
    # the port's time restated for the reference's own code (tests/golden/cpu_calibration.json: imported reference vs port, interleaved)
    cal = whole["cpu_baseline"]["port_over_reference"]
    assert 0.5 < cal["port_over_reference"] < 1.5 and cal["measured_at"] == [150, 720, 1280]
    assert abs(whole["cpu_baseline"]["reference_equivalent_value"] - whole["cpu_baseline"]["value"] * cal["port_over_reference"]) < 1e-9
    assert abs(whole["config"]["loss"] - kept["config"]["loss"]) <= 1e-6 * abs(kept["config"]["loss"])
    assert abs(whole["cpu_baseline"]["loss"] - kept["cpu_baseline"]["loss"]) <= 1e-6 * abs(kept["cpu_baseline"]["loss"])


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_training_step_mode_drives_the_packages_wrapper(mode):
    """bench.py --training-step eager | graph: the step is the reference-layout package's ModelWrapperOverfit.training_step in a trainer's
    order; on the host double install(graph=True) captures nothing and the line says so.  Same loss as the default installed step."""
    plain = _run(1, ["--config", "c2"])
    line = _run(1, ["--config", "c2", "--training-step", mode])
    info = line["via_install"]["training_step"]
    assert info["mode"] == mode and "training_step" in line["config"]["workload"]
    if mode == "graph":
        assert info["captures"] == 0 and info["replays"] == 0 and info["disabled"] is None
    assert abs(line["config"]["loss"] - plain["config"]["loss"]) <= 1e-6 * abs(plain["config"]["loss"])
    assert plain["via_install"]["training_step"] is None


def _literal(argv, expect_ok=True, extra_env=None):
    """`python bench.py <argv>` — the driver's command shape, no launcher around it.  The dry run's two environment variables say which device the
    ranks compute on (cpu: gloo) and which script they run (the launcher that injects the host test double before bench.py's main())."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2", FLOWMAP_BENCH_DEVICE="cpu", FLOWMAP_BENCH_LAUNCHER=str(ROOT / "tests" / "tools" / "bench_dryrun.py"), **(extra_env or {}))
    done = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if not expect_ok:
        return done
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, done.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


SMALL = ["--frames", "9", "--height", "24", "--width", "32", "--points", "60", "--cpu-frames", "0"]


def test_the_literal_command_launches_n_ranks():
    """`python bench.py --gpus 2 --steps 2 --warmup 1` (no torch.distributed.run in front) is a TWO-rank run: bench.py becomes
    the launcher (overfit.py:94-108 goes multi-GPU without one too), the ranks meet in a process group of that size, the line says so and
    names the frames each rank holds; the loss is the whole video's."""
    single = _run(1, ["--config", "c1"])
    line = _literal(["--gpus", "2", "--steps", "2", "--warmup", "1", *SMALL])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo" and line["steps"] == 2 and line["warmup"] == 1
    assert line["frame_split"] == [[0, 4], [4, 8]] and line["config"]["frames_per_gpu"] == 5 and line["scaling"] == "strong"
    assert abs(line["config"]["loss"] - single["config"]["loss"]) <= 2e-5 * abs(single["config"]["loss"])
    assert single["rccl_ranks"] == 1 and single["frame_split"] == [[0, 8]]
    four = _literal(["--gpus", "4", "--steps", "2", "--warmup", "1", *SMALL])
    assert four["n_gpus"] == 4 and four["rccl_ranks"] == 4 and four["frame_split"] == [[0, 2], [2, 4], [4, 6], [6, 8]]
    assert abs(four["config"]["loss"] - single["config"]["loss"]) <= 2e-5 * abs(single["config"]["loss"])


def test_a_rank_count_that_disagrees_with_gpus_is_refused():
    """A launcher that started another number of ranks than --gpus names must not produce a line (it would enter a scaling record under the wrong N)."""
    done = _literal(["--gpus", "2", "--steps", "1", "--warmup", "0", *SMALL], expect_ok=False, extra_env={"WORLD_SIZE": "1", "RANK": "0"})
    assert done.returncode != 0 and "disagree" in done.stderr and not [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
