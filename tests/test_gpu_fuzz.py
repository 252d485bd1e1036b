"""Randomised parity sweep of the full step on the GPU (same generator as the CPU sweep,
different seed, more cases)."""

import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", fuzz_cases.configs(seed=2, count=24), ids=lambda c: f"{c[1]}x{c[2]}x{c[3]}-P{c[4]}-{c[5]}-t{int(c[6])}-l{int(c[7])}")
def test_random_step_gpu(cfg):
    fuzz_cases.run_case(cfg, "cuda:0")
