"""Randomised parity sweep of the full step (random F/H/W incl. odd sizes, P, mapping kind,
tracks on/off, lazy on/off) against the fp64 oracle — CPU, host double."""

import pytest

import fuzz_cases
from flowmap_amd import _lib
from helpers import build_host_sim


@pytest.fixture(autouse=True, scope="module")
def host_double():
    _lib.set_library_for_testing(build_host_sim())
    yield
    _lib.set_library_for_testing(None)


@pytest.mark.parametrize("cfg", fuzz_cases.configs(seed=1, count=12), ids=lambda c: f"{c[1]}x{c[2]}x{c[3]}-P{c[4]}-{c[5]}-t{int(c[6])}-l{int(c[7])}")
def test_random_step(cfg):
    fuzz_cases.run_case(cfg, "cpu")
