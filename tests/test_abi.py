"""The C-ABI library loads and exports every entry point include/flowmap_hip.h declares;
the ctypes table matches the header; the product path refuses to run without a GPU."""

import ctypes
import re
from pathlib import Path

import pytest
import torch

from flowmap_amd import _lib

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "flowmap_hip.h").read_text()
DECLS = re.findall(r"^int\s+(fm_\w+)\s*\(([^;]*?)\)\s*;", HEADER, flags=re.M | re.S)


def test_header_declares_entry_points():
    assert len(DECLS) >= 30
    assert {n for n, _ in DECLS} == set(_lib.SIGNATURES), "ctypes table and header disagree on the symbol set"


@pytest.mark.parametrize("name,args", DECLS)
def test_signature_arity_matches_header(name, args):
    n_args = len([a for a in args.split(",") if a.strip() and a.strip() != "void"])
    assert n_args == len(_lib.SIGNATURES[name]), f"{name}: header has {n_args} parameters"


def test_library_exports_every_symbol():
    if not _lib.LIB_PATH.exists():
        from flowmap_amd.build import build_library

        build_library()
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [n for n, _ in DECLS if not hasattr(lib, n)]
    assert not missing, missing


def test_host_double_exports_every_symbol():
    from helpers import build_host_sim

    lib = ctypes.CDLL(str(build_host_sim()))
    missing = [n for n, _ in DECLS if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback():
    """CPU tensors must be refused loudly by the product path (no eager fallback)."""
    from flowmap_amd.model import projection as fm

    _lib.set_library_for_testing(None)
    with pytest.raises(RuntimeError, match="no CPU fallback|needs a GPU"):
        fm.get_extrinsics(torch.eye(4).repeat(1, 3, 1, 1))


@pytest.mark.parametrize("name", sorted(_lib.SIGNATURES))
def test_null_arguments_are_rejected_without_touching_the_gpu(name):
    """Every entry point validates its arguments before the first HIP call and reports
    FM_ERR_ARG (1) as a status — nothing is thrown across the boundary, nothing is launched."""
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    fn = getattr(lib, name)
    fn.argtypes = _lib.SIGNATURES[name]
    fn.restype = ctypes.c_int
    zeros = [None if t is ctypes.c_void_p else t(0) for t in _lib.SIGNATURES[name]]
    if name == "fm_abi_version":  # no arguments: reports the interface version the header states
        assert fn() == int(re.search(r"#define FM_ABI_VERSION (\d+)", HEADER).group(1))
        return
    assert fn(*zeros) == 1


def test_track_tile_constant_matches_header():
    from flowmap_amd import _ops

    assert int(re.search(r"#define FM_TRACK_TILE (\d+)", HEADER).group(1)) == _ops.TRACK_TILE


def test_integration_doc_names_every_entry_point():
    """Every entry point the header declares is named in INTEGRATION.md's table (x_fwd/bwd and x(_suffix)
    shorthands included)."""
    import re

    header = (ROOT / "include" / "flowmap_hip.h").read_text()
    names = set(re.findall(r"\bint (fm_\w+)\(", header))
    integration = (ROOT / "INTEGRATION.md").read_text()
    mentioned = integration
    for base in re.findall(r"`(fm_\w+)_fwd/bwd`", integration):
        mentioned += f" {base}_fwd {base}_bwd"
    for base in re.findall(r"`(fm_\w+)/bwd`", integration):
        mentioned += f" {base} {base.rsplit('_', 1)[0]}_bwd"
    for base, suffix in re.findall(r"`(fm_\w+)\((_\w+)\)`", integration):
        mentioned += f" {base} {base}{suffix}"
    missing = sorted(n for n in names if n not in mentioned)
    assert not missing, missing
