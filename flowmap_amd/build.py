"""Build libflowmap_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m flowmap_amd.build [--force]
"""

from __future__ import annotations

import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "libflowmap_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
# Per-file extras (none at the moment; see DESIGN.md §3.1 for the variants that were measured).
FILE_FLAGS: dict = {}

def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build flowmap_amd")
    return exe


def sources():
    return sorted(CSRC.glob("*.hip"))


def up_to_date() -> bool:
    if not OUT.exists():
        return False
    newest = max(p.stat().st_mtime for p in [*CSRC.glob("*.hip"), *CSRC.glob("*.h"), HERE.parent / "include" / "flowmap_hip.h"])
    return OUT.stat().st_mtime >= newest


def build_library(force: bool = False, verbose: bool = True) -> Path:
    if not force and up_to_date():
        return OUT
    objs = []
    build_dir = HERE / "csrc" / "build"
    build_dir.mkdir(exist_ok=True)
    for src in sources():
        obj = build_dir / (src.stem + ".o")
        cmd = [hipcc(), f"--offload-arch={ARCH}", *FLAGS, *FILE_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(str(obj))
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(OUT)
