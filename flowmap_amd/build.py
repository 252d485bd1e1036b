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


TORCH_SRC = CSRC / "fm_torch.cpp"
TORCH_OUT = HERE / "libflowmap_torch.so"


def torch_binding_up_to_date() -> bool:
    if not TORCH_OUT.exists():
        return False
    newest = max(p.stat().st_mtime for p in [TORCH_SRC, HERE.parent / "include" / "flowmap_hip.h"])
    return TORCH_OUT.stat().st_mtime >= newest


def build_torch_binding(force: bool = False, verbose: bool = True) -> Path:
    """libflowmap_torch.so: the at::Tensor / autograd shims over the C ABI (csrc/fm_torch.cpp), registered with
    TORCH_LIBRARY.  Host-only C++ against this torch's headers and HIP's (c10::hip stream / device guards):
    include / library paths come from torch.utils.cpp_extension, the compiler is g++ (no device code, no hipify)."""
    if not force and torch_binding_up_to_date():
        return TORCH_OUT
    import torch
    from torch.utils import cpp_extension as ce

    rocm = ce.ROCM_HOME or "/opt/rocm"
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found; a C++17 compiler is required to build the torch binding")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wall", "-Wno-unused-function", "-Wno-sign-compare"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", str(TORCH_SRC), "-o", str(TORCH_OUT)]
    for lib_dir in ce.library_paths():
        cmd += [f"-L{lib_dir}", f"-Wl,-rpath,{lib_dir}"]
    cmd += ["-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return TORCH_OUT


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    build_torch_binding(force="--force" in sys.argv)
    print(OUT, TORCH_OUT)
