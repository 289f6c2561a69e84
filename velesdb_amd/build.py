"""Build recipe for libvelesdb_hip.so (hand-written HIP for gfx950, no other target).

    python -m velesdb_amd.build          # compile every .hip under csrc/ and link in-tree

hipcc cross-compiles gfx950 without a GPU.  The .so stays in velesdb_amd/lib/ (git-ignored,
but shipped to the GPU box with the repo snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
SO = os.path.join(LIBDIR, "libvelesdb_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: fma only where the source writes fmaf (canonical arithmetic, vdb_device.hpp)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-function"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC)
               if f.endswith((".hpp", ".h", ".inc"))) if os.listdir(CSRC) else 0


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJDIR, src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    inc = os.path.join(HERE, "..", "include", "velesdb_hip.h")
    newest = max(os.path.getmtime(srcp), _deps_mtime(), os.path.getmtime(inc))
    if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
        subprocess.check_call([HIPCC, *FLAGS, "-c", srcp, "-o", obj])
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(o) for o in objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs])
    if verbose:
        print(f"built {SO}")
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
