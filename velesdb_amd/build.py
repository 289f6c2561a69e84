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


# The PROBE build: the same sources with -DVDB_PROBE_SWITCHES, i.e. with the environment switches of csrc/vdb_probe_env.hpp alive
# (kernel / schedule A-B probes, the loop-back collective transport of the tests).  Only the files that hold a switch are compiled a
# second time; the others are shared with the product.  The package never loads it: tests/conftest.py and tools/probes do.
PROBE_SO = os.path.join(LIBDIR, "libvelesdb_hip_probe.so")
PROBE_OBJDIR = os.path.join(HERE, "lib", "obj_probe")
PROBE_FLAG = "-DVDB_PROBE_SWITCHES"


def _has_switches(src: str) -> bool:
    with open(os.path.join(CSRC, src)) as f:
        return "probe_env(" in f.read()


def _compile(src: str, force: bool, probe: bool = False) -> str:
    obj = os.path.join(PROBE_OBJDIR if probe else OBJDIR, src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    inc = os.path.join(HERE, "..", "include", "velesdb_hip.h")
    newest = max(os.path.getmtime(srcp), _deps_mtime(), os.path.getmtime(inc))
    if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
        subprocess.check_call([HIPCC, *FLAGS, *([PROBE_FLAG] if probe else []), "-c", srcp, "-o", obj])
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(PROBE_OBJDIR, exist_ok=True)
    srcs = sources()
    jobs = [(s, False) for s in srcs] + [(s, True) for s in srcs if _has_switches(s)]
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(jobs))) as ex:
        done = list(ex.map(lambda j: _compile(j[0], force, j[1]), jobs))
    objs = done[:len(srcs)]
    probe_objs = {os.path.basename(o): o for o in done[len(srcs):]}
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(o) for o in objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs])
    pobjs = [probe_objs.get(os.path.basename(o), o) for o in objs]
    if force or not os.path.exists(PROBE_SO) or os.path.getmtime(PROBE_SO) < max(os.path.getmtime(o) for o in pobjs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE_SO, *pobjs])
    if verbose:
        print(f"built {SO} and {PROBE_SO}")
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
