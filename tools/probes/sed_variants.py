#!/usr/bin/env python3
"""Experiment variants of one csrc/*.hip file: a patched COPY compiled into tools/probes/out/ and linked with the product's other
objects (the product source carries no experiment hooks).  usage: sed_variants.py <file.hip> <name> <marker> <old> <new> [...]
Each (marker, old, new): the first occurrence of `old` AFTER the first occurrence of `marker` is replaced by `new`."""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(R, "tools", "probes", "out")
os.makedirs(O, exist_ok=True)
src, name = sys.argv[1], sys.argv[2]
text = open(os.path.join(R, "velesdb_amd", "csrc", src)).read()
import re
text = re.sub(r'#include "(g16_\w+\.inc)"', lambda m: open(os.path.join(R, "velesdb_amd", "csrc", m.group(1))).read(), text)  # patches reach the includes
args = sys.argv[3:]
for i in range(0, len(args), 3):
    marker, old, new = args[i:i + 3]
    at = text.index(marker)
    j = text.index(old, at)
    text = text[:j] + new + text[j + len(old):]
stem = src[:-4]
cp = os.path.join(O, f"{stem}_{name}.hip")
open(cp, "w").write(text)
FL = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w", "-I" + os.path.join(R, "velesdb_amd", "csrc")]
obj = os.path.join(O, f"{stem}_{name}.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", *FL, "-c", cp, "-o", obj])
objs = [os.path.join(R, "velesdb_amd", "lib", "obj", f) for f in sorted(os.listdir(os.path.join(R, "velesdb_amd", "lib", "obj"))) if f != stem + ".o"]
so = os.path.join(O, f"libvelesdb_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs, obj])
print("built", so)
