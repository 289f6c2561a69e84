#!/usr/bin/env python3
"""Turns the FETCH_SIZE pass of `bench.py --no-tiles` (rocprofv3 --pmc FETCH_SIZE --kernel-trace, see
tools/gpu_round_check.sh) into profiles/pmc_traffic.json, which bench.py reads to fill roofline.traffic.

    tools/pmc_traffic.py <counter_collection.csv> <bench_line_of_that_pass.json> <out.json> <source note>

FETCH_SIZE is in KiB and counts half of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM
section): bytes = value * 1024 * 2.  Keyed by "<bench kernel name>@<queries per step>"."""
import collections
import csv
import json
import re
import sys

src, line_path, dst, note = sys.argv[1:5]
line = json.loads(open(line_path).read().strip().splitlines()[-1])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024 * 2)
out = {"source": note, "kernels": {}}


def match(bench_name, prof_name):
    # bench: sweep_topk_gemm_f32<cosine,NQF=4> / hnsw_search_kernel<cosine,CPL=3>; profiler: vdb::sweep_topk_gemm_f32<0, 4, ...>
    base = bench_name.split("<")[0]
    nums = re.findall(r"=(\d+)", bench_name)
    if ("vdb::" + base + "<") not in prof_name:
        return False
    pn = re.findall(r"\d+", prof_name.split(base + "<")[1].split(">")[0])
    return all(n in pn[1:] for n in nums)


for key, kname in ((line["roofline"]["kernel"] + "@%d" % line["config"]["queries_per_step"], line["roofline"]["kernel"]),
                   (line["hnsw"]["roofline"]["kernel"] + "@hnsw", line["hnsw"]["roofline"]["kernel"]) if line.get("hnsw") else (None, None)):
    if not key:
        continue
    for pname, vals in acc.items():
        # the traversal leg's launch is the register-list instantiation (NS = 4); ef > 192 (the ef curve) runs the LDS-list one
        if kname.startswith("hnsw_search_kernel") and ", 4, false>" not in pname:
            continue
        if match(kname, pname):
            out["kernels"][key] = {"fetch_bytes_per_launch": round(sum(vals) / len(vals)), "dispatches": len(vals),
                                   "profiler_kernel": pname[:120]}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
