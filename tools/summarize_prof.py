#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the short summaries committed under profiles/.

    tools/summarize_prof.py stats <kernel_stats.csv> <out.csv>      # --kernel-trace --stats pass
    tools/summarize_prof.py pmc   <counter_collection.csv> <out.csv> # --pmc pass (one counter per run)

Kernel names are cut to 100 characters (torch's RNG kernels have multi-KB names).  For FETCH_SIZE the
summary applies the gfx950 correction from MI355X_MICROARCH.md (HBM section): the counter is in KiB and
reports half of the bytes of a wide coalesced streaming read, so bytes = value * 1024 * 2.
"""
import collections
import csv
import sys


def short(n):
    return n if len(n) <= 100 else n[:97] + "..."


def stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])


def pmc(src, dst):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(src)):
        key = (short(r["Kernel_Name"]), r["Counter_Name"])
        e = acc.setdefault(key, {"n": 0, "sum": 0.0, "min": None, "max": None, "vgpr": r["VGPR_Count"],
                                 "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"], "scratch": r["Scratch_Size"],
                                 "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]})
        v = float(r["Counter_Value"])
        e["n"] += 1
        e["sum"] += v
        e["min"] = v if e["min"] is None else min(e["min"], v)
        e["max"] = v if e["max"] is None else max(e["max"], v)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "AvgValue", "Min", "Max", "CorrectedBytesPerDispatch",
                    "VGPR", "SGPR", "LDS", "Scratch", "Grid", "Workgroup"])
        for (k, c), e in acc.items():
            avg = e["sum"] / e["n"]
            corr = ""
            if c == "FETCH_SIZE":
                corr = "%.0f" % (avg * 1024 * 2)
            elif c == "WRITE_SIZE":
                corr = "%.0f (uncalibrated, KiB*1024)" % (avg * 1024)
            w.writerow([k, c, e["n"], "%.4f" % avg, e["min"], e["max"], corr, e["vgpr"], e["sgpr"], e["lds"],
                        e["scratch"], e["grid"], e["wg"]])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
