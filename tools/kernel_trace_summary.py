#!/usr/bin/env python3
"""Per kernel and grid size: calls and mean / min / max duration from a rocprofv3 --kernel-trace output directory.

usage: kernel_trace_summary.py "<command that was profiled>" <dir> > profiles/xxx.json"""
import collections
import csv
import glob
import json
import re
import sys


def main():
    acc = collections.defaultdict(list)
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
            name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:48]
            grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            acc[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {"command": sys.argv[1], "unit": "ns", "kernels": []}
    for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        out["kernels"].append({"kernel": name, "grid": grid, "calls": len(v), "mean": round(sum(v) / len(v), 1), "min": min(v),
                               "max": max(v), "total": sum(v)})
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
