#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> "<workload>" ["<command profiled>"] > profiles/xxx.json
FETCH_SIZE is doubled (on gfx950 a 128-byte read request is counted as 64 bytes); both counters are in KB."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(directory, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"][:40]
            acc[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    # a kernel that was launched with several grid sizes (the bench also launches the forward kernel on
    # four images at once) is reported for the size it was launched with most often
    best = {}
    for (name, grid), values in acc.items():
        if name not in best or len(values) > len(best[name][1]):
            best[name] = (grid, values)
    return {name: values for name, (grid, values) in best.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    command = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 20 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1"
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `" + command + "`; FETCH_SIZE doubled for gfx950 (128-byte read requests are "
                   "counted as 64 bytes), WRITE_SIZE taken as is; both in KB; means over the launches of each kernel "
                   "with its most frequent grid size",
           "workload": sys.argv[3], "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
        w = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
        out["kernels"][k] = {"FETCH_SIZE_KB_mean": round(f, 1), "FETCH_SIZE_launches": len(fetch.get(k, [])),
                             "WRITE_SIZE_KB_mean": round(w, 1), "WRITE_SIZE_launches": len(write.get(k, [])),
                             "hbm_bytes_per_launch": int(round((2 * f + w) * 1024))}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
