#!/usr/bin/env python3
"""Per-kernel means of every counter found under the rocprofv3 --pmc output directories given.

usage: pmc_counters.py "<workload>" <dir> [<dir> ...] > profiles/xxx.json
A kernel launched with several grid sizes is reported per grid size; values are means over its launches.
SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count in units of four shader cycles;
GRBM_GUI_ACTIVE is summed over the eight XCDs."""
import collections
import csv
import glob
import json
import re
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for d in sys.argv[2:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                name = m.group(1) if m else r["Kernel_Name"][:40]
                key = "%s grid=%s" % (name, r["Grid_Size"])
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[key] = {k: r.get(k) for k in ("Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
    out = {"workload": sys.argv[1], "kernels": {}}
    for key in sorted(acc):
        out["kernels"][key] = dict(meta[key])
        for c, v in sorted(acc[key].items()):
            out["kernels"][key][c] = {"mean": round(sum(v) / len(v), 1), "launches": len(v)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
