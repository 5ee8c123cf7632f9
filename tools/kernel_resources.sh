#!/bin/bash
# usage: kernel_resources.sh <file.hip> [name filter]  -- registers, spills, scratch, LDS and occupancy of every kernel of
# a source file, one line per kernel (compiles for gfx950 with the library's flags; nothing is linked or run)
set -e
cd "$(dirname "$0")/../jpeg-encoder_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kernel_resources.o 2>&1 | python3 -c '
import re, sys, subprocess
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void jea::\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name: continue
    print("%-62s vgpr %-4s agpr %-3s spill %-3s scratch %-5s lds %-6s occ %s" % (name[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
' "$2"
