#!/bin/bash
# usage: step_timeline.sh <first kernel of a step> [bench args]; kernel by kernel what one step of bench.py does: start
# (relative to the step's first kernel), duration, workgroups -- one step from the middle of a kernel trace
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/stept"; mkdir -p "$O"; rm -rf "$O/kt"
FIRST="$1"; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --no-cpu-baseline "$@" > "$O/line.json" 2> "$O/err.log" || { tail -5 "$O/err.log"; exit 1; }
cd "$R" && python3 - "$O/kt" "$FIRST" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    return n.split("::")[-1]
steps, cur = [], None
for r in rows:
    k = short(r["Kernel_Name"])
    if k.startswith(sys.argv[2]) and not (cur and cur[-1][0].startswith(sys.argv[2])):
        if cur: steps.append(cur)
        cur = []
    if cur is not None:
        wg = 1
        for d in "XYZ":
            wg *= max(1, int(r["Grid_Size_" + d]) // max(1, int(r["Workgroup_Size_" + d])))
        cur.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), wg))
if cur: steps.append(cur)
common = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
steps = [s for s in steps if len(s) == common]
c = steps[len(steps) // 2]
print("%d steps of %d kernels; one of them:" % (len(steps), common))
t0 = c[0][1]
for k, a, b, g in c:
    print("%-60s start %8.1f us  dur %8.1f us  workgroups %d" % (k[:60], (a - t0) / 1e3, (b - a) / 1e3, g))
print("first start to last end: %.1f us" % ((c[-1][2] - t0) / 1e3))
PY
