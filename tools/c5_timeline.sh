#!/bin/bash
# usage: c5_timeline.sh [bench args]; kernel by kernel what one batched decode of bench.py --config c5 does: start (relative to
# the call's first kernel), duration, grid -- median call of a kernel trace
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/c5t"; mkdir -p "$O"; rm -rf "$O/kt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --config c5 --no-cpu-baseline --steps 4 --warmup 1 "$@" > "$O/line.json" 2> "$O/err.log" || { tail -5 "$O/err.log"; exit 1; }
cd "$R" && python3 - "$O/kt" <<'PY'
import csv, glob, sys, statistics
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    return n.split("::")[-1]
calls, cur = [], []
for r in rows:
    k = short(r["Kernel_Name"])
    if "decode_sync_kernel" in k and cur and not any("decode_sync" in c[0] for c in cur[-1:]) and any("idct" in c[0] for c in cur):
        calls.append(cur); cur = []
    if cur or "decode_sync_kernel" in k:
        grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else 0
        cur.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), grid))
if cur: calls.append(cur)
calls = [c for c in calls if len(c) == len(calls[-1])]
c = calls[len(calls) // 2 :][0] if calls else []
print("%d decode calls; one of them:" % len(calls))
t0 = c[0][1]
for k, a, b, g in c:
    print("%-64s start %8.1f us  dur %8.1f us  workgroups %d" % (k[:64], (a - t0) / 1e3, (b - a) / 1e3, g))
print("first start to last end: %.1f us" % ((c[-1][2] - t0) / 1e3))
PY
