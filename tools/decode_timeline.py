"""Reads a rocprofv3 --kernel-trace csv of tools/time_decode.py: for the kernels of one decode call in the back-to-back
section, their durations and the idle time before each (median over the calls)."""
import collections, csv, glob, statistics, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    return n.split("::")[-1].split("<")[0]
# a decode call ends with the inverse transform; kernels of the encode that made the file come before the first call
calls, cur = [], []
for r in rows:
    k = short(r["Kernel_Name"])
    if k.startswith("__amd") and not cur:
        continue
    cur.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if k == "idct_dequant_kernel":
        calls.append(cur)
        cur = []
calls = [c for c in calls if len(c) == len(calls[len(calls) // 2])][5:-5]
n = len(calls[0])
print("%d calls of %d kernels" % (len(calls), n))
total = 0.0
for i in range(n):
    dur = statistics.median(c[i][2] - c[i][1] for c in calls) / 1e3
    gap = statistics.median(c[i][1] - c[i - 1][2] for c in calls) / 1e3 if i else 0.0
    total += dur + gap
    print("%-28s gap %6.2f us  kernel %7.2f us" % (calls[0][i][0], gap, dur))
print("first start to last end: %.1f us (sum of the medians)" % total)
