#!/bin/bash
# Kernel durations and instruction counts of the encoder's kernels on dense content (bench.py --mode 1 noise, 2 photo-like),
# one context, on the GPU box.  Results under gpurun_out/dense/.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/dense"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
Q="--quick --no-cpu-baseline --no-other-configs --streams 1 --pool 4"
for m in ${MODES:-1 2}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k$m" -- python3 "$R/bench.py" $Q --mode $m --steps 400 --warmup 40 > "$O/bench_mode$m.json" 2> "$O/k$m.err" || { tail -5 "$O/k$m.err"; exit 1; }
  cp "$O"/k$m/*/*kernel_stats.csv "$O/mode${m}_kernel_stats.csv"
  timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR --output-format csv -d "$O/p$m" -- python3 "$R/bench.py" $Q --mode $m --steps 20 --warmup 2 --ramp-seconds 0 > "$O/p$m.log" 2>&1 || { tail -5 "$O/p$m.log"; exit 1; }
  python3 - "$O/p$m" $m <<'PY' > "$O/mode${m}_counters.json"
import csv, glob, sys, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            n[k] += 1
out = {"command": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR -- python3 bench.py --quick --no-cpu-baseline --no-other-configs --streams 1 --pool 4 --mode %s --steps 20 --warmup 2 --ramp-seconds 0" % sys.argv[2],
       "per_wave_means": {}}
for k, c in acc.items():
    if not k.startswith(("fdct", "tile_bits", "pack_stuff")):
        continue
    w = c.get("SQ_WAVES", 1) or 1
    out["per_wave_means"][k] = dict(launches=n[k], waves_per_launch=round(w / max(n[k], 1), 1), **{x: round(v / w, 1) for x, v in c.items() if x != "SQ_WAVES"})
print(json.dumps(out, indent=1))
PY
done
for m in ${MODES:-1 2}; do echo "mode $m"; cut -d, -f1-4 "$O/mode${m}_kernel_stats.csv" | cut -c1-120 | head -5; done
