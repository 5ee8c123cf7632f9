#!/bin/bash
# Parity tests, then the bench with three contexts and with one, and the kernels' average durations.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1
tail -2 gpurun_out/t.log
grep -q " passed" gpurun_out/t.log && ! grep -q failed gpurun_out/t.log || exit 1
timeout -k 10 120 python bench.py | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('3 contexts', d['value'], d['ms_per_step'])" || exit 1
timeout -k 10 120 python bench.py --mode 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('3 contexts photo', d['value'], d['ms_per_step'])" || exit 1
cd /tmp && export TMPDIR=/tmp
for m in 0 2; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/q$m" -- python3 "$GRAFT_REPO_ROOT/bench.py" --streams 1 --steps 500 --warmup 50 --mode $m > "$GRAFT_REPO_ROOT/gpurun_out/q$m.log" 2>&1 || exit 1
  python3 - "$GRAFT_REPO_ROOT/gpurun_out/q$m" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(sys.argv[1][-2:], r['Name'].split('(')[1 if r['Name'].startswith('void') else 0][-40:], r['Calls'], round(float(r['AverageNs']) / 1000, 2))
PY
done
