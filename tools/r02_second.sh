#!/bin/bash
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/second"; mkdir -p "$O"
cd "$R" || exit 1
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > "$O/noalloc.log" 2>&1; echo "no_alloc alone rc=$?"; tail -3 "$O/noalloc.log"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$O/t.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/t.log"
timeout -k 10 200 python tools/ab_k1.py 0 > "$O/ab0.log" 2>&1 || { tail -5 "$O/ab0.log"; exit 1; }
cat "$O/ab0.log"
timeout -k 10 200 ./tools/valu_rates sweep > "$O/sweep.log" 2>&1 || exit 1
cat "$O/sweep.log"
