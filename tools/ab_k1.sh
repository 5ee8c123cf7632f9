#!/bin/bash
# Parity tests of the transform stages, then the A/B of the forward-kernel builds present.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/ab"; mkdir -p "$O"
cd "$R" || exit 1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "fdct or quant or colour or encode or golden or component" > "$O/t.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/t.log"
grep -q " passed" "$O/t.log" && ! grep -q failed "$O/t.log" || exit 1
timeout -k 10 200 python tools/ab_k1.py ${1:-0} > "$O/ab.log" 2>&1 || { tail -5 "$O/ab.log"; exit 1; }
cat "$O/ab.log"
