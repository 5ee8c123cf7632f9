#!/bin/bash
# Parity tests that exercise the forward kernel, its launch times (tools/ab_k1.py), then its LDS counters.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/k1c"; mkdir -p "$O"
cd "$R" || exit 1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "fdct or quant or colour or encode or golden or component or c3 or c4" > "$O/t.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/t.log"
grep -q " passed" "$O/t.log" && ! grep -q failed "$O/t.log" || exit 1
timeout -k 10 120 python tools/ab_k1.py ${1:-0} 2>&1 | grep -v amdgpu.ids || exit 1
bash tools/pmc_k1_lds.sh ${2:-rows}
