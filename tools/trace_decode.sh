#!/bin/bash
# Kernel durations of a few decodes: trace_decode.sh <tag> <w> <h> <quality> <mode> <n> -> gpurun_out/trace_decode_<tag>.csv
set -o pipefail
R="$GRAFT_REPO_ROOT"; T="$1"; shift; O="$R/gpurun_out/trd_$T"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k" -- python3 "$R/tools/decode_some.py" "$@" 10 > "$O/k.log" 2>&1 || { tail -5 "$O/k.log"; exit 1; }
cp "$O"/k/*/*kernel_stats.csv "$R/gpurun_out/trace_decode_$T.csv"
rm -rf "$O"
cut -d, -f1-4 "$R/gpurun_out/trace_decode_$T.csv" | sed 's/void jea::(anonymous namespace):://' | cut -c1-150 | head -8
