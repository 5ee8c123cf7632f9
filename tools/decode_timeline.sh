#!/bin/bash
# usage: decode_timeline.sh [w h q mode n]; wall time and phases of the decode, then every kernel's duration and the gap before it
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/dect"; mkdir -p "$O"; rm -rf "$O/kt"
A="${@:-3840 2160 75 0 1}"
cd "$R" || exit 1
timeout -k 10 200 python3 tools/time_decode.py $A > "$O/time.log" 2>&1 || { tail -5 "$O/time.log"; exit 1; }
cat "$O/time.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 "$R/tools/time_decode.py" $A > "$O/kt.log" 2>&1 || { tail -5 "$O/kt.log"; exit 1; }
cd "$R" && python3 tools/decode_timeline.py "$O/kt"
