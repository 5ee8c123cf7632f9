#!/bin/bash
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/trace_host"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for m in pinned pageable; do
  timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O/$m" -- python3 "$R/tools/trace_decode_host.py" $m > "$O/$m.log" 2>&1 || { tail -5 "$O/$m.log"; exit 1; }
  grep decode_host_batch "$O/$m.log"
done
