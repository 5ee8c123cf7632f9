#!/bin/bash
# kernel durations of the decode of one picture for every variant build present
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/dect"; mkdir -p "$O"
A="${@:-3840 2160 75 0 1}"
cd /tmp && export TMPDIR=/tmp
for lib in "" $(cd "$R/jpeg-encoder_amd" && ls libjpegenc_amd_*.so 2>/dev/null); do
  rm -rf "$O/kv"
  export JEA_LIB="$lib"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/kv" -- python3 "$R/tools/time_decode.py" $A > "$O/kv.log" 2>&1 || { tail -5 "$O/kv.log"; exit 1; }
  echo "== ${lib:-libjpegenc_amd.so}"
  python3 "$R/tools/decode_timeline.py" "$O/kv" | grep -v "^first"
done
