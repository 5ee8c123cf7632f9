#!/bin/bash
# The real library against variant builds, decode of batches: ab_decode.sh <variant> ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" "$@"; do
  echo "== ${lib:-real library}"
  if [ -n "$lib" ]; then export JEA_LIB=libjpegenc_amd_$lib.so; fi
  timeout -k 10 100 python tools/time_decode.py 1920 1080 50 0 256 | tail -3 || exit 1
  timeout -k 10 100 python tools/time_decode.py 1920 1080 90 0 256 | tail -3 || exit 1
  timeout -k 10 100 python tools/time_decode.py 3840 2160 75 2 8 | tail -3 || exit 1
  timeout -k 10 100 python tools/time_decode.py 3840 2160 75 0 1 | tail -3 || exit 1
done
