#!/bin/bash
# The real library against variant builds on smooth frames at several qualities (where the decoder's two-kernel route
# starts and stops paying): ab_decode_c5.sh <variant> ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" "$@"; do
  echo "== ${lib:-real library}"
  if [ -n "$lib" ]; then export JEA_LIB=libjpegenc_amd_$lib.so; fi
  for q in ${QUALITIES:-25 50 60 75 85}; do
    timeout -k 10 100 python tools/time_decode.py 1920 1080 $q 0 256 | grep -E "^decode|phases" | tail -2 || exit 1
  done
done
