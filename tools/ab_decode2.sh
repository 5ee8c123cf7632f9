#!/bin/bash
# The real library against variant builds on the decoder's cases of interest: ab_decode2.sh <variant> ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" "$@"; do
  echo "== ${lib:-real library}"
  if [ -n "$lib" ]; then export JEA_LIB=libjpegenc_amd_$lib.so; fi
  for c in "1920 1080 50 0 256" "1920 1080 90 0 256" "1920 1080 50 1 16" "3840 2160 75 1 1" "3840 2160 75 2 8" "3840 2160 75 0 1"; do
    timeout -k 10 100 python tools/time_decode.py $c | grep -E "phases|per call|matches" | tail -3 || exit 1
  done
done
