#!/bin/bash
# The real library against variant builds on streams that synchronise late (noise at several sizes and qualities):
# ab_decode_dense.sh <variant> ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" "$@"; do
  echo "== ${lib:-real library}"
  if [ -n "$lib" ]; then export JEA_LIB=libjpegenc_amd_$lib.so; fi
  for c in "3840 2160 75 1 1" "1920 1080 50 1 16" "1024 1024 90 1 1" "1024 1024 95 1 1" "1920 1080 90 1 4" "512 512 95 1 64"; do
    timeout -k 10 100 python tools/time_decode.py $c | grep -E "phases|per call|matches" | tail -3
  done
done
