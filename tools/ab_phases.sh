#!/bin/bash
# The real library against variant builds on single 4K encodes (phases, one context, three contexts): ab_phases.sh <variant> ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for lib in "" "$@"; do
  echo "== ${lib:-real library}"
  if [ -n "$lib" ]; then export JEA_LIB=libjpegenc_amd_$lib.so; else unset JEA_LIB; fi
  timeout -k 10 120 python tools/time_encode_phases.py ${MODES:-0 2} | grep "^mode" || exit 1
done
done
