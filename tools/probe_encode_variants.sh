#!/bin/bash
# Times the real library and probe builds of it (tools/build_variant.sh) with tools/time_encode_phases.py.
# usage: probe_encode_variants.sh "<modes>" <variant> ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
modes="$1"; shift
echo "== real library"
timeout -k 10 120 python tools/time_encode_phases.py $modes || exit 1
for v in "$@"; do
  echo "== $v"
  JEA_LIB=libjpegenc_amd_$v.so timeout -k 10 120 python tools/time_encode_phases.py $modes || exit 1
done
