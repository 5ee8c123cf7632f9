#!/bin/bash
# Randomised runs of the tools stress_decoder.sh leaves out (row batches, sparse batches, large host cases):
# stress_rest.sh <seed>.  Logs under gpurun_out/stress/.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
S="${1:-1}"; O=gpurun_out/stress; mkdir -p $O
( while sleep 45; do echo "... $(date +%T)"; done ) &
beat=$!
trap 'kill $beat 2>/dev/null' EXIT
run() { name=$1; shift; timeout -k 10 "$@" > $O/$name.log 2>&1; rc=$?; tail -1 $O/$name.log; [ $rc -eq 0 ] || exit $rc; }
run rows_$S    250 python -u tools/stress_rows.py 60 $S
run sparse_$S  250 python -u tools/stress_sparse.py 500 $S
run large_$S   250 python -u tools/stress_large_host.py 10 $S
