#!/bin/bash
# Randomised decode-heavy runs on the GPU box with a new seed: stress_decoder.sh <seed> [scale].  Logs under gpurun_out/stress/.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
S="${1:-1}"; K="${2:-1}"; O=gpurun_out/stress; mkdir -p $O
( while sleep 45; do echo "... $(date +%T)"; done ) &
beat=$!
trap 'kill $beat 2>/dev/null' EXIT
run() { name=$1; shift; timeout -k 10 "$@" > $O/$name.log 2>&1; rc=$?; tail -1 $O/$name.log; [ $rc -eq 0 ] || exit $rc; }
run parity_$S   $((40 + 130 * K)) python -u tools/stress_parity.py $((1500 * K)) $S
run ragged_$S   $((40 + 130 * K)) python -u tools/stress_ragged.py $((400 * K)) $S
run damaged_$S  $((40 + 130 * K)) python -u tools/stress_damaged.py $((2500 * K)) $S
run batches_$S  $((40 + 130 * K)) python -u tools/stress_batches.py $((250 * K)) $S
