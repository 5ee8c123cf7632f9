#!/bin/bash
# usage: gpu_tests.sh [pytest args]; the GPU suite (or the tests named), output to gpurun_out/tests.log
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest -m gpu -x -v --timeout 120 --timeout-method thread --durations=8 "${@:-tests}" > gpurun_out/tests.log 2>&1
rc=$?
grep -E "PASSED|FAILED|ERROR" gpurun_out/tests.log | grep -v PASSED | head -20
tail -15 gpurun_out/tests.log
exit $rc
