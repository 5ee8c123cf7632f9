#!/bin/bash
# Regenerates what profiles/ holds, on the GPU box: bench line, kernel stats with three contexts and with one,
# the bench lines measured under the profiler, HBM traffic from two counter passes, decode kernel stats.
# Results land in gpurun_out/profiles_new/.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/profiles_new"; mkdir -p "$O"
cd "$R" || exit 1
timeout -k 10 200 python bench.py > "$O/r01_bench_single4k.json" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k3" -- python3 "$R/bench.py" --no-cpu-baseline > "$O/r01_bench_single4k_under_rocprof.json" 2> "$O/k3.err" || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k1" -- python3 "$R/bench.py" --no-cpu-baseline --streams 1 > "$O/r01_bench_single4k_1stream_under_rocprof.json" 2> "$O/k1.err" || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pf" -- python3 "$R/bench.py" --steps 20 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1 > "$O/pf.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pw" -- python3 "$R/bench.py" --steps 20 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1 > "$O/pw.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kd" -- python3 "$R/tools/time_decode.py" 1920 1080 50 0 1024 > "$O/kd.log" 2>&1 || exit 1
cp "$O"/k3/*/*kernel_stats.csv "$O/r01_bench_single4k_kernel_stats.csv"
cp "$O"/k1/*/*kernel_stats.csv "$O/r01_bench_single4k_1stream_kernel_stats.csv"
cp "$O"/kd/*/*kernel_stats.csv "$O/r01_decode_1024x1080p_q50_kernel_stats.csv"
python3 "$R/tools/pmc_summary.py" "$O/pf" "$O/pw" "single 3840x2160 q=75 smooth" > "$O/r01_pmc_traffic.json"
rm -rf "$O"/k3 "$O"/k1 "$O"/kd "$O"/pf "$O"/pw
ls -la "$O"
