#!/bin/bash
# Regenerates what profiles/ holds (files named $P_*), on the GPU box.  Results land in gpurun_out/profiles_new/.
#   bench lines of all configurations, kernel durations (rocprofv3 --kernel-trace --stats) of the headline command with
#   three contexts and with one, the forward kernel's issue counters and HBM traffic (separate --pmc passes), decode kernels.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/profiles_new"; mkdir -p "$O"
P=r03
cd "$R" || exit 1
timeout -k 10 400 python bench.py > "$O/${P}_bench_c2.json" 2> "$O/c2.err" || { tail -5 "$O/c2.err"; exit 1; }
# the driver's own arguments, for comparison with its record
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs > "$O/${P}_bench_c2_steps20_warmup5.json" 2> "$O/c2b.err" || { tail -5 "$O/c2b.err"; exit 1; }
for c in c3 c4 c5; do
  timeout -k 10 400 python bench.py --config $c > "$O/${P}_bench_$c.json" 2> "$O/$c.err" || { tail -5 "$O/$c.err"; exit 1; }
done
echo benches done
cd /tmp && export TMPDIR=/tmp
Q="--quick --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k3" -- python3 "$R/bench.py" $Q > "$O/${P}_bench_c2_3streams_under_rocprof.json" 2> "$O/k3.err" || { tail -5 "$O/k3.err"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k1" -- python3 "$R/bench.py" $Q --streams 1 > "$O/${P}_bench_c2_1stream_under_rocprof.json" 2> "$O/k1.err" || { tail -5 "$O/k1.err"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k5" -- python3 "$R/bench.py" --config c5 --no-cpu-baseline > "$O/${P}_bench_c5_under_rocprof.json" 2> "$O/k5.err" || { tail -5 "$O/k5.err"; exit 1; }
# dense content and the other configurations: kernel durations
for m in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/km$m" -- python3 "$R/bench.py" $Q --streams 1 --mode $m --pool 4 --steps 400 --warmup 40 > "$O/${P}_bench_c2_mode${m}_1stream_under_rocprof.json" 2> "$O/km$m.err" || { tail -5 "$O/km$m.err"; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k3c" -- python3 "$R/bench.py" --config c3 --no-cpu-baseline > "$O/${P}_bench_c3_under_rocprof.json" 2> "$O/k3c.err" || { tail -5 "$O/k3c.err"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k4c" -- python3 "$R/bench.py" --config c4 --no-cpu-baseline > "$O/${P}_bench_c4_under_rocprof.json" 2> "$O/k4c.err" || { tail -5 "$O/k4c.err"; exit 1; }
echo traces done
B="$Q --steps 20 --warmup 2 --ramp-seconds 0 --streams 1"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
timeout -s KILL 150 rocprofv3 --pmc $SQ --output-format csv -d "$O/pa" -- python3 "$R/bench.py" $B > "$O/pa.log" 2>&1 || { tail -5 "$O/pa.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc $SQ --output-format csv -d "$O/pb" -- python3 "$R/bench.py" $B --pool 64 --marginal-images 64 > "$O/pb.log" 2>&1 || { tail -5 "$O/pb.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pf" -- python3 "$R/bench.py" $B > "$O/pf.log" 2>&1 || { tail -5 "$O/pf.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pw" -- python3 "$R/bench.py" $B > "$O/pw.log" 2>&1 || { tail -5 "$O/pw.log"; exit 1; }
C5="--config c5 --no-cpu-baseline --steps 2 --warmup 1"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/qf" -- python3 "$R/bench.py" $C5 > "$O/qf.log" 2>&1 || { tail -5 "$O/qf.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/qw" -- python3 "$R/bench.py" $C5 > "$O/qw.log" 2>&1 || { tail -5 "$O/qw.log"; exit 1; }
for m in 1 2; do
  BM="$Q --steps 20 --warmup 2 --ramp-seconds 0 --streams 1 --mode $m --pool 4"
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/mf$m" -- python3 "$R/bench.py" $BM > "$O/mf$m.log" 2>&1 || { tail -5 "$O/mf$m.log"; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/mw$m" -- python3 "$R/bench.py" $BM > "$O/mw$m.log" 2>&1 || { tail -5 "$O/mw$m.log"; exit 1; }
done
echo counters done
cd "$R"
cp "$O"/k3/*/*kernel_stats.csv "$O/${P}_bench_c2_3streams_kernel_stats.csv"
cp "$O"/k1/*/*kernel_stats.csv "$O/${P}_bench_c2_1stream_kernel_stats.csv"
cp "$O"/k5/*/*kernel_stats.csv "$O/${P}_bench_c5_kernel_stats.csv"
python3 tools/kernel_trace_summary.py "python3 bench.py $Q --streams 1" "$O/k1" > "$O/${P}_bench_c2_1stream_kernels_by_grid.json"
python3 tools/kernel_trace_summary.py "python3 bench.py $Q" "$O/k3" > "$O/${P}_bench_c2_3streams_kernels_by_grid.json"
python3 tools/kernel_trace_summary.py "python3 bench.py --config c5 --no-cpu-baseline" "$O/k5" > "$O/${P}_bench_c5_kernels_by_grid.json"
for m in 1 2; do
  cp "$O"/km$m/*/*kernel_stats.csv "$O/${P}_bench_c2_mode${m}_1stream_kernel_stats.csv"
  python3 tools/pmc_summary.py "$O/mf$m" "$O/mw$m" "single 3840x2160 q=75, bench.py --mode $m (1 = uniform noise, 2 = photo-like)" "python3 bench.py $Q --steps 20 --warmup 2 --ramp-seconds 0 --streams 1 --mode $m --pool 4" > "$O/${P}_pmc_traffic_mode$m.json"
done
cp "$O"/k3c/*/*kernel_stats.csv "$O/${P}_bench_c3_kernel_stats.csv"
cp "$O"/k4c/*/*kernel_stats.csv "$O/${P}_bench_c4_kernel_stats.csv"
python3 tools/kernel_trace_summary.py "python3 bench.py --config c3 --no-cpu-baseline" "$O/k3c" > "$O/${P}_bench_c3_kernels_by_grid.json"
python3 tools/kernel_trace_summary.py "python3 bench.py --config c4 --no-cpu-baseline" "$O/k4c" > "$O/${P}_bench_c4_kernels_by_grid.json"
python3 tools/k1_issue_summary.py "$O/pa" "$O/pb" "$O/pf" "$O/pw" > "$O/${P}_k1_issue.json" || exit 1
python3 tools/pmc_summary.py "$O/pf" "$O/pw" "single 3840x2160 q=75 smooth" > "$O/${P}_pmc_traffic.json"
python3 tools/pmc_summary.py "$O/qf" "$O/qw" "decode of 1024 1920x1080 q=50 files in one call" "python3 bench.py $C5" > "$O/${P}_pmc_traffic_c5.json"
rm -rf "$O"/k3 "$O"/k1 "$O"/k5 "$O"/pa "$O"/pb "$O"/pf "$O"/pw "$O"/qf "$O"/qw "$O"/km1 "$O"/km2 "$O"/k3c "$O"/k4c "$O"/mf1 "$O"/mf2 "$O"/mw1 "$O"/mw2
ls -la "$O"
