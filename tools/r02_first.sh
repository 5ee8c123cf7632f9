#!/bin/bash
# First GPU pass of the round: parity suite, A/B of forward-kernel builds, shader clock under VALU load,
# issue counters of the forward kernel.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/first"; mkdir -p "$O"
cd "$R" || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$O/t.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/t.log"
grep -q " passed" "$O/t.log" || exit 1
timeout -k 10 200 python tools/ab_k1.py 0 > "$O/ab0.log" 2>&1 || { tail -5 "$O/ab0.log"; exit 1; }
cat "$O/ab0.log"
timeout -k 10 120 ./tools/valu_rates clock > "$O/clock.log" 2>&1 || exit 1
tail -4 "$O/clock.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 -L > "$O/counters_list.txt" 2>&1
B="--steps 20 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$O/pa" -- python3 "$R/bench.py" $B > "$O/pa.log" 2>&1 || { tail -5 "$O/pa.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$O/pb" -- python3 "$R/bench.py" $B --batch 16 > "$O/pb.log" 2>&1 || { tail -5 "$O/pb.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "$O/pc" -- python3 "$R/bench.py" $B > "$O/pc.log" 2>&1 || { tail -5 "$O/pc.log"; exit 1; }
head -2 "$O"/pa/*/*counter_collection.csv
python3 "$R/tools/pmc_counters.py" "single 3840x2160 q=75 smooth, 1 stream" "$O/pa" "$O/pc" > "$O/issue_b1.json"
python3 "$R/tools/pmc_counters.py" "16 x 3840x2160 q=75 smooth per launch, 1 stream" "$O/pb" > "$O/issue_b16.json"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kb" -- python3 "$R/bench.py" $B --batch 16 > "$O/kb.log" 2>&1 || exit 1
cp "$O"/kb/*/*kernel_stats.csv "$O/b16_kernel_stats.csv"
rm -rf "$O"/pa "$O"/pb "$O"/pc "$O"/kb
ls "$O"
