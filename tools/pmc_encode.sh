#!/bin/bash
# Issue counters of the encode kernels for one 4K picture: pmc_encode.sh <mode> <tag> -> gpurun_out/pmc_encode_<tag>.json
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmce_$2"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$O/a" -- python3 "$R/tools/encode_some.py" $1 8 ${3:-75} > "$O/a.log" 2>&1 || { tail -5 "$O/a.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d "$O/b" -- python3 "$R/tools/encode_some.py" $1 8 ${3:-75} > "$O/b.log" 2>&1 || { tail -5 "$O/b.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k" -- python3 "$R/tools/encode_some.py" $1 30 ${3:-75} > "$O/k.log" 2>&1 || { tail -5 "$O/k.log"; exit 1; }
cd "$R"
python3 tools/pmc_counters.py "4K picture mode $1" "$O/a" "$O/b" > gpurun_out/pmc_encode_$2.json
cp "$O"/k/*/*kernel_stats.csv gpurun_out/pmc_encode_$2_kernel_stats.csv
rm -rf "$O"
