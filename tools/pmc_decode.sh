#!/bin/bash
# Issue counters of the decode kernels: pmc_decode.sh <tag> <w> <h> <quality> <mode> <n> -> gpurun_out/pmc_decode_<tag>.json
set -o pipefail
R="$GRAFT_REPO_ROOT"; T="$1"; shift; O="$R/gpurun_out/pmcd_$T"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$O/a" -- python3 "$R/tools/decode_some.py" "$@" 3 > "$O/a.log" 2>&1 || { tail -5 "$O/a.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d "$O/b" -- python3 "$R/tools/decode_some.py" "$@" 3 > "$O/b.log" 2>&1 || { tail -5 "$O/b.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/f" -- python3 "$R/tools/decode_some.py" "$@" 3 > "$O/f.log" 2>&1 || { tail -5 "$O/f.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/w" -- python3 "$R/tools/decode_some.py" "$@" 3 > "$O/w.log" 2>&1 || { tail -5 "$O/w.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k" -- python3 "$R/tools/decode_some.py" "$@" 10 > "$O/k.log" 2>&1 || { tail -5 "$O/k.log"; exit 1; }
cd "$R"
python3 tools/pmc_counters.py "decode $*" "$O/a" "$O/b" "$O/f" "$O/w" > gpurun_out/pmc_decode_$T.json
cp "$O"/k/*/*kernel_stats.csv gpurun_out/pmc_decode_${T}_kernel_stats.csv
rm -rf "$O"
