#!/bin/bash
# Issue counters of the kernels of bench.py --config c5 -> gpurun_out/pmc_c5.json
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmc_c5"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$O/a" -- python3 "$R/bench.py" --config c5 --no-cpu-baseline --steps 2 --warmup 1 > "$O/a.log" 2>&1 || { tail -5 "$O/a.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d "$O/b" -- python3 "$R/bench.py" --config c5 --no-cpu-baseline --steps 2 --warmup 1 > "$O/b.log" 2>&1 || { tail -5 "$O/b.log"; exit 1; }
cd "$R"
python3 tools/pmc_counters.py "bench.py --config c5 --no-cpu-baseline --steps 2 --warmup 1" "$O/a" "$O/b" > gpurun_out/pmc_c5.json
rm -rf "$O"
