#!/bin/bash
# LDS and issue counters of the forward kernel for 4K pictures -> gpurun_out/pmc_k1_lds_<tag>.json
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmcl_$1"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 60 rocprofv3 --list-avail > "$R/gpurun_out/counters_avail.txt" 2>&1
grep -o "SQ_LDS[A-Z_]*\|SQ_INSTS_BRANCH[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_THREAD_CYCLES_VALU\|SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_LDS\|SQ_INSTS_LDS" "$R/gpurun_out/counters_avail.txt" | sort -u | tr '\n' ' '; echo
run() { timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d "$O/$1" -- python3 "$R/tools/encode_some.py" 0 8 75 > "$O/$1.log" 2>&1 || { tail -5 "$O/$1.log"; exit 1; }; }
run a "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
run b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
cd "$R"
python3 tools/pmc_counters.py "4K picture mode 0" "$O/a" "$O/b" "$O/c" > gpurun_out/pmc_k1_lds_$1.json
rm -rf "$O"
python3 - <<'PY' "$R/gpurun_out/pmc_k1_lds_$1.json"
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d["kernels"].items():
    if "fdct" in k:
        print(k, {c:(x["mean"] if isinstance(x,dict) else x) for c,x in v.items()})
PY
