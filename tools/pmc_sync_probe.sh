#!/bin/bash
# vector instructions per wave of decode_sync_kernel for the real library and the probe builds present
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/syncpmc"; mkdir -p "$O"
A="${@:-1920 1080 50 0 256}"
cd /tmp && export TMPDIR=/tmp
for lib in "" $(cd "$R/jpeg-encoder_amd" && ls libjpegenc_amd_*.so 2>/dev/null); do
  rm -rf "$O/p"
  export JEA_LIB="$lib"
  timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d "$O/p" -- python3 "$R/tools/time_decode.py" $A > "$O/p.log" 2>&1 || { tail -5 "$O/p.log"; exit 1; }
  python3 - "$O/p" "${lib:-libjpegenc_amd.so}" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("decode_sync_kernel", "decode_write_kernel"):
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    w = c["SQ_WAVES"]
    print("%-26s %-20s" % (sys.argv[2], k), {x: round(v / w, 1) for x, v in c.items() if x != "SQ_WAVES"})
PY
done
