#!/bin/bash
# first-round duration of decode_sync_kernel for the real library and every variant build present (probe builds that
# stop the kernel after staging or after N walk steps): probe_sync_steps.sh [time_decode.py arguments]
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/syncp"; mkdir -p "$O"
A="${@:-1920 1080 50 0 256}"
cd /tmp && export TMPDIR=/tmp
for lib in "" $(cd "$R/jpeg-encoder_amd" && ls libjpegenc_amd_*.so 2>/dev/null); do
  rm -rf "$O/kv"
  export JEA_LIB="$lib"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/kv" -- python3 "$R/tools/time_decode.py" $A > "$O/kv.log" 2>&1 || { tail -5 "$O/kv.log"; exit 1; }
  python3 - "$O/kv" "${lib:-libjpegenc_amd.so}" <<'PY'
import csv, glob, statistics, sys
d = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_sync_kernel" in r["Kernel_Name"]:
            d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
d.sort()
big = [x for x in d if x > 0.5 * d[-1]]
print("%-28s %4d launches; the long ones (%d): median %.1f us" % (sys.argv[2], len(d), len(big), statistics.median(big)))
PY
done
