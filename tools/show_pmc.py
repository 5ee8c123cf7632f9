"""Prints the per-wave means of a tools/pmc_counters.py file for the kernels whose names contain one of the words given:
show_pmc.py <file.json> <word> ..."""
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    if not any(w in k for w in sys.argv[2:]):
        continue
    waves = v.get("SQ_WAVES", {}).get("mean")
    print(k)
    for c, x in sorted(v.items()):
        if isinstance(x, dict) and "mean" in x:
            print("   %-22s %14.0f  per wave %10.1f" % (c, x["mean"], x["mean"] / waves if waves else float("nan")))
