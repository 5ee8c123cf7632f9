#!/usr/bin/env python3
"""Vector-instruction issue figures of the fused forward kernel -> profiles/r02_k1_issue.json.

usage: k1_issue_summary.py <issue-counter pass, 1 image/launch> <issue-counter pass, 64 images/launch> <FETCH_SIZE pass> <WRITE_SIZE pass>

Dynamic counts (instructions per wave, waves, cycles) come from the rocprofv3 --pmc passes.  What share of the
instructions issues at the full rate and what share at half of it comes from the kernel's ISA listing: the
listing of fdct_quant_rows_kernel with the pass-through colour branch compiled out (it never runs for
RGBToYCbCrMapper) and without the basic blocks that hold f64 instructions (the tie fall-back, run for about
one pixel in a thousand).  Issue costs per class: tools/valu_rates.hip on this chip.
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL_NS, HALF_NS = 1.03, 1.72
FULL_RATE = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32",
             "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mul_legacy_f32"}


def counters(directory, kernel, grid=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    acc[int(r["Grid_Size"])]["_duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not acc:
        return None, {}
    g = min(acc) if grid == "smallest" else max(acc)   # single-picture launches, or the largest launch
    return g, {c: sum(v) / len(v) for c, v in acc[g].items()}


def listing_classes():
    src = open(os.path.join(ROOT, "jpeg-encoder_amd", "csrc", "fdct.hip")).read()
    assert "if (mode.passthrough) {" in src
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "fdct.hip")
        open(path, "w").write(src.replace("if (mode.passthrough) {", "if (false) {"))
        out = os.path.join(tmp, "fdct.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                        "-fno-slp-vectorize", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "jpeg-encoder_amd", "csrc"),
                        path, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    m = re.search(r"^(_ZN3jea12_GLOBAL__N_122fdct_quant_rows_kernelILb1E[^\n]*):.*?\n(.*?)s_endpgm", text, re.S | re.M)
    blocks, cur = [], []
    for line in m.group(2).split("\n"):
        t = line.strip()
        if not t or t[0] in ";.":
            if t.startswith(".LBB") or t.endswith(":"):
                blocks.append(cur)
                cur = []
            continue
        if t.endswith(":"):
            blocks.append(cur)
            cur = []
            continue
        cur.append(t)
    blocks.append(cur)
    full = half = other = 0
    for b in blocks:
        if any("_f64" in i for i in b):
            continue
        for i in b:
            op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", i.split()[0])
            if not op.startswith("v_"):
                other += 1
                continue
            operands = i.split(None, 1)[1] if " " in i else ""
            scalar_operand = re.search(r"(?<![\w\[])s\d+\b|s\[\d+:\d+\]", operands.split(",", 1)[1] if "," in operands else "") is not None
            if op in FULL_RATE and not scalar_operand:
                full += 1
            else:
                half += 1
    return full, half, other


def main():
    one_dir, many_dir, fetch_dir, write_dir = sys.argv[1:5]
    many_images = int(sys.argv[5]) if len(sys.argv) > 5 else 64   # images per launch in the second pass
    kernel = "fdct_quant_rows_kernel"
    g1, c1 = counters(one_dir, kernel, "smallest")
    gm, cm = counters(many_dir, kernel)
    _, cf = counters(fetch_dir, kernel, "smallest")
    _, cw = counters(write_dir, kernel, "smallest")
    full, half, other = listing_classes()
    waves = c1["SQ_WAVES"]
    insts = c1["SQ_INSTS_VALU"] / waves
    share = half / (full + half)
    # the listing's split applied to the measured instruction count
    issue_ns = insts * ((1 - share) * FULL_NS + share * HALF_NS)
    out = {
        "_how": "rocprofv3 --pmc passes over `python3 bench.py --quick --steps 20 --warmup 2 --ramp-seconds 0 --no-cpu-baseline "
                "--streams 1` (single-image launches, the kernel's smallest grid) and the same with --pool 64 --marginal-images 64 (64 images per launch in the "
                "roofline section); SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count in units of four cycles; "
                "GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE doubled for gfx950 (a 128-byte request counts as 64), "
                "both in KB; instruction classes from the ISA listing (tools/k1_issue_summary.py), issue costs from tools/valu_rates.hip",
        "kernel": kernel, "workload": "3840x2160 q=75 smooth, one image per launch",
        "waves_per_launch": int(waves), "valu_insts_per_wave": round(insts, 1),
        "listing_full_rate_insts": full, "listing_half_rate_insts": half, "listing_scalar_and_memory_insts": other,
        "half_rate_share": round(share, 4),
        "issue_ns_per_wave": round(issue_ns, 1),
        "issue_bound_us_per_image": round(issue_ns * waves / 1024 / 1e3, 2),
        "counters_single_image_launch": {k: round(v, 1) for k, v in sorted(c1.items())},
        "hbm_bytes_per_launch": int(round((2 * cf.get("FETCH_SIZE", 0.0) + cw.get("WRITE_SIZE", 0.0)) * 1024)) if cf and cw else None,
    }
    if cm:
        cycles = cm["GRBM_GUI_ACTIVE"] / 8
        out["counters_many_image_launch"] = {"grid": gm, **{k: round(v, 1) for k, v in sorted(cm.items())}}
        if cm.get("_duration_ns"):
            # GRBM_GUI_ACTIVE / 8 / duration: within a few per cent of the in-kernel clock on dispatches of a millisecond and more
            out["shader_clock_mhz"] = round(cycles / cm["_duration_ns"] * 1e3)
        busy = cm["SQ_INSTS_VALU"] / 1024 / cycles
        out["many_image_launch"] = {
            "images": many_images, "gui_active_cycles_per_xcd": round(cycles),
            "valu_insts_per_simd_per_cycle": round(busy, 4),
            "cycles_per_valu_inst_per_simd": round(1 / busy, 3),
            "mean_waves_per_simd": round(cm["SQ_WAVE_CYCLES"] * 4 / 1024 / cycles, 2),
            "wave_cycle_split": {"issuing_valu": round(cm["SQ_ACTIVE_INST_VALU"] / cm["SQ_WAVE_CYCLES"], 4),
                                 "issuing_any": round(cm["SQ_ACTIVE_INST_ANY"] / cm["SQ_WAVE_CYCLES"], 4),
                                 "waiting_to_issue": round(cm["SQ_WAIT_INST_ANY"] / cm["SQ_WAVE_CYCLES"], 4),
                                 "parked_waitcnt_or_barrier": round(cm["SQ_WAIT_ANY"] / cm["SQ_WAVE_CYCLES"], 4)}}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
