#!/usr/bin/env python3
"""Opcode histogram of one kernel in a hipcc -S listing, priced with the per-instruction issue times
measured by tools/valu_rates.hip on MI355X (ns per wave-instruction per SIMD, four waves per SIMD)."""
import collections
import re
import sys

FULL = 1.2
HALF = 1.85
COST = {"v_add_f32": FULL, "v_sub_f32": FULL, "v_mul_f32": FULL, "v_fma_f32": FULL, "v_fmac_f32": FULL,
        "v_mac_f32": FULL, "v_and_b32": FULL, "v_or_b32": FULL, "v_xor_b32": FULL, "v_add_u32": FULL,
        "v_sub_u32": FULL, "v_lshrrev_b32": FULL, "v_lshlrev_b32": FULL, "v_mov_b32": FULL, "v_subrev_f32": FULL,
        "v_pk_mul_f32": 1.96, "v_pk_add_f32": 1.96, "v_pk_fma_f32": 1.85, "v_rcp_f32": 3.5}


def main(path, pattern):
    s = open(path).read()
    m = re.search(r"^(%s[^\n]*):.*?\n(.*?)s_endpgm" % pattern, s, re.S | re.M)
    ops = collections.Counter()
    for line in m.group(2).split("\n"):
        line = line.strip()
        if not line or line[0] in ";." or line.endswith(":"):
            continue
        op = line.split()[0]
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        ops[op] += 1
    total = 0.0
    for k, v in ops.most_common():
        c = 0.0 if k.startswith("s_") else COST.get(k, HALF)
        total += c * v
    print("instructions:", sum(ops.values()), " VALU+mem issue estimate: %.0f ns per wave" % total)
    for k, v in ops.most_common(40):
        c = 0.0 if k.startswith("s_") else COST.get(k, HALF)
        print("%-26s %5d  %7.0f ns" % (k, v, c * v))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
