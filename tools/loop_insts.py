"""Counts the instructions of the innermost loops of one kernel in a gfx950 assembly listing (hipcc -S):
loop_insts.py <file.s> <kernel name substring> -- per loop (label of its header): vector, scalar, LDS and memory
instructions between the header and the last branch back to it."""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = None
for i, l in enumerate(lines):
    if l.startswith("_Z") and ":" in l and want in l.split(":")[0]:
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\ts_endpgm"))
body = lines[start:end]
labels = {l[:-1].split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
loops = {}
for i, l in enumerate(body):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops[m.group(1)] = max(loops.get(m.group(1), 0), i)
for lab, last in sorted(loops.items(), key=lambda kv: labels[kv[0]]):
    seg = [l.strip() for l in body[labels[lab]:last + 1] if l.startswith("\t")]
    kinds = {"v_": 0, "s_": 0, "ds_": 0, "global_": 0, "v_mov": 0, "v_cndmask": 0}
    for l in seg:
        for k in kinds:
            if l.startswith(k):
                kinds[k] += 1
    print(lab, "lines", labels[lab], "-", last, {k: v for k, v in kinds.items()})
