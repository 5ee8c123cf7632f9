"""Randomised run on the GPU box: damaged scans must fail in both decoders or decode to the same picture in both.
usage: stress_damaged.py [cases] [seed]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = jea.Context(0)
outcomes = {"both fail": 0, "both decode": 0}
bad_cases = 0
t0 = time.time()
for i in range(cases):
    w, h = int(rng.randint(8, 400)), int(rng.randint(8, 300))
    q = int(rng.choice([10, 50, 75, 95, 100]))
    mode = int(rng.randint(2))
    lum, chrom = jea.binding.quant_tables(q)
    jpeg = po.encode(po.gen_image(w, h, mode, int(rng.randint(1 << 30))), q)
    if len(jpeg) < 612:
        continue
    v = bytearray(jpeg)
    kind = rng.randint(5)
    if kind == 0:
        for _ in range(rng.randint(1, 4)):
            v[rng.randint(607, len(v) - 2)] ^= 1 << rng.randint(8)
    elif kind == 1:
        cut = rng.randint(1, max(2, len(v) - 609))
        v = v[: len(v) - 2 - cut] + b"\xff\xd9"
    elif kind == 2:
        v = v[:-2] + bytes(rng.randint(0, 256, rng.randint(1, 400)).astype(np.uint8)) + b"\xff\xd9"
    elif kind == 3:
        a = rng.randint(607, len(v) - 2)
        v[a: a + rng.randint(1, 64)] = b""
    else:
        a = rng.randint(607, len(v) - 2)
        v[a:a] = bytes(rng.randint(0, 256, rng.randint(1, 16)).astype(np.uint8))
    v = bytes(v)
    try:
        want = po.decode(v, w, h, q)
    except RuntimeError:
        want = None
    try:
        got = ctx.decode_host(v, w, h, lum, chrom)
    except jea.JeaError:
        got = None
    ok = (want is None) == (got is None) and (want is None or np.array_equal(got, want))
    if not ok:
        bad_cases += 1
        print("MISMATCH case", i, (w, h, q, mode, int(kind)), "oracle", "fails" if want is None else "decodes", "gpu",
              "fails" if got is None else "decodes", flush=True)
    outcomes["both fail" if want is None else "both decode"] += 1
print(cases, "cases", outcomes, bad_cases, "mismatches, %.1f s" % (time.time() - t0))
sys.exit(1 if bad_cases else 0)
