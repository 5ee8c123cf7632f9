"""Randomised run on the GPU box: host-buffer encode and decode of pictures of 2-12 MB (threaded, pipelined copies)."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = jea.Context(0)
bad = 0
for i in range(cases):
    w, h = int(rng.randint(900, 2600)), int(rng.randint(700, 1700))
    q = int(rng.choice([50, 75, 95]))
    rgb = po.gen_image(w, h, int(rng.randint(2)), int(rng.randint(1 << 30)))
    lum, chrom = jea.binding.quant_tables(q)
    want = po.encode(rgb, q)
    ok = bytes(ctx.encode_host(rgb, lum, chrom)) == want and np.array_equal(ctx.decode_host(want, w, h, lum, chrom), po.decode(want, w, h, q))
    if not ok:
        bad += 1
        print("MISMATCH", i, w, h, q, flush=True)
print(cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
