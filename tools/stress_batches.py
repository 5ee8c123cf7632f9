"""Randomised run on the GPU box: batches of equally sized pictures through jea_encode_batch and
jea_decode_batch against the oracle.  usage: stress_batches.py [cases] [seed]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = jea.Context(0)
bad = 0
t0 = time.time()
for i in range(cases):
    w, h, n = int(rng.randint(1, 500)), int(rng.randint(1, 300)), int(rng.randint(1, 40))
    q = int(rng.choice([5, 50, 75, 90, 100]))
    lum, chrom = jea.binding.quant_tables(q)
    imgs = [po.gen_image(w, h, int(rng.randint(2)), int(rng.randint(1 << 30))) for _ in range(n)]
    want = [po.encode(im, q) for im in imgs]
    d_rgb = ctx.to_device(np.stack(imgs).reshape(-1))
    out, off, size = ctx.encode_batch(d_rgb, n, w, h, lum, chrom)
    blob = out.cpu().numpy()
    ok = all(blob[int(off[k]): int(off[k]) + int(size[k])].tobytes() == want[k] for k in range(n))
    dec = ctx.decode_batch(out, [int(o) for o in off], [int(s) for s in size], w, h, lum, chrom).cpu().numpy().reshape(n, h, w, 3)
    ok_d = all(np.array_equal(dec[k], po.decode(want[k], w, h, q)) for k in range(n))
    if not (ok and ok_d):
        bad += 1
        print("MISMATCH case", i, (w, h, n, q), "encode" if not ok else "", "decode" if not ok_d else "", flush=True)
print("%d batches, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
