"""Randomised parity run on the GPU box: encode and decode of random shapes, qualities and contents against the
oracle.  usage: stress_parity.py [cases] [seed]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = jea.Context(0)


def content(w, h, kind):
    if kind == 0:
        return po.gen_image(w, h, 0, int(rng.randint(1 << 30)))
    if kind == 1:
        return po.gen_image(w, h, 1, int(rng.randint(1 << 30)))
    if kind == 2:   # flat areas of random colours with noisy patches
        img = np.empty((h, w, 3), np.uint8)
        img[:] = rng.randint(0, 256, 3)
        for _ in range(rng.randint(1, 6)):
            y0, x0 = rng.randint(0, h), rng.randint(0, w)
            y1, x1 = y0 + rng.randint(1, h + 1), x0 + rng.randint(1, w + 1)
            if rng.randint(2):
                img[y0:y1, x0:x1] = rng.randint(0, 256, 3)
            else:
                img[y0:y1, x0:x1] = rng.randint(0, 256, img[y0:y1, x0:x1].shape)
        return img
    if kind == 3:   # low-amplitude noise around a level: few, small coefficients
        base = rng.randint(0, 256, 3)
        return np.clip(base + rng.randint(-6, 7, (h, w, 3)), 0, 255).astype(np.uint8)
    # saturated colours and hard edges
    img = (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)
    return np.repeat(np.repeat(img[::8, ::8], 8, 0), 8, 1)[:h, :w].copy() if rng.randint(2) else img


t0 = time.time()
bad = 0
for i in range(cases):
    w, h = int(rng.randint(1, 900)), int(rng.randint(1, 500))
    if rng.randint(4) == 0:
        w, h = 8 * (w // 8 + 1), 8 * (h // 8 + 1)
    q = int(rng.choice([1, 5, 10, 25, 50, 75, 90, 95, 100]))
    kind = int(rng.randint(5))
    rgb = np.ascontiguousarray(content(w, h, kind))
    lum, chrom = jea.binding.quant_tables(q)
    try:
        want = po.encode(rgb, q)
    except Exception as e:   # the oracle refuses what the reference cannot code
        print("case", i, (w, h, q, kind), "oracle:", e)
        continue
    got = ctx.encode_host(rgb, lum, chrom)
    ok_e = bytes(got) == want
    dec = ctx.decode_host(want, w, h, lum, chrom)
    ok_d = np.array_equal(dec, po.decode(want, w, h, q))
    if not (ok_e and ok_d):
        bad += 1
        print("MISMATCH case", i, (w, h, q, kind), "encode ok" if ok_e else "ENCODE", "decode ok" if ok_d else "DECODE", flush=True)
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
