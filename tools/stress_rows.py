"""Randomised run on the GPU box aimed at the kernels for pictures whose sides are multiples of 8: batches through
jea_encode_batch (row-lane forward kernel, one to six units per wave) and jea_decode_batch (from 4096 groups of 64 MCUs
on and where nine files in ten are small, the inverse kernel for single-colour groups and its list) against the oracle.
usage: stress_rows.py [cases] [seed]"""
import concurrent.futures
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = jea.Context(0)
pool = concurrent.futures.ThreadPoolExecutor(16)


def content(w, h, kind, seed):
    r = np.random.RandomState(seed)
    if kind == 0:
        return po.gen_image(w, h, 0, seed)
    if kind == 1:
        return po.gen_image(w, h, 1, seed)
    if kind == 2:   # flat areas of random colours with a few noisy or textured patches
        img = np.empty((h, w, 3), np.uint8)
        img[:] = r.randint(0, 256, 3)
        for _ in range(r.randint(1, 6)):
            y0, x0 = r.randint(0, h), r.randint(0, w)
            y1, x1 = y0 + r.randint(1, h + 1), x0 + r.randint(1, w + 1)
            img[y0:y1, x0:x1] = r.randint(0, 256, 3) if r.randint(2) else r.randint(0, 256, img[y0:y1, x0:x1].shape)
        return img
    if kind == 3:   # low-amplitude noise around a level: few, small coefficients
        return np.clip(r.randint(0, 256, 3) + r.randint(-6, 7, (h, w, 3)), 0, 255).astype(np.uint8)
    img = po.gen_image(w, h, 0, seed).astype(np.int16)   # smooth with an edge inside some block columns
    img[:, :, r.randint(3)] += np.where((np.arange(w) % 8) < 4, -20, 20).astype(np.int16)[None, :] * (np.arange(w) // 8 % r.randint(1, 9) == 0)[None, :]
    return np.clip(img, 0, 255).astype(np.uint8)


t0 = time.time()
bad = 0
routed = 0
for i in range(cases):
    small = rng.randint(2) == 0                      # many small pictures (the decoder's two-kernel route) or fewer large ones
    w = 8 * int(rng.randint(8, 40 if small else 200))
    h = 8 * int(rng.randint(1, 24 if small else 120))
    groups = ((w // 8) * (h // 8) + 63) // 64
    n = int(rng.randint(4096 // groups, 4 * 4096 // groups + 2)) if small else int(rng.randint(1, max(2, int(2.5e7 / (w * h)))))
    n = max(1, min(n, int(4e7 / (w * h))))
    q = int(rng.choice([10, 50, 50, 75, 90, 100]))
    kinds = rng.choice([0, 2, 3, 4] if small else [0, 1, 2, 3, 4], size=rng.randint(1, 4))
    distinct = min(n, 24)
    imgs = list(pool.map(lambda k: content(w, h, int(kinds[k % len(kinds)]), 1000 * i + k), range(distinct)))
    lum, chrom = jea.binding.quant_tables(q)
    want = list(pool.map(lambda im: po.encode(im, q), imgs))
    src = [k % distinct for k in range(n)]
    d_rgb = ctx.to_device(np.stack([imgs[k] for k in src]).reshape(-1))
    out, off, size = ctx.encode_batch(d_rgb, n, w, h, lum, chrom)
    blob = out.cpu().numpy()
    ok = all(blob[int(off[k]): int(off[k]) + int(size[k])].tobytes() == want[src[k]] for k in range(n))
    small_files = sum(len(want[s]) - 607 < 2.2 * (w // 8) * (h // 8) for s in src)
    routed += n * groups >= 4096 and 10 * small_files >= 9 * n
    dec = ctx.decode_batch(out, [int(o) for o in off], [int(s) for s in size], w, h, lum, chrom).cpu().numpy().reshape(n, h, w, 3)
    ref = list(pool.map(lambda k: po.decode(want[k], w, h, q), range(distinct)))
    ok_d = all(np.array_equal(dec[k], ref[src[k]]) for k in range(n))
    if not (ok and ok_d):
        bad += 1
        print("MISMATCH case", i, (w, h, n, q, list(kinds)), "encode" if not ok else "", "decode" if not ok_d else "", flush=True)
    if i % 10 == 9:
        print("case", i + 1, "of", cases, "%.0f s" % (time.time() - t0), flush=True)
print("%d batches (%d through the decoder's two-kernel route), %d mismatches, %.1f s" % (cases, routed, bad, time.time() - t0))
sys.exit(1 if bad else 0)
