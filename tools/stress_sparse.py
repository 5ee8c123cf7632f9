"""Randomised run on the GPU box aimed at the decoder's two-kernel inverse route (idct_sparse_kernel and the list it
hands to idct_dequant_kernel): batches of mostly single-colour pictures -- large flat areas, a few clusters of detail, so
that groups of 64 MCUs with none, a few, more than 16 and only detailed MCUs all occur -- with widths that are multiples
of 8, any height, both colour mappers, encoded by the oracle and decoded by jea_decode_batch, against the oracle's decode.
usage: stress_sparse.py [cases] [seed]"""
import concurrent.futures
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pool = concurrent.futures.ThreadPoolExecutor(16)
ctxs = {0: jea.Context(0), 1: jea.Context(0)}
ctxs[1].set_colour_mapper(1)


def picture(w, h, seed, clusters):
    r = np.random.RandomState(seed)
    img = np.empty((h, w, 3), np.uint8)
    img[:] = r.randint(0, 256, 3)
    for _ in range(r.randint(0, 4)):   # large flat areas, MCU-aligned (three in four) or not
        a = 8 if r.randint(4) else 1
        y0, x0 = r.randint(0, h) // a * a, r.randint(0, w) // a * a
        img[y0:y0 + r.randint(8, h + 8), x0:x0 + r.randint(8, w + 8)] = r.randint(0, 256, 3)
    for _ in range(clusters):          # detail: noise, a soft ramp, or single pixels
        y0, x0 = r.randint(0, h), r.randint(0, w)
        hh, ww = r.randint(1, 20), r.randint(1, 64)
        kind = r.randint(3)
        part = img[y0:y0 + hh, x0:x0 + ww]
        if kind == 0:
            part[:] = r.randint(0, 256, part.shape)
        elif kind == 1:
            part[:] = np.clip(part.astype(np.int16) + (np.arange(part.shape[1]) % 23)[None, :, None] * r.randint(1, 5), 0, 255)
        else:
            part[::r.randint(1, 9), ::r.randint(1, 9)] = r.randint(0, 256, 3)
    return img


t0 = time.time()
bad = routed = 0
for i in range(cases):
    w = 8 * int(rng.randint(8, 320))
    h = int(rng.randint(8, 260))
    mcus = (w // 8) * ((h + 7) // 8)
    groups = (mcus + 63) // 64
    n = int(rng.randint(4096 // groups + 1, 4 * 4096 // groups + 3))
    n = max(1, min(n, int(6e7 / (w * h))))
    q = int(rng.choice([25, 50, 50, 60, 75]))
    mapper = int(rng.randint(2))
    distinct = min(n, 16)
    busy = rng.randint(distinct + 1) if rng.randint(3) == 0 else 0   # some batches carry a few files with much detail
    limit = 2.8 if mapper else 2.2
    po.set_mapper(mapper)

    def make(k):   # a picture and its file; the quiet ones are made again with less detail until the file is small
        clusters = 40 if k < busy else int(rng.randint(0, 6))
        while True:
            im = picture(w, h, 1000 * i + k + 100 * clusters, clusters)
            f = po.encode(im, q)
            if k < busy or clusters == 0 or len(f) - 607 < limit * mcus:
                return im, f
            clusters //= 2

    made = list(pool.map(make, range(distinct)))
    imgs, files = [m[0] for m in made], [m[1] for m in made]
    ref = list(pool.map(lambda k: po.decode(files[k], w, h, q), range(distinct)))
    po.set_mapper(0)
    src = [int(rng.randint(distinct)) for _ in range(n)]
    small_files = sum(len(files[s]) - 607 < limit * mcus for s in src)
    routed += n * groups >= 4096 and 10 * small_files >= 9 * n
    blob = np.frombuffer(b"".join(files[s] for s in src), np.uint8)
    sizes = [len(files[s]) for s in src]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
    lum, chrom = jea.binding.quant_tables(q)
    ctx = ctxs[mapper]
    dec = ctx.decode_batch(ctx.to_device(blob), offs, sizes, w, h, lum, chrom).cpu().numpy().reshape(n, h, w, 3)
    ok = all(np.array_equal(dec[k], ref[src[k]]) for k in range(n))
    if not ok:
        bad += 1
        print("MISMATCH case", i, (w, h, n, q, mapper), flush=True)
    if i % 10 == 9:
        print("case", i + 1, "of", cases, "routed", routed, "%.0f s" % (time.time() - t0), flush=True)
print("%d batches (%d through the decoder's two-kernel route), %d mismatches, %.1f s" % (cases, routed, bad, time.time() - t0))
sys.exit(1 if bad else 0)
