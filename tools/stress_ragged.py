"""Randomised run on the GPU box: batches whose files differ wildly in size (smooth, noisy, flat and periodic pictures of
one size in one batch, 1 to 400 of them) through jea_decode_batch and, every few cases, through the host-batch
entry points, against the oracle.  usage: stress_ragged.py [cases] [seed]"""
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
last = t0
for i in range(cases):
    w, h = int(rng.randint(8, 400)), int(rng.randint(8, 260))
    n = int(rng.choice([1, 2, 5, 16, 17, 18, 40, 130, 400])) if w * h < 40000 else int(rng.randint(1, 40))
    q = int(rng.choice([10, 50, 75, 92, 100]))
    lum, chrom = jea.binding.quant_tables(q)
    kinds = []
    for k in range(int(rng.randint(2, 6))):
        kind = int(rng.randint(5))
        if kind == 0:
            im = po.gen_image(w, h, 0, int(rng.randint(1 << 30)))
        elif kind == 1:
            im = po.gen_image(w, h, 1, int(rng.randint(1 << 30)))
        elif kind == 2:
            im = np.full((h, w, 3), int(rng.randint(256)), np.uint8)
        elif kind == 3:
            im = np.full((h, w, 3), 100, np.uint8)
            im[:, ::16] = 200
        else:
            im = (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)
        kinds.append(im)
    files = [po.encode(im, q) for im in kinds]
    want = [po.decode(f, w, h, q) for f in files]
    pick = [int(rng.randint(len(kinds))) for _ in range(n)]
    offsets, sizes, blob = [], [], b""
    for k in pick:
        offsets.append(len(blob)); sizes.append(len(files[k]))
        blob += files[k] + b"\0" * (-len(files[k]) % 64)
    host = np.frombuffer(blob, np.uint8)
    dec = ctx.decode_batch(ctx.to_device(host), offsets, sizes, w, h, lum, chrom).cpu().numpy().reshape(n, h, w, 3)
    ok = all(np.array_equal(dec[j], want[k]) for j, k in enumerate(pick))
    ok_h = True
    if i % 3 == 0:
        back = ctx.decode_host_batch(host, offsets, sizes, w, h, lum, chrom)
        ok_h = all(np.array_equal(back[j], want[k]) for j, k in enumerate(pick))
        rgb = np.stack([kinds[k] for k in pick])
        out, off, size = ctx.encode_host_batch(rgb, lum, chrom)
        ok_h = ok_h and all(out[int(off[j]): int(off[j] + size[j])].tobytes() == files[k] for j, k in enumerate(pick))
    if not (ok and ok_h):
        bad += 1
        print("MISMATCH case", i, (w, h, n, q), "device" if not ok else "", "host batch" if not ok_h else "", flush=True)
    if time.time() - last > 30:
        last = time.time()
        print("case", i, "of", cases, flush=True)
print("%d ragged batches, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
