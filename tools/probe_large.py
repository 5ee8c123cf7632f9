"""Ad-hoc probe on the GPU box: 16384x16384 encode against the recorded reference hash, and decode timings."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import pyoracle as po
pkg = importlib.import_module("jpeg-encoder_amd")
ctx = pkg.Context(0)
b = pkg.binding

def t(f, n=3):
    f(); ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    ctx.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

if "big" in sys.argv:
    w = h = 16384
    lum, chrom = b.quant_tables(75)
    t0 = time.time(); rgb = po.gen_image(w, h, 0, 1234); print("gen %.1fs" % (time.time() - t0), flush=True)
    d = ctx.to_device(rgb.reshape(-1))
    cap = 64 << 20
    out, off, size = ctx.encode_batch(d, 1, w, h, lum, chrom, capacity=cap)
    jpeg = out[: int(size[0])].cpu().numpy().tobytes()
    print("16384^2 bytes", len(jpeg), po.fnv(jpeg), "want 5970b2252be8b37d", flush=True)
    dt = t(lambda: ctx.encode_batch(d, 1, w, h, lum, chrom, capacity=cap, d_out=out), 3)
    print("16384^2 encode %.2f ms = %.1f GPix/s" % (dt * 1e3, w * h / dt / 1e9), flush=True)
    import json
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_hashes.json")))
    rows = rows if isinstance(rows, list) else rows["rows"]
    want_rgb = [r for r in rows if r["w"] == 16384][0]["rgb_fnv"]
    dec = ctx.decode_batch(out, [0], [len(jpeg)], w, h, lum, chrom)
    print("16384^2 decoded rgb hash", po.fnv(dec.cpu().numpy()), "want", want_rgb, flush=True)
    dt = t(lambda: ctx.decode_batch(out, [0], [len(jpeg)], w, h, lum, chrom), 3)
    print("16384^2 decode %.2f ms = %.1f GPix/s" % (dt * 1e3, w * h / dt / 1e9), flush=True)
    del dec
    del d, out

for (w, h, q, mode, n) in [(3840, 2160, 75, 0, 1), (1920, 1080, 50, 0, 64), (1920, 1080, 50, 1, 16)]:
    lum, chrom = b.quant_tables(q)
    imgs = [po.gen_image(w, h, mode, 10 + i) for i in range(min(n, 4))]
    files = [po.encode(im, q) for im in imgs]
    blob, offs, sizes = b"", [], []
    for i in range(n):
        f = files[i % len(files)]
        offs.append(len(blob)); sizes.append(len(f)); blob += f + b"\0" * (-len(f) % 64)
    dj = ctx.to_device(np.frombuffer(blob, np.uint8))
    rgb = ctx.decode_batch(dj, offs, sizes, w, h, lum, chrom)
    ok = np.array_equal(rgb[: 3 * w * h].cpu().numpy().reshape(h, w, 3), po.decode(files[0], w, h, q))
    dt = t(lambda: ctx.decode_batch(dj, offs, sizes, w, h, lum, chrom), 2)
    print("decode %dx%d q%d mode%d n=%d ok=%s: %.2f ms = %.2f GPix/s (avg file %d B)" % (w, h, q, mode, n, ok, dt * 1e3, n * w * h / dt / 1e9, sum(sizes) // n), flush=True)
