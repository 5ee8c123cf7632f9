"""Full-size runs of the batch configurations on one GPU: 1024 x 1920x1080 encode at q=90 and the
decode of 1024 1080p q=50 files.  Prints wall times of device-resident calls and checks samples against the oracle."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import pyoracle as po
pkg = importlib.import_module("jpeg-encoder_amd")
ctx = pkg.Context(0)
b = pkg.binding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w, h, distinct = 1920, 1080, 8

def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best

imgs = [po.gen_image(w, h, 0, 100 + i) for i in range(distinct)]
d_rgb = ctx.to_device(np.concatenate([imgs[i % distinct].reshape(-1) for i in range(n)]))
print("input resident: %.2f GB" % (d_rgb.numel() / 1e9), flush=True)
for q in (90, 50):
    lum, chrom = b.quant_tables(q)
    cap = n * (1 << 20)
    out, off, size = ctx.encode_batch(d_rgb, n, w, h, lum, chrom, capacity=cap)
    for i in (0, 5, n - 1):
        f = out[int(off[i]): int(off[i]) + int(size[i])].cpu().numpy().tobytes()
        assert f == po.encode(imgs[i % distinct], q), (q, i)
    dt = timed(lambda: ctx.encode_batch(d_rgb, n, w, h, lum, chrom, capacity=cap, d_out=out))
    print("encode %d x %dx%d q=%d: %.2f ms = %.1f GPix/s (files %.0f KB avg, byte-identical samples)" %
          (n, w, h, q, dt * 1e3, n * w * h / dt / 1e9, float(np.mean(size)) / 1e3), flush=True)
offs, sizes = [int(o) for o in off], [int(s) for s in size]
dec = ctx.decode_batch(out, offs, sizes, w, h, lum, chrom)
for i in (0, 5, n - 1):
    f = out[offs[i]: offs[i] + sizes[i]].cpu().numpy().tobytes()
    got = dec[i * 3 * w * h: (i + 1) * 3 * w * h].cpu().numpy().reshape(h, w, 3)
    assert np.array_equal(got, po.decode(f, w, h, 50)), i
dt = timed(lambda: ctx.decode_batch(out, offs, sizes, w, h, lum, chrom))
print("decode %d x %dx%d q=50: %.2f ms = %.1f GPix/s (bit-identical samples)" % (n, w, h, dt * 1e3, n * w * h / dt / 1e9), flush=True)
print("peak device memory seen by torch: %.1f GB; free/total %s" % (torch.cuda.max_memory_allocated() / 1e9, [round(x / 1e9, 1) for x in torch.cuda.mem_get_info()]))
