"""Forward-kernel launch time against the number of images per launch (fixed cost vs per-image cost)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import pyoracle as po
pkg = importlib.import_module("jpeg-encoder_amd")
b = pkg.binding
if len(sys.argv) > 1:
    b.LIB_PATH = os.path.join(ROOT, "jpeg-encoder_amd", sys.argv[1])
ctx = pkg.Context(0)
W, H = 3840, 2160
lum, chrom = b.quant_tables(75)
img = po.gen_image(W, H, 0, 1234)
nmax = 8
d_rgb = ctx.to_device(np.stack([img] * nmax).reshape(-1))
d_coefs = torch.empty(nmax * b.n_mcu(W, H) * 192, dtype=torch.int16, device=ctx.tdev)
ext = torch.cuda.ExternalStream(ctx.stream_handle, device=ctx.tdev)
for h in (2160, 1080, 544, 272, 136):   # fractions of one image too
    for n in ([1] if h != 2160 else [1, 2, 3, 4, 8]):
        with torch.cuda.stream(ext):
            for _ in range(5): ctx.fdct_quant_async(d_rgb, n, W, h, lum, chrom, d_coefs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): ctx.fdct_quant_async(d_rgb, n, W, h, lum, chrom, d_coefs)
            e1.record()
        ctx.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        waves = n * (W // 8) * ((h + 7) // 8) * 3 / 64
        print("h=%4d n=%d: %7.2f us  (%.2f waves per SIMD, %.2f us per wave-per-SIMD)" % (h, n, us, waves / 1024, us / (waves / 1024)))
