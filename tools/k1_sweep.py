"""Forward-kernel launch time against the number of workgroups per launch, for every build present
(jpeg-encoder_amd/libjpegenc_amd.so and libjpegenc_amd_*.so): parts of one 3840-wide image, then several images."""
import glob
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch
import pyoracle as po

pkg = importlib.import_module("jpeg-encoder_amd")
b = pkg.binding
libs = [os.path.join(ROOT, "jpeg-encoder_amd", "libjpegenc_amd.so")] + sorted(glob.glob(os.path.join(ROOT, "jpeg-encoder_amd", "libjpegenc_amd_*.so")))
W, H, NB = 3840, 2160, 8
lum, chrom = b.quant_tables(75)
imgs = np.stack([po.gen_image(W, H, 0, 1234 + i) for i in range(2 * NB)])
cases = [(136, 1), (272, 1), (544, 1), (1080, 1), (1360, 1), (1632, 1), (1896, 1), (2160, 1), (2160, 2), (2160, 3), (2160, 4), (2160, 8)]
rows = {}
d_rgb = None
for path in libs:
    b._lib = None
    b.LIB_PATH = path
    ctx = pkg.Context(0)
    if d_rgb is None:
        d_rgb = ctx.to_device(imgs.reshape(-1))
    d_coefs = torch.zeros(NB * b.n_mcu(W, H) * 192, dtype=torch.int16, device=ctx.tdev)
    d_masks = torch.zeros(NB * b.n_mcu(W, H) * 3, dtype=torch.uint8, device=ctx.tdev)
    ext = torch.cuda.ExternalStream(ctx.stream_handle, device=ctx.tdev)
    out = []
    for h, n in cases:
        best = []
        for rnd in range(3):
            with torch.cuda.stream(ext):
                for i in range(3):
                    ctx.fdct_quant_planar_async(d_rgb[(i % 2) * NB * 3 * W * H:], n, W, h, lum, chrom, d_coefs, d_masks)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(30):
                    ctx.fdct_quant_planar_async(d_rgb[(i % 2) * NB * 3 * W * H:], n, W, h, lum, chrom, d_coefs, d_masks)
                e1.record()
            ctx.synchronize()
            best.append(e0.elapsed_time(e1) / 30 * 1e3)
        out.append(sorted(best)[1])
    rows[os.path.basename(path)] = out
    ctx.close()
print("%-28s" % "workgroups (waves per SIMD)" + "".join("%9d" % (n * (W // 8) * ((h + 7) // 8) // 64) for h, n in cases))
print("%-28s" % "" + "".join("%9.2f" % (n * (W // 8) * ((h + 7) // 8) * 3 / 64 / 1024) for h, n in cases))
for k, v in rows.items():
    print("%-28s" % k + "".join("%9.2f" % t for t in v))
