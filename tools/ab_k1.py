"""A/B of fused-forward-kernel builds in one process: average launch time over interleaved rounds."""
import ctypes, importlib, os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import pyoracle as po
pkg = importlib.import_module("jpeg-encoder_amd")
b = pkg.binding
libs = sorted(glob.glob(os.path.join(ROOT, "jpeg-encoder_amd", "libjpegenc_amd_*.so")))
W, H = 3840, 2160
lum, chrom = b.quant_tables(75)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
imgs = np.stack([po.gen_image(W, H, 0, 1234 + i) for i in range(batch)])
ctxs = {}
for path in libs:
    b._lib = None; b.LIB_PATH = path
    ctxs[os.path.basename(path)] = pkg.Context(0)
res = {k: [] for k in ctxs}
for k, ctx in ctxs.items():
    ctx.d_rgb = ctx.to_device(imgs.reshape(-1))
    ctx.d_coefs = torch.empty(batch * b.n_mcu(W, H) * 192, dtype=torch.int16, device=ctx.tdev)
    ctx.ext = torch.cuda.ExternalStream(ctx.stream_handle, device=ctx.tdev)
for rnd in range(6):
    for k, ctx in ctxs.items():
        with torch.cuda.stream(ctx.ext):
            for _ in range(3): ctx.fdct_quant_async(ctx.d_rgb, batch, W, H, lum, chrom, ctx.d_coefs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): ctx.fdct_quant_async(ctx.d_rgb, batch, W, H, lum, chrom, ctx.d_coefs)
            e1.record()
        ctx.synchronize()
        res[k].append(e0.elapsed_time(e1) / 30 * 1e3 / batch)
ref = None
for k, v in res.items():
    print("%-36s median %.2f us/image  min %.2f" % (k, sorted(v)[len(v) // 2], min(v)))
