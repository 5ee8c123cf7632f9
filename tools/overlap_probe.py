"""How well do kernels of independent contexts overlap?  Forward kernel alone on 1 and on several
contexts, against the whole encode on the same contexts."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch

W, H, Q = 3840, 2160, 75
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctxs = [jea.Context(0) for _ in range(n_ctx)]
lum, chrom = jea.binding.quant_tables(Q)
img = po.gen_image(W, H, 0, 1234)
d_rgb = ctxs[0].to_device(img.reshape(-1))
dev = ctxs[0].tdev
n_mcu = jea.binding.n_mcu(W, H)
bufs = [dict(coefs=torch.empty(n_mcu * 192, dtype=torch.int16, device=dev), masks=torch.empty(n_mcu * 3, dtype=torch.uint8, device=dev),
             out=torch.empty(3 * W * H // 2, dtype=torch.uint8, device=dev), off=torch.zeros(1, dtype=torch.int64, device=dev),
             size=torch.zeros(1, dtype=torch.int64, device=dev), st=torch.zeros(16, dtype=torch.int32, device=dev)) for _ in ctxs]

def run(fn, k, steps=300):
    for i in range(20): fn(i % k)
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): fn(i % k)
    for c in ctxs: c.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6

def fdct(i):
    ctxs[i].fdct_quant_planar_async(d_rgb, 1, W, H, lum, chrom, bufs[i]["coefs"], bufs[i]["masks"])
def enc(i):
    b = bufs[i]
    ctxs[i].encode_batch_async(d_rgb, 1, W, H, lum, chrom, b["out"], b["off"], b["size"], b["st"])

for k in range(1, n_ctx + 1):
    print(f"contexts {k}: forward kernel alone {run(fdct, k):6.1f} us/image   whole encode {run(enc, k):6.1f} us/image")
