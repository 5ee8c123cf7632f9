import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po, torch
ctx = jea.Context(0)
for (w, h) in [(3840, 2160), (3841, 2160), (3836, 2157), (1366, 768)]:
    lum, chrom = jea.binding.quant_tables(75)
    img = po.gen_image(w, h, 0, 1)
    d = ctx.to_device(img.reshape(-1))
    dev = ctx.tdev
    cap = int(jea.binding.max_jpeg_bytes(w, h))
    out = torch.empty(cap, dtype=torch.uint8, device=dev); off = torch.zeros(1, dtype=torch.int64, device=dev); size = torch.zeros(1, dtype=torch.int64, device=dev); st = torch.zeros(16, dtype=torch.int32, device=dev)
    for _ in range(20): ctx.encode_batch_async(d, 1, w, h, lum, chrom, out, off, size, st)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(300): ctx.encode_batch_async(d, 1, w, h, lum, chrom, out, off, size, st)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 300
    n = int(size[0].item())
    ok = out[:n].cpu().numpy().tobytes() == po.encode(img, 75)
    jpeg = out[:n].clone()
    for _ in range(3): dec = ctx.decode_batch(jpeg, [0], [n], w, h, lum, chrom)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dec = ctx.decode_batch(jpeg, [0], [n], w, h, lum, chrom)
    torch.cuda.synchronize(); dd = (time.perf_counter() - t0) / 20
    print(f"{w}x{h}: encode {dt*1e6:7.1f} us = {w*h/dt/1e9:6.1f} GPix/s ok={ok}; decode {dd*1e6:7.1f} us")
