"""What overlapping the phases of a large batch encode would buy: 1024 1080p q=90 frames through one context in one
call, and split over 2, 3, 4 contexts (own streams) in pieces of `piece` frames that alternate over the contexts."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch

w, h, q, n = 1920, 1080, 90, 1024
lum, chrom = jea.binding.quant_tables(q)
frames = [po.gen_image(w, h, 1 if i % 64 == 63 else 0, 1234 + i) for i in range(16)]
ctxs = [jea.Context(0) for _ in range(4)]
d_rgb = ctxs[0].to_device(np.stack([frames[i % 16] for i in range(n)]).reshape(-1))
image_bytes = 3 * w * h
cap_per = int(jea.binding.max_jpeg_bytes(w, h))


def run(n_ctx, piece):
    pieces = [(first, min(piece, n - first)) for first in range(0, n, piece)]
    bufs = []
    for k, (first, cnt) in enumerate(pieces):
        c = ctxs[k % n_ctx]
        bufs.append(dict(out=torch.empty(cnt * cap_per // 8, dtype=torch.uint8, device=c.tdev), off=torch.zeros(cnt, dtype=torch.int64, device=c.tdev),
                         size=torch.zeros(cnt, dtype=torch.int64, device=c.tdev), status=torch.zeros(16, dtype=torch.int32, device=c.tdev)))
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k, (first, cnt) in enumerate(pieces):
            c = ctxs[k % n_ctx]; b = bufs[k]
            c.encode_batch_async(d_rgb[first * image_bytes:(first + cnt) * image_bytes], cnt, w, h, lum, chrom, b["out"], b["off"], b["size"], b["status"])
        for c in ctxs[:n_ctx]:
            c.synchronize()
        best = min(best, time.perf_counter() - t0)
    total = sum(int(b["size"].sum()) for b in bufs)
    print(f"{n_ctx} contexts, pieces of {piece}: {best * 1e3:.3f} ms  {n * w * h / best / 1e9:.1f} GPix/s  ({total} bytes of files)", flush=True)


run(1, 1024)
for n_ctx in (2, 3, 4):
    for piece in (32, 64, 128):
        run(n_ctx, piece)
run(1, 64)
