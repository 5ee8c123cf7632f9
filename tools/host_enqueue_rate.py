"""Is the single-picture encode loop bound by the host?  Steps of the headline configuration (three contexts in turn,
3840x2160, q=75) are enqueued without waiting: time for the host to enqueue them, time until the GPU has finished,
and the same with the call made straight through ctypes with its arguments prepared once."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
import torch  # noqa: E402

jea = importlib.import_module("jpeg-encoder_amd")
W, H, Q = 3840, 2160, 75
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = 3000
lum, chrom = jea.binding.quant_tables(Q)
ctxs = [jea.Context(0) for _ in range(n_ctx)]
imgs = [ctxs[0].to_device(po.gen_image(W, H, 0, 1234 + i).reshape(-1)) for i in range(16)]
dev = torch.device("cuda", 0)
outs = [dict(out=torch.empty(4 << 20, dtype=torch.uint8, device=dev), offsets=torch.zeros(1, dtype=torch.int64, device=dev),
             sizes=torch.zeros(1, dtype=torch.int64, device=dev), status=torch.zeros(16, dtype=torch.int32, device=dev)) for _ in ctxs]


def run(step):
    for i in range(300):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6


def through_wrapper(i):
    o = outs[i % n_ctx]
    ctxs[i % n_ctx].encode_batch_async(imgs[i % 16], 1, W, H, lum, chrom, o["out"], o["offsets"], o["sizes"], o["status"])


prepared = []
for i in range(16 * n_ctx):
    c, o = ctxs[i % n_ctx], outs[i % n_ctx]
    prepared.append((c.lib.jea_encode_batch_async, (c.handle, imgs[i % 16].data_ptr(), 1, W, H, lum.ctypes.data, chrom.ctypes.data,
                                                    o["out"].data_ptr(), o["out"].numel(), o["offsets"].data_ptr(), o["sizes"].data_ptr(),
                                                    o["status"].data_ptr())))


def straight(i):
    f, a = prepared[i % len(prepared)]
    f(*a)


for name, fn in (("through the Python wrapper", through_wrapper), ("ctypes call, arguments prepared", straight)):
    host, total = run(fn)
    print("%d contexts, %s: host %.1f us per step to enqueue, %.1f us per step until done (%.1f GPix/s)" % (
        n_ctx, name, host, total, W * H / total / 1e3), flush=True)
