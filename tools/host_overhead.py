"""Host-side cost of one jea_encode_batch_async call: tiny images, so that the GPU is never the limit."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po, torch
ctxs = [jea.Context(0) for _ in range(3)]
w, h = 64, 64
lum, chrom = jea.binding.quant_tables(75)
d_rgb = ctxs[0].to_device(po.gen_image(w, h, 0, 1).reshape(-1))
outs = [dict(out=torch.empty(1 << 20, dtype=torch.uint8, device="cuda"), off=torch.zeros(1, dtype=torch.int64, device="cuda"),
             size=torch.zeros(1, dtype=torch.int64, device="cuda"), st=torch.zeros(16, dtype=torch.int32, device="cuda")) for _ in ctxs]
def run(n):
    for i in range(n):
        c, o = ctxs[i % 3], outs[i % 3]
        c.encode_batch_async(d_rgb, 1, w, h, lum, chrom, o["out"], o["off"], o["size"], o["st"])
run(50)
for c in ctxs: c.synchronize()
t0 = time.perf_counter(); run(2000); t1 = time.perf_counter()
for c in ctxs: c.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us per call; with final sync %.1f us per call" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
