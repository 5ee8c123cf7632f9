"""Time the GPU encode of a batch of device-resident pictures: w h quality mode n."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch

w, h, q, mode, n = (int(a) for a in sys.argv[1:6])
if os.environ.get("JEA_LIB"):   # a build made by tools/build_variant.sh
    jea.binding.LIB_PATH = os.path.join(ROOT, "jpeg-encoder_amd", os.environ["JEA_LIB"])
ctx = jea.Context(0)
lum, chrom = jea.binding.quant_tables(q)
if mode == 2:
    from bench import photo_like_image
    rgb = photo_like_image(w, h, 1234)
else:
    rgb = po.gen_image(w, h, mode, 1234)
d_rgb = ctx.to_device(np.tile(rgb.reshape(-1), n))
cap = n * int(jea.binding.max_jpeg_bytes(w, h)) if w * h < 1 << 22 else n * (64 << 20)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, off, size = ctx.encode_batch(d_rgb, n, w, h, lum, chrom, capacity=cap)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"encode {n} x {w}x{h} q{q} mode{mode}: {1e3*(t1-t0):.3f} ms  {n*w*h/(t1-t0)/1e6:.0f} MPix/s  file {int(size[0])} B")
o0, s0 = int(off[n - 1]), int(size[n - 1])
print("matches oracle:", out.cpu().numpy()[o0:o0 + s0].tobytes() == po.encode(rgb, q))
