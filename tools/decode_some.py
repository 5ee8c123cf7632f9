"""A few decodes of a batch of resident files on one context (the program the counter passes of tools/pmc_decode.sh run):
w h quality mode n [calls]."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch

w, h, q, mode, n = (int(a) for a in sys.argv[1:6])
calls = int(sys.argv[6]) if len(sys.argv) > 6 else 4
if os.environ.get("JEA_LIB"):
    jea.binding.LIB_PATH = os.path.join(ROOT, "jpeg-encoder_amd", os.environ["JEA_LIB"])
ctx = jea.Context(0)
lum, chrom = jea.binding.quant_tables(q)
if mode == 2:
    from bench import photo_like_image
    rgb = photo_like_image(w, h, 1234)
else:
    rgb = po.gen_image(w, h, mode, 1234)
d_rgb = ctx.to_device(np.tile(rgb.reshape(-1), n))
out, off, size = ctx.encode_batch(d_rgb, n, w, h, lum, chrom, capacity=n * int(jea.binding.max_jpeg_bytes(w, h)) if w * h < 1 << 22 else n * (64 << 20))
offs = [int(o) for o in off]; sizes = [int(s) for s in size]
for _ in range(calls):
    dec = ctx.decode_batch(out, offs, sizes, w, h, lum, chrom)
torch.cuda.synchronize()
print("decoded", n, "files of", sizes[0], "bytes")
