"""Per-image Huffman tables: encode time of one resident 4K picture on one context, standard tables beside it."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch
w, h, q = 3840, 2160, 75
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if os.environ.get("JEA_LIB"):   # a build made by tools/build_variant.sh
    jea.binding.LIB_PATH = os.path.join(ROOT, "jpeg-encoder_amd", os.environ["JEA_LIB"])
ctx = jea.Context(0)
lum, chrom = jea.binding.quant_tables(q)
rgb = po.gen_image(w, h, mode, 1234)
d_rgb = ctx.to_device(rgb.reshape(-1))
out = torch.empty(64 << 20, dtype=torch.uint8, device=ctx.tdev)
off = torch.zeros(1, dtype=torch.int64, device=ctx.tdev); size = torch.zeros(1, dtype=torch.int64, device=ctx.tdev)
status = torch.zeros(16, dtype=torch.int32, device=ctx.tdev)
for on in (False, True):
    ctx.set_huffman_optimised(on)
    for _ in range(20):
        ctx.encode_batch_async(d_rgb, 1, w, h, lum, chrom, out, off, size, status)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        ctx.encode_batch_async(d_rgb, 1, w, h, lum, chrom, out, off, size, status)
    ctx.synchronize()
    t = (time.perf_counter() - t0) / 300
    print("per-image tables" if on else "Annex K tables  ", "%.1f us per image, file %d bytes" % (t * 1e6, int(size[0])))
