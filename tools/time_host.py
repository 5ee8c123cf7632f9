"""Host-buffer entry points: wall time of encode_host / decode_host for one 4K image (includes PCIe both ways)."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
ctx = jea.Context(0)
w, h, q = 3840, 2160, 75
lum, chrom = jea.binding.quant_tables(q)
rgb = po.gen_image(w, h, 0, 1234)
for it in range(4):
    t0 = time.perf_counter(); jpeg = ctx.encode_host(rgb, lum, chrom); t1 = time.perf_counter()
    dec = ctx.decode_host(jpeg, w, h, lum, chrom); t2 = time.perf_counter()
    print("encode_host %.2f ms (%.1f GPix/s, %.1f GB/s in)  decode_host %.2f ms" % (1e3 * (t1 - t0), w * h / (t1 - t0) / 1e9, 3 * w * h / (t1 - t0) / 1e9, 1e3 * (t2 - t1)), flush=True)
