"""jea_decode_host_batch on pictures in host memory, for a trace of its copies and kernels:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/trace_decode_host.py [pinned]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

jea = importlib.import_module("jpeg-encoder_amd")
if os.environ.get("JEA_LIB"):   # a build made for an experiment
    jea.binding.LIB_PATH = os.path.join(ROOT, "jpeg-encoder_amd", os.environ["JEA_LIB"])
pinned = len(sys.argv) > 1 and sys.argv[1] == "pinned"
w, h, q, n = 3840, 2160, 75, 32
ctx = jea.Context(0)
lum, chrom = jea.binding.quant_tables(q)
imgs = [po.gen_image(w, h, 0, 1234 + i) for i in range(4)]
per = 3 * w * h
rgb = np.empty((n, h, w, 3), np.uint8)
for i in range(n):
    rgb[i] = imgs[i % 4]
out = np.empty(n * (per // 8), np.uint8)
_, off, size = ctx.encode_host_batch(rgb, lum, chrom, out=out)
back = ctx.host_array(n * per) if pinned else np.ones(n * per, np.uint8)
ctx.decode_host_batch(out, off[:4], size[:4], w, h, lum, chrom, out=back)
for _ in range(3):
    t0 = time.perf_counter()
    ctx.decode_host_batch(out, off, size, w, h, lum, chrom, out=back)
    dt = time.perf_counter() - t0
    print("decode_host_batch %s: %.2f ms, %.1f GB/s out" % ("pinned" if pinned else "pageable", dt * 1e3, n * per / dt / 1e9), flush=True)
