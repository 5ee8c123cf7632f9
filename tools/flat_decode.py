"""Decode of flat and periodic images: their streams repeat with a short period, so a decoder that starts in
the wrong state can stay out of phase for ever and the synchronisation has to travel group by group."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch
ctx = jea.Context(0)
q = 75
lum, chrom = jea.binding.quant_tables(q)
for (w, h, kind) in [(512, 512, "flat"), (3840, 2160, "flat"), (3840, 2160, "stripes"), (1024, 1024, "checker")]:
    rgb = np.full((h, w, 3), 128, np.uint8)
    if kind == "stripes":
        rgb[:, ::16] = 200
    if kind == "checker":
        rgb[::8, ::8] = 255
    jpeg = po.encode(rgb, q)
    want = po.decode(jpeg, w, h, q)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got = ctx.decode_host(jpeg, w, h, lum, chrom)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(kind, w, h, "file", len(jpeg), "ok", np.array_equal(got, want), "%.2f ms" % (1e3 * (t1 - t0)), flush=True)
