"""Encode of resident 4K pictures: phase times on one context (events between the kernels), one-context latency
and three-context throughput, for smooth, photo-like and noise content.  JEA_LIB names a build made by
tools/build_variant.sh (probe builds may write wrong files on purpose: the comparison with the oracle is only
printed)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch
from bench import photo_like_image

w, h, q = 3840, 2160, 75
modes = [int(a) for a in sys.argv[1:]] or [0, 2, 1]
if os.environ.get("JEA_LIB"):
    jea.binding.LIB_PATH = os.path.join(ROOT, "jpeg-encoder_amd", os.environ["JEA_LIB"])
ctxs = [jea.Context(0) for _ in range(3)]
lum, chrom = jea.binding.quant_tables(q)
for mode in modes:
    n_pool = 4
    imgs = [photo_like_image(w, h, 1234 + i) if mode == 2 else po.gen_image(w, h, mode, 1234 + i) for i in range(n_pool)]
    d = [ctxs[0].to_device(im.reshape(-1)) for im in imgs]
    outs = [dict(out=torch.empty(64 << 20, dtype=torch.uint8, device=c.tdev), off=torch.zeros(1, dtype=torch.int64, device=c.tdev),
                 size=torch.zeros(1, dtype=torch.int64, device=c.tdev), status=torch.zeros(16, dtype=torch.int32, device=c.tdev))
            for c in ctxs]

    def enc(c, i):
        o = outs[c]
        ctxs[c].encode_batch_async(d[i % n_pool], 1, w, h, lum, chrom, o["out"], o["off"], o["size"], o["status"])

    enc(0, 0)
    ctxs[0].synchronize()
    size = int(outs[0]["size"][0])
    same = outs[0]["out"][:size].cpu().numpy().tobytes() == po.encode(imgs[0], q)
    # clocks up
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for i in range(20):
            enc(0, i)
        ctxs[0].synchronize()
    ctxs[0].set_profiling(True)
    acc = np.zeros(3)
    for i in range(45):
        enc(0, i)
        if i >= 5:
            acc += np.array(ctxs[0].last_phase_ms())
    ctxs[0].set_profiling(False)
    ph = acc / 40 * 1e3
    t0 = time.perf_counter()
    for i in range(400):
        enc(0, i)
    ctxs[0].synchronize()
    one = (time.perf_counter() - t0) / 400 * 1e6
    for i in range(60):
        enc(i % 3, i)
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for i in range(900):
        enc(i % 3, i)
    for c in ctxs:
        c.synchronize()
    three = (time.perf_counter() - t0) / 900 * 1e6
    print("mode %d file %8d B oracle-equal %-5s | phases us: forward %.2f bits %.2f pack %.2f | one context %.2f us | "
          "three contexts %.2f us/image = %.1f GPix/s" % (mode, size, same, ph[0], ph[1], ph[2], one, three, w * h / three / 1e3), flush=True)
