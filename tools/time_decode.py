"""Time the GPU decode (entropy decode + inverse transform) of device-resident files."""
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
if os.environ.get("JEA_LIB"):   # probe builds may decode nonsense on purpose: only their kernel times matter
    _decode = ctx.decode_batch
    def _tolerant(*a):
        try:
            return _decode(*a)
        except jea.JeaError:
            return None
    ctx.decode_batch = _tolerant
lum, chrom = jea.binding.quant_tables(q)
if mode == 2:
    from bench import photo_like_image
    rgb = photo_like_image(w, h, 1234)
else:
    rgb = po.gen_image(w, h, mode, 1234)
d_rgb = ctx.to_device(np.tile(rgb.reshape(-1), n))
out, off, size = ctx.encode_batch(d_rgb, n, w, h, lum, chrom, capacity=n * int(jea.binding.max_jpeg_bytes(w, h)) if w * h < 1 << 22 else n * (64 << 20))
offs = [int(o) for o in off]; sizes = [int(s) for s in size]
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec = ctx.decode_batch(out, offs, sizes, w, h, lum, chrom)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"decode {n} x {w}x{h} q{q} mode{mode}: {1e3*(t1-t0):.3f} ms  {n*w*h/(t1-t0)/1e6:.0f} MPix/s  file {sizes[0]} B")
reps = 200 if n * w * h <= 1 << 24 else 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(reps):
    dec = ctx.decode_batch(out, offs, sizes, w, h, lum, chrom)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"{reps} calls back to back: {1e3*(t1-t0)/reps:.3f} ms per call  {reps*n*w*h/(t1-t0)/1e6:.0f} MPix/s")
ctx.set_profiling(True)
acc = None
for it in range(20):
    dec = ctx.decode_batch(out, offs, sizes, w, h, lum, chrom)
    ph = ctx.last_phase_ms()
    acc = ph if acc is None else [a + b for a, b in zip(acc, ph)]
ctx.set_profiling(False)
print("phases (us): unstuff %.1f  sync rounds %.1f  scan+write %.1f  inverse transform %.1f" % tuple(1e3 * a / 20 for a in acc[:4]))
want = po.decode(out.cpu().numpy()[offs[0]:offs[0] + sizes[0]].tobytes(), w, h, q)
print("matches oracle:", dec is not None and np.array_equal(dec.cpu().numpy().reshape(n, h, w, 3)[n - 1], want))
