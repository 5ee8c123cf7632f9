"""How much of a large decode batch's time is hidden when its pieces run on several contexts at once (one host thread
each): the entropy kernels of one piece beside the inverse transform of another.  Usage: probe_decode_lanes.py [frames]"""
import concurrent.futures
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
import torch  # noqa: E402

pkg = importlib.import_module("jpeg-encoder_amd")
binding = pkg.binding
FW, FH, Q = 1920, 1080, 50
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with concurrent.futures.ThreadPoolExecutor(16) as tp:
    frames = np.stack(list(tp.map(lambda i: po.gen_image(FW, FH, 1 if i % 64 == 63 else 0, 1234 + i), range(n))))
lum, chrom = binding.quant_tables(Q)
ctx = pkg.Context(0)
d_rgb = ctx.to_device(frames.reshape(-1))
d_jpeg, offs, sizes = ctx.encode_batch(d_rgb, n, FW, FH, lum, chrom, capacity=n * (1 << 19) + (n // 64 + 1) * (3 << 20))
del d_rgb
offs = np.ascontiguousarray(offs, np.uint64)
sizes = np.ascontiguousarray(sizes, np.uint64)
d_out = ctx.empty(n * 3 * FW * FH, torch.uint8)


def decode(c, first, count):
    c._check(c.lib.jea_decode_batch(c.handle, d_jpeg.data_ptr(), offs[first:].ctypes.data, sizes[first:].ctypes.data, count, FW, FH,
                                    lum.ctypes.data, chrom.ctypes.data, d_out.data_ptr() + first * 3 * FW * FH))


def timed(fn, reps=6):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("one call: %.3f ms" % timed(lambda: decode(ctx, 0, n)), flush=True)
want = d_out.clone()
ctxs = [ctx] + [pkg.Context(0) for _ in range(3)]
for piece in (512, 256, 128, 64):
    for lanes in (1, 2, 3, 4):
        def run():
            nxt = [0]
            lock = threading.Lock()

            def worker(c):
                while True:
                    with lock:
                        first = nxt[0]
                        nxt[0] += piece
                    if first >= n:
                        return
                    decode(c, first, min(piece, n - first))
            ts = [threading.Thread(target=worker, args=(ctxs[i],)) for i in range(lanes)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        d_out.zero_()
        ms = timed(run)
        print("pieces of %4d on %d contexts: %.3f ms  %s" % (piece, lanes, ms, "same pixels" if torch.equal(d_out, want) else "DIFFERENT"), flush=True)

# two lanes out of phase: the second starts `delay` ms after the first
for piece in (512, 256):
    for delay in (0.0, 0.2, 0.4, 0.6, 0.8):
        def run():
            def worker(c, lane):
                if lane:
                    t0 = time.perf_counter()
                    while (time.perf_counter() - t0) * 1e3 < delay:
                        pass
                for first in range(lane * piece, n, 2 * piece):
                    decode(c, first, min(piece, n - first))
            ts = [threading.Thread(target=worker, args=(ctxs[i], i)) for i in range(2)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        ms = timed(run)
        print("two lanes, pieces of %d, second lane %.1f ms late: %.3f ms" % (piece, delay, ms), flush=True)
