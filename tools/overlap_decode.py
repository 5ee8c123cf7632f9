"""What overlapping the phases of a large batch decode would buy: 1024 1080p q=50 files (every 64th frame noise) through
one context in one call, and as pieces that alternate over 2 or 3 contexts, one host thread each (the call blocks)."""
import concurrent.futures
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch

w, h, q, n = 1920, 1080, 50, 1024
lum, chrom = jea.binding.quant_tables(q)
with concurrent.futures.ThreadPoolExecutor(16) as tp:
    files = list(tp.map(lambda i: po.encode(po.gen_image(w, h, 1 if i % 64 == 63 else 0, 1234 + i), q), range(64)))
ctxs = [jea.Context(0) for _ in range(3)]
src = [i % 64 for i in range(n)]
sizes = [len(files[s]) for s in src]
offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
blob = ctxs[0].to_device(np.frombuffer(b"".join(files[s] for s in src), np.uint8))


def run(n_ctx, piece):
    pieces = [(first, min(piece, n - first)) for first in range(0, n, piece)]
    outs = [torch.empty(cnt * w * h * 3, dtype=torch.uint8, device=ctxs[0].tdev) for _, cnt in pieces]

    def work(c):
        for k in range(c, len(pieces), n_ctx):
            first, cnt = pieces[k]
            o = np.ascontiguousarray(offs[first:first + cnt], np.uint64)
            z = np.ascontiguousarray(sizes[first:first + cnt], np.uint64)
            rc = ctxs[c].lib.jea_decode_batch(ctxs[c].handle, blob.data_ptr(), o.ctypes.data, z.ctypes.data, cnt, w, h,
                                              lum.ctypes.data, chrom.ctypes.data, outs[k].data_ptr())
            assert rc == 0, rc

    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        ts = [threading.Thread(target=work, args=(c,)) for c in range(n_ctx)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{n_ctx} contexts, pieces of {piece}: {best * 1e3:.3f} ms  {n * w * h / best / 1e9:.1f} GPix/s", flush=True)


run(1, 1024)
run(1, 128)
for n_ctx in (2, 3):
    for piece in (64, 128, 256):
        run(n_ctx, piece)
