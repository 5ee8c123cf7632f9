"""A few encodes of one resident 4K picture on one context (the program the counter passes of tools/pmc_encode.sh run):
mode [encodes] [quality]."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po
import torch
from bench import photo_like_image

mode = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 10; q = int(sys.argv[3]) if len(sys.argv) > 3 else 75
w, h = 3840, 2160
if os.environ.get("JEA_LIB"):
    jea.binding.LIB_PATH = os.path.join(ROOT, "jpeg-encoder_amd", os.environ["JEA_LIB"])
ctx = jea.Context(0)
lum, chrom = jea.binding.quant_tables(q)
rgb = photo_like_image(w, h, 1234) if mode == 2 else po.gen_image(w, h, mode, 1234)
d = ctx.to_device(rgb.reshape(-1))
out = torch.empty(64 << 20, dtype=torch.uint8, device=ctx.tdev)
off = torch.zeros(1, dtype=torch.int64, device=ctx.tdev); size = torch.zeros(1, dtype=torch.int64, device=ctx.tdev)
status = torch.zeros(16, dtype=torch.int32, device=ctx.tdev)
for _ in range(n):
    ctx.encode_batch_async(d, 1, w, h, lum, chrom, out, off, size, status)
ctx.synchronize()
print("file", int(size[0]), "status", int(status[0]))
