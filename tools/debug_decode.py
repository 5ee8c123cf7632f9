"""Compare the GPU entropy decoder with the oracle on one generated image and print where they differ."""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
jea = importlib.import_module("jpeg-encoder_amd")
import pyoracle as po

w, h, q, mode = (int(a) for a in sys.argv[1:5])
ctx = jea.Context(0)
lum, chrom = jea.binding.quant_tables(q)
rgb = po.gen_image(w, h, mode, 1234)
coefs = po.image_to_coefs(rgb, lum, chrom)
scan, _ = po.coefs_to_scan(coefs)
seg = scan + b"\xff\xd9"
want, used, nlen = po.scan_to_coefs(seg, coefs.shape[0])
got = ctx.entropy_decode(ctx.to_device(np.frombuffer(seg, np.uint8)), [0], [len(seg)], coefs.shape[0])
got = got.cpu().numpy().reshape(-1, 64)
want = want.reshape(-1, 64)
bad = np.argwhere(got != want)
print("scan bytes", len(seg), "blocks", want.shape[0], "mismatches", len(bad))
if len(bad):
    b = bad[0][0]
    print("first bad block", b, "positions", bad[bad[:, 0] == b][:, 1])
    print("want", want[b]); print("got ", got[b])
    print("bad blocks:", np.unique(bad[:, 0])[:40])
d = (got.astype(int) - want.astype(int))[:, 0].reshape(-1, 3)
chg = np.argwhere(np.diff(d, axis=0, prepend=0) != 0)
print("DC error changes at (mcu, ch, new error):", [(int(m), int(c), int(d[m, c])) for m, c in chg[:30]])
wy = want[0::3, 0].astype(int); gy = got[0::3, 0].astype(int)
wd = np.diff(wy, prepend=0); gd = np.diff(gy, prepend=0)
for m in np.argwhere(wd != gd)[:12, 0]:
    print("mcu", m, "want diff", wd[m], bin(wd[m] & 0xfff), "got diff", gd[m], bin(gd[m] & 0xfff))
# bit position of every block start in the unstuffed stream
lens = ctx.block_bit_lengths(ctx.to_device(coefs.reshape(-1)), 1, coefs.shape[0]).cpu().numpy().view(np.uint16).astype(np.int64).reshape(-1)
starts = np.concatenate([[0], np.cumsum(lens)])
for m in np.argwhere(wd != gd)[:12, 0]:
    s = starts[3 * m]
    print("mcu", m, "Y block starts at bit", s, "= sub", s // 512, "+", s % 512, " prev block start", starts[3*m-1] // 512, starts[3*m-1] % 512, "len", lens[3*m])
