"""The benchmark configurations at their full sizes, on one GPU, against the recorded reference hashes and the oracle.

  C3  1024 x 1920x1080 at q=90 in one encode call
  C4  one 16384x16384 image at q=75: on one context, and as 8 stripes of 2048 rows on 8 contexts
  C5  1024 x 1920x1080 JPEGs at q=50 in one decode call
(C1 and C2, 512x512 and 3840x2160, are rows of test_golden_hashes.)  Bar: bit-exact.
"""
import concurrent.futures
import importlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_FRAMES, FW, FH = 1024, 1920, 1080
THREADS = 16   # the oracle is plain C behind ctypes (the GIL is released): one image per thread


def _golden_row(w, h, q, mode):
    rows = json.load(open(os.path.join(GOLDEN, "reference_hashes.json")))["rows"]
    return next(r for r in rows if (r["w"], r["h"], r["q"], r["mode"]) == (w, h, q, mode))


@pytest.fixture(scope="module")
def gigapixel(po):
    return po.gen_image(16384, 16384, 0, 1234)


def test_c4_gigapixel_on_one_context(ctx, po, jea, gigapixel):
    import torch

    w = h = 16384
    r = _golden_row(w, h, 75, 0)
    lum, chrom = jea.binding.quant_tables(75)
    d_rgb = ctx.to_device(gigapixel.reshape(-1))
    out, off, size = ctx.encode_batch(d_rgb, 1, w, h, lum, chrom, capacity=64 << 20)
    jpeg = out[: int(size[0])].cpu().numpy().tobytes()
    assert len(jpeg) == r["jpeg_bytes"] and po.fnv(jpeg) == r["jpeg_fnv"]
    del d_rgb
    d_jpeg = ctx.to_device(np.frombuffer(jpeg, np.uint8))
    rgb = ctx.decode_batch(d_jpeg, [0], [len(jpeg)], w, h, lum, chrom)
    assert po.fnv(rgb.cpu().numpy()) == r["rgb_fnv"]
    del rgb, out
    torch.cuda.empty_cache()


def test_c4_gigapixel_as_eight_stripes(jea, po, gigapixel):
    """The C4 partition: 8 stripes of 2048 rows, each on a context of its own, through the HIP stripe steps."""
    stripes = importlib.import_module("jpeg-encoder_amd.stripes")
    w = h = 16384
    r = _golden_row(w, h, 75, 0)
    lum, chrom = jea.binding.quant_tables(75)
    rows = stripes.stripe_rows(h, 8)
    assert rows == [(i * 2048, (i + 1) * 2048) for i in range(8)]
    ctxs = [jea.Context(0) for _ in rows]
    try:
        d_rgb = ctxs[0].to_device(gigapixel.reshape(-1))
        backends = [stripes.HipBackend(c, d_rgb[y0 * w * 3: y1 * w * 3], w, y1 - y0, lum, chrom)
                    for c, (y0, y1) in zip(ctxs, rows)]
        jpeg = stripes.encode_in_process(backends, w, h)
        assert len(jpeg) == r["jpeg_bytes"] and po.fnv(jpeg) == r["jpeg_fnv"]
        del jpeg
        # and through the entry points that keep the exchanged facts in device memory (what bench.py --config c4 runs)
        on_device = [stripes.DeviceStripe(c, b.d_rgb, w, b.h, lum, chrom, c.empty(3 * w * b.h // 2 + (1 << 20), c.torch.uint8))
                     for c, b in zip(ctxs, backends)]
        jpeg = stripes.encode_in_process_on_device(on_device, w, h)
        assert len(jpeg) == r["jpeg_bytes"] and po.fnv(jpeg) == r["jpeg_fnv"]
    finally:
        for c in ctxs:
            c.close()


def _frame(po, i):
    # distinct sources: smooth pictures with their own noise, and every 64th one uniform noise
    return po.gen_image(FW, FH, 1 if i % 64 == 63 else 0, 1234 + i)


@pytest.fixture(scope="module")
def frames(po):
    with concurrent.futures.ThreadPoolExecutor(THREADS) as pool:
        return np.stack(list(pool.map(lambda i: _frame(po, i), range(N_FRAMES))))


def test_c3_batch_of_1024_frames_in_one_call(ctx, po, jea, frames):
    import torch

    q = 90
    lum, chrom = jea.binding.quant_tables(q)
    with concurrent.futures.ThreadPoolExecutor(THREADS) as pool:
        want = list(pool.map(lambda i: po.encode(frames[i], q), range(N_FRAMES)))
    assert po.fnv(want[0]) == _golden_row(FW, FH, q, 0)["jpeg_fnv"]        # frame 0 is the recorded picture
    d_rgb = ctx.to_device(frames.reshape(-1))
    cap = sum(len(f) for f in want) + 64 * N_FRAMES
    out, off, size = ctx.encode_batch(d_rgb, N_FRAMES, FW, FH, lum, chrom, capacity=cap)
    del d_rgb
    assert np.array_equal(size, np.array([len(f) for f in want], dtype=size.dtype))
    host = out.cpu().numpy()
    del out
    torch.cuda.empty_cache()
    for i in range(N_FRAMES):                                                # every file, byte for byte
        assert host[int(off[i]): int(off[i] + size[i])].tobytes() == want[i], i


def test_c5_decode_of_1024_frames_in_one_call(ctx, po, jea, frames):
    import torch

    q = 50
    lum, chrom = jea.binding.quant_tables(q)
    d_rgb = ctx.to_device(frames.reshape(-1))
    out, off, size = ctx.encode_batch(d_rgb, N_FRAMES, FW, FH, lum, chrom, capacity=N_FRAMES * (3 << 20))
    del d_rgb
    host = out.cpu().numpy()
    files = [host[int(off[i]): int(off[i] + size[i])].tobytes() for i in range(N_FRAMES)]
    assert po.fnv(files[0]) == _golden_row(FW, FH, q, 0)["jpeg_fnv"]
    rgb = ctx.decode_batch(out, off, size, FW, FH, lum, chrom)
    got = rgb.cpu().numpy().reshape(N_FRAMES, FH, FW, 3)
    del rgb, out
    torch.cuda.empty_cache()
    assert po.fnv(got[0].reshape(-1)) == _golden_row(FW, FH, q, 0)["rgb_fnv"]
    with concurrent.futures.ThreadPoolExecutor(THREADS) as pool:
        same = list(pool.map(lambda i: np.array_equal(got[i], po.decode(files[i], FW, FH, q)), range(N_FRAMES)))
    assert all(same), [i for i, s in enumerate(same) if not s][:8]


def test_decode_batch_whose_detailed_groups_overflow_the_inverse_transforms_list(ctx, po, jea):
    """A batch that the files' sizes send to the inverse kernel for single-colour groups (nine files in ten under 2.2
    bytes of scan per MCU) but that holds more groups with detail (507 per textured frame) than that kernel's list for
    the general kernel has places (32 768): the groups that find the list full are transformed by the first kernel
    itself.  Two distinct sources, many copies: every decoded copy must equal the oracle's decode of its source."""
    import torch

    q = 50
    lum, chrom = jea.binding.quant_tables(q)
    smooth = po.gen_image(FW, FH, 0, 4321)
    textured = smooth.astype(np.int16)
    textured[:, :, 1] += np.where((np.arange(FW) % 8) < 4, -24, 24).astype(np.int16)[None, :]   # an edge inside every block
    textured = np.clip(textured, 0, 255).astype(np.uint8)
    d_two = ctx.to_device(np.stack([smooth, textured]).reshape(-1))
    out, off, size = ctx.encode_batch(d_two, 2, FW, FH, lum, chrom, capacity=2 * (3 << 20))
    host = out.cpu().numpy()
    files = [host[int(off[i]): int(off[i] + size[i])].tobytes() for i in range(2)]
    want = [torch.from_numpy(po.decode(f, FW, FH, q).copy()).to(out.device) for f in files]
    assert not torch.equal(want[0], want[1])
    mcus = (FW // 8) * (FH // 8)
    per_mcu = [(len(f) - 607) / mcus for f in files]
    n_textured = 32768 // 507 + 4                        # 68 frames: 34 476 groups with detail
    assert per_mcu[0] < 2.2 < per_mcu[1], per_mcu
    n = 10 * n_textured + 8                              # the textured frames are under a tenth of the batch
    kind = np.zeros(n, dtype=np.int64)
    kind[np.linspace(0, n - 1, n_textured).astype(np.int64)] = 1      # the textured frames spread over the batch
    padded = [f + bytes(-len(f) % 64) for f in files]
    sizes = np.array([len(files[k]) for k in kind], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum([len(padded[k]) for k in kind])[:-1]]).astype(np.uint64)
    assert 10 * int((sizes - 607 < 2.2 * mcus).sum()) >= 9 * n        # the route under test is the one taken
    d_jpeg = ctx.to_device(np.frombuffer(b"".join(padded[k] for k in kind), np.uint8))
    rgb = ctx.decode_batch(d_jpeg, offs, sizes, FW, FH, lum, chrom).view(n, FH, FW, 3)
    wrong = [i for i in range(n) if not torch.equal(rgb[i], want[kind[i]].view(FH, FW, 3))]
    del rgb, d_jpeg
    torch.cuda.empty_cache()
    assert not wrong, (len(wrong), wrong[:8])


@pytest.mark.parametrize("mapper", [0, 1])
def test_decode_batch_of_many_small_smooth_pictures_with_both_mappers(po, jea, mapper):
    """Thousands of small pictures in one call reach the inverse kernel for single-colour groups (4096 groups of 64
    MCUs, small files) too -- with the YCbCr mapper and with the pass-through mapper (colour_mapping.cpp:16-42), and
    with a last group of each picture that is not full (120 MCUs: groups of 64 and 56)."""
    import torch

    w, h, q, n = 120, 64, 50, 4200
    lum, chrom = jea.binding.quant_tables(q)
    ctx = jea.Context(0)
    try:
        ctx.set_colour_mapper(mapper)
        po.set_mapper(mapper)
        flat = np.empty((h, w, 3), np.uint8)
        flat[:] = (90, 140, 200)
        flat[8:24, 16:40] = (91, 139, 202)
        flat[40:48, 64:72, 1] += (np.arange(8, dtype=np.uint8) * 9)[None, :]        # one MCU with detail
        steps = np.empty((h, w, 3), np.uint8)                                       # one colour per MCU, changing from MCU to MCU
        steps[:, :, 0] = (80 + (np.arange(w) // 8) % 4)[None, :]
        steps[:, :, 1] = (120 + (np.arange(h) // 8) % 3)[:, None]
        steps[:, :, 2] = 60
        steps[16:24, 8:12, 0] += 6                                                   # and one MCU with an edge
        sources = [steps, flat]
        files = [po.encode(s, q) for s in sources]
        want = [torch.from_numpy(po.decode(f, w, h, q).copy()).to(ctx.tdev) for f in files]
        mcus = ((w + 7) // 8) * (h // 8)
        assert all(len(f) - 607 < (2.8 if mapper else 2.2) * mcus for f in files), [len(f) for f in files]   # the route's condition
        kind = np.arange(n) % 2
        padded = [f + bytes(-len(f) % 64) for f in files]
        sizes = np.array([len(files[k]) for k in kind], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum([len(padded[k]) for k in kind])[:-1]]).astype(np.uint64)
        d_jpeg = ctx.to_device(np.frombuffer(b"".join(padded[k] for k in kind), np.uint8))
        rgb = ctx.decode_batch(d_jpeg, offs, sizes, w, h, lum, chrom).view(n, h, w, 3)
        wrong = [i for i in range(n) if not torch.equal(rgb[i], want[kind[i]].view(h, w, 3))]
        assert not wrong, (len(wrong), wrong[:8])
    finally:
        po.set_mapper(0)
        ctx.close()


def test_decode_batch_whose_block_records_leave_in_fours_at_every_alignment(ctx, po, jea):
    """Where a batch has work to spare the write pass stores four block records at a time, at 16-byte aligned
    addresses.  459 blocks per picture (an odd number) move every picture's records to another alignment, and the
    sub-sequences' first and last blocks fall anywhere: 10 400 small pictures of four kinds against the oracle."""
    import torch

    w, h, q, n = 136, 72, 50, 10400
    lum, chrom = jea.binding.quant_tables(q)
    sources = []
    for k in range(4):
        img = po.gen_image(w, h, 0, 700 + k)
        img[8 * k: 8 * k + 16, 24: 24 + 8 * (k + 1)] = np.random.RandomState(k).randint(0, 256, (16, 8 * (k + 1), 3))
        sources.append(img)
    files = [po.encode(sx, q) for sx in sources]
    total = sum(len(files[i % 4]) - 607 for i in range(n))
    assert total >= 256 * 248 * 4 * 16 and total * 8 < 40 * 459 * n, total      # the condition for records in fours
    want = [torch.from_numpy(po.decode(f, w, h, q).copy()).to(ctx.tdev) for f in files]
    kind = (np.arange(n) * 7 // 3) % 4
    sizes = np.array([len(files[k]) for k in kind], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    d_jpeg = ctx.to_device(np.frombuffer(b"".join(files[k] for k in kind), np.uint8))
    rgb = ctx.decode_batch(d_jpeg, offs, sizes, w, h, lum, chrom).view(n, h, w, 3)
    wrong = [i for i in range(n) if not torch.equal(rgb[i], want[kind[i]].view(h, w, 3))]
    assert not wrong, (len(wrong), wrong[:8])


def test_image_beyond_the_row_lane_kernels_range(jea, po):
    """32768 x 23040 (755 MPix, 2.26 GB of pixels): with more than 4 GiB / 384 MCUs the whole image leaves the
    row-lane forward kernel for the general one, while stripes of it still take the row-lane kernel -- the two routes
    must write the same file.  The file is decoded again and sampled MCU rows are compared with the oracle, run on
    those rows alone (a block's pixels depend on its own coefficients only)."""
    import torch

    stripes = importlib.import_module("jpeg-encoder_amd.stripes")
    w, h, quality = 32768, 23040, 75
    assert (w // 8) * (h // 8) * 384 >= 1 << 32
    lum, chrom = jea.binding.quant_tables(quality)
    tile = po.gen_image(4096, 2880, 0, 77)
    tile[1000:1400, 500:3000] = po.gen_image(2500, 400, 1, 78)      # a band of noise in every tile
    rgb = np.tile(tile, (8, 8, 1))
    rgb[::2880, :, 0] ^= 0x55                                        # and what makes the tiles' first rows differ
    assert rgb.shape == (h, w, 3)
    rows = stripes.stripe_rows(h, 4)
    ctxs = [jea.Context(0) for _ in rows]
    try:
        c = ctxs[0]
        d_rgb = c.to_device(rgb.reshape(-1))
        cap = 1 << 30
        out, off, size = c.encode_batch(d_rgb, 1, w, h, lum, chrom, capacity=cap)
        whole = out[: int(size[0])].cpu().numpy()
        del out
        striped = [stripes.DeviceStripe(x, d_rgb[y0 * w * 3: y1 * w * 3], w, y1 - y0, lum, chrom, x.empty(cap // 2, torch.uint8))
                   for x, (y0, y1) in zip(ctxs, rows)]
        parts = stripes.encode_in_process_on_device(striped, w, h)
        assert len(parts) == whole.size and parts == whole.tobytes()
        del striped, parts, d_rgb
        torch.cuda.empty_cache()
        d_jpeg = c.to_device(whole)
        dec = c.decode_batch(d_jpeg, [0], [whole.size], w, h, lum, chrom).cpu().numpy().reshape(h, w, 3)
    finally:
        for x in ctxs:
            x.close()
    rng = np.random.RandomState(5)
    sample = sorted(set([0, h // 8 - 1] + list(rng.randint(0, h // 8, 200))))

    def band_ok(r):
        band = np.ascontiguousarray(rgb[8 * r: 8 * r + 8])
        return np.array_equal(dec[8 * r: 8 * r + 8], po.decode(po.encode(band, quality), w, 8, quality))

    with concurrent.futures.ThreadPoolExecutor(THREADS) as tp:
        ok = list(tp.map(band_ok, sample))
    assert all(ok), [r for r, good in zip(sample, ok) if not good][:10]
