"""Per-image Huffman tables (jea_set_huffman_optimised): files against the oracle's statement of the procedure and
against an independent decoder (PIL / libjpeg)."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def octx(jea):
    c = jea.Context(0)
    c.set_huffman_optimised(True)
    yield c
    c.close()


@pytest.mark.parametrize("w,h,mode,q", [(64, 48, 0, 75), (200, 120, 1, 90), (512, 512, 0, 50), (37, 21, 1, 100), (8, 8, 0, 10),
                                        (1920, 1080, 0, 90), (640, 480, 3, 60)])
def test_optimised_file_equals_oracle(octx, po, jea, w, h, mode, q):
    rgb = po.gen_image(w, h, mode, 11)
    lum, chrom = jea.binding.quant_tables(q)
    want = po.encode_optimised(rgb, q)
    got = octx.encode_host(rgb, lum, chrom)
    assert got == want
    assert len(got) <= len(po.encode(rgb, q)) + 8


def test_optimised_batch_equals_oracle(octx, po, jea):
    w, h, q, n = 136, 72, 80, 6
    lum, chrom = jea.binding.quant_tables(q)
    imgs = np.stack([po.gen_image(w, h, i % 2, 40 + i) for i in range(n)])
    out, off, size = octx.encode_batch(octx.to_device(imgs.reshape(-1)), n, w, h, lum, chrom)
    host = out.cpu().numpy()
    for i in range(n):
        assert host[int(off[i]): int(off[i] + size[i])].tobytes() == po.encode_optimised(imgs[i], q), i


def test_optimised_file_decodes_elsewhere(octx, po, jea):
    from PIL import Image

    w, h, q = 320, 200, 85
    rgb = po.gen_image(w, h, 1, 3)
    lum, chrom = jea.binding.quant_tables(q)
    opt = octx.encode_host(rgb, lum, chrom)
    std = po.encode(rgb, q)
    a = np.asarray(Image.open(io.BytesIO(std)).convert("RGB"))
    b = np.asarray(Image.open(io.BytesIO(opt)).convert("RGB"))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("w,h,mode,q", [(64, 48, 0, 75), (200, 120, 1, 90), (512, 512, 0, 50), (37, 21, 1, 100), (1920, 1080, 0, 50),
                                        (3840, 2160, 0, 75)])
def test_optimised_file_decodes_to_the_same_pixels(octx, ctx, po, jea, w, h, mode, q):
    rgb = po.gen_image(w, h, mode, 21)
    lum, chrom = jea.binding.quant_tables(q)
    opt = octx.encode_host(rgb, lum, chrom)
    got = octx.decode_host(opt, w, h, lum, chrom)
    assert np.array_equal(got, po.decode_file_tables(opt, w, h, q))
    # the same coefficients as the file with the Annex K tables holds
    assert np.array_equal(got, ctx.decode_host(po.encode(rgb, q), w, h, lum, chrom))
    # and a context in that mode still reads files with the Annex K tables, which carry them in the same place
    assert np.array_equal(octx.decode_host(po.encode(rgb, q), w, h, lum, chrom), got)


def test_optimised_batch_round_trip(octx, po, jea):
    w, h, q, n = 136, 72, 80, 20
    lum, chrom = jea.binding.quant_tables(q)
    imgs = np.stack([po.gen_image(w, h, i % 2, 70 + i) for i in range(n)])
    out, off, size = octx.encode_batch(octx.to_device(imgs.reshape(-1)), n, w, h, lum, chrom)
    rgb = octx.decode_batch(out, off, size, w, h, lum, chrom).cpu().numpy().reshape(n, h, w, 3)
    host = out.cpu().numpy()
    for i in range(n):
        f = host[int(off[i]): int(off[i] + size[i])].tobytes()
        assert np.array_equal(rgb[i], po.decode_file_tables(f, w, h, q)), i


def test_broken_tables_fail_cleanly(octx, po, jea):
    w, h, q = 96, 64, 70
    lum, chrom = jea.binding.quant_tables(q)
    opt = bytearray(octx.encode_host(po.gen_image(w, h, 1, 5), lum, chrom))
    for at, value in ((177, 0x07), (178 + 1, 200), (206 + 1 + 3, 255), (206 + 17 + 5, 0xFF)):
        bad = bytearray(opt)
        bad[at] = value
        try:
            octx.decode_host(bytes(bad), w, h, lum, chrom)
        except jea.JeaError:
            pass


def test_optimised_random_cases(octx, po, jea):
    rng = np.random.RandomState(99)
    for case in range(60):
        w, h = int(rng.randint(1, 400)), int(rng.randint(1, 300))
        q = int(rng.choice([1, 10, 35, 50, 75, 90, 100]))
        mode = int(rng.randint(0, 4))
        rgb = po.gen_image(w, h, mode, 1000 + case)
        if case % 7 == 0:
            rgb[:] = rng.randint(0, 256, 3).astype(np.uint8)       # one colour: every block is two or three bits long
        lum, chrom = jea.binding.quant_tables(q)
        want = po.encode_optimised(rgb, q)
        got = octx.encode_host(rgb, lum, chrom)
        assert got == want, (case, w, h, q, mode)
        assert np.array_equal(octx.decode_host(got, w, h, lum, chrom), po.decode_file_tables(want, w, h, q)), (case, w, h, q, mode)


def test_entry_points_without_the_mode_refuse_it(octx, po, jea):
    """Stripes (the symbol counts would need an exchange) and the static-scratch encode keep the Annex K tables: in
    the mode they return an error instead of writing a file the caller did not ask for."""
    import torch

    w, h = 64, 32
    lum, chrom = jea.binding.quant_tables(75)
    d_rgb = octx.to_device(po.gen_image(w, h, 0, 1).reshape(-1))
    with pytest.raises(jea.JeaError):
        octx.stripe_transform(d_rgb, w, h, lum, chrom)
    out = torch.empty(1 << 16, dtype=torch.uint8, device=octx.tdev)
    off = torch.zeros(1, dtype=torch.int64, device=octx.tdev)
    size = torch.zeros(1, dtype=torch.int64, device=octx.tdev)
    status = torch.zeros(16, dtype=torch.int32, device=octx.tdev)
    with pytest.raises(jea.JeaError):
        octx.encode_batch_static_async(d_rgb, 1, w, h, lum, chrom, out, off, size, status)
    octx.set_huffman_optimised(False)
    octx.encode_batch_static_async(d_rgb, 1, w, h, lum, chrom, out, off, size, status)
    octx.synchronize()
    assert out[: int(size[0])].cpu().numpy().tobytes() == po.encode(po.gen_image(w, h, 0, 1), 75)
