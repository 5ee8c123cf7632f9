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
