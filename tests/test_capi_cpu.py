"""No-GPU checks of the C-ABI library: it loads, exports every declared symbol, and its host-side helpers
(tables, header, header validation, size bound) agree with the oracle.  No compute entry point is called."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(jea):
    lib = jea.load_library()
    header = open(os.path.join(ROOT, "include", "jpegenc_amd.h")).read()
    declared = set(re.findall(r"\b(jea_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(jea.binding.EXPORTS)
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_quant_tables_match_oracle(jea, po):
    for q in list(range(-3, 104)):
        a, b = jea.binding.quant_tables(q), po.quant_tables(q)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), q


def test_header_matches_oracle(jea, po):
    for q, w, h in [(75, 3840, 2160), (10, 1, 1), (100, 65535, 65535), (50, 17, 33)]:
        lum, chrom = jea.binding.quant_tables(q)
        assert jea.binding.write_header(w, h, lum, chrom) == po.header(w, h, lum, chrom)


def test_max_size_bound(jea, po):
    for w, h in [(1, 1), (8, 8), (13, 7), (3840, 2160), (16384, 16384)]:
        assert jea.binding.max_jpeg_bytes(w, h) == po.LIB.jo_max_jpeg_bytes(w, h)
    # a worst-case-ish file stays under the bound
    rgb = po.gen_image(64, 64, 1, 1)
    assert len(po.encode(rgb, 100)) <= jea.binding.max_jpeg_bytes(64, 64)


def test_check_header_messages(jea, po):
    """Messages and field order of Encoder::decodeHeader (encoder.cpp:141-281)."""
    good = bytearray(po.encode(po.gen_image(24, 16, 0, 1), 75))
    ok, msg, w, h = jea.binding.check_header(bytes(good))
    assert ok and msg is None and (w, h) == (24, 16)
    cases = {
        0: "Failed to find SOI marker",
        3: "Failed to find APP-0 marker",
        5: "APP-0 length parameter does not correspond to payload size",
        6: "Failed to find 'J'(FIF) in APP-0 payload",
        12: "JFIF version indicated by APP-0 payload is not 1.02",
        155: "Failed to find SOF0 marker",
        158: "Failed to find precision in SOF0 payload",
        163: "Failed to find number of components in SOF0 payload",
        166: "Failed to find first component quantisation table ID in SOF0 payload",
        594: "Failed to find SOS marker",
        598: "Failed to find first component ID in SOS payload",
        604: "Failed to find spectral alignment in SOS payload",
        606: "Failed to find skip byte in SOS payload",
    }
    for pos, want in cases.items():
        bad = bytearray(good)
        bad[pos] ^= 0x40
        ok, msg, _, _ = jea.binding.check_header(bytes(bad))
        assert not ok and msg == want, (pos, msg)
    # the DQT and DHT payloads are skipped, not validated (encoder.cpp:188, :240)
    bad = bytearray(good); bad[30] ^= 0xFF; bad[300] ^= 0xFF
    assert jea.binding.check_header(bytes(bad))[0]
    assert not jea.binding.check_header(bytes(good[:100]))[0]


def test_read_header_returns_the_files_own_tables(jea, po):
    """Dimensions and DQT tables as the file states them (host-only entry point)."""
    for quality, (w, h) in [(75, (40, 24)), (37, (8, 8)), (100, (13, 7))]:
        jpeg = po.encode(po.gen_image(w, h, 1, quality), quality)
        lum, chrom = jea.binding.quant_tables(quality)
        assert lum.max() <= 255 and chrom.max() <= 255
        rw, rh, rlum, rchrom = jea.binding.read_header(jpeg)
        assert (rw, rh) == (w, h) and np.array_equal(rlum, lum) and np.array_equal(rchrom, chrom)
    # quality 10: entries above 255 are truncated to 8 bits in the file (quantiser.cpp:87-102)
    lum, chrom = jea.binding.quant_tables(10)
    jpeg = po.encode(po.gen_image(16, 16, 1, 1), 10)
    _, _, rlum, _ = jea.binding.read_header(jpeg)
    assert lum.max() > 255 and np.array_equal(rlum, lum & 0xff)
    bad = bytearray(jpeg); bad[20 + 4] = 7
    with pytest.raises(jea.JeaError):
        jea.binding.read_header(bytes(bad))
    # a DHT segment that declares another code (one symbol changed): the strict check of the reference
    # does not look at it, reading the header for a decode does
    other = bytearray(jpeg); other[173 + 2 + 2 + 1 + 16 + 3] ^= 1
    assert jea.binding.check_header(bytes(other))[0]
    with pytest.raises(jea.JeaError):
        jea.binding.read_header(bytes(other))


def test_missing_library_fails_loudly(jea, monkeypatch, tmp_path):
    monkeypatch.setattr(jea.binding, "_lib", None)
    monkeypatch.setattr(jea.binding, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        jea.binding.load_library()
    except ImportError as e:
        assert "not found" in str(e)
    else:
        raise AssertionError("a missing library must raise")


def test_read_header_ex_accepts_per_image_tables(jea, po):
    """jea_read_header refuses files whose DHT segment is not the reference's; jea_read_header_ex accepts per-image
    tables in the same layout and says so."""
    rgb = po.gen_image(80, 56, 1, 2)
    std, opt = po.encode(rgb, 70), po.encode_optimised(rgb, 70)
    w, h, lum, chrom, own = jea.binding.read_header_ex(std)
    assert (w, h, own) == (80, 56, False)
    w, h, lum2, chrom2, own = jea.binding.read_header_ex(opt)
    assert (w, h, own) == (80, 56, True) and (lum2 == lum).all() and (chrom2 == chrom).all()
    with pytest.raises(jea.JeaError):
        jea.binding.read_header(opt)
    bad = bytearray(opt)
    bad[177 + 1 + 11] = 1            # a DC code of 12 bits
    with pytest.raises(jea.JeaError):
        jea.binding.read_header_ex(bytes(bad))
    bad = bytearray(opt)
    bad[177 + 29] = 0x11             # wrong table id
    with pytest.raises(jea.JeaError):
        jea.binding.read_header_ex(bytes(bad))
