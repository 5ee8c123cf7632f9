"""CPU tests of the oracle: its pins, and the host-side (no GPU) helpers of the C-ABI against it."""
import ctypes
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROWS = json.load(open(os.path.join(GOLDEN, "reference_hashes.json")))["rows"]


def _row_id(r):
    return "%dx%d-q%d-m%d" % (r["w"], r["h"], r["q"], r["mode"])


@pytest.mark.parametrize("row", [r for r in ROWS if r["w"] * r["h"] <= 1920 * 1080], ids=_row_id)
def test_oracle_reproduces_reference_hashes(po, row):
    """JPEG bytes and decoded RGB of the reference itself, recorded as hashes (see the fixture's _provenance)."""
    rgb = po.gen_image(row["w"], row["h"], row["mode"], 1234)
    jpeg = po.encode(rgb, row["q"])
    assert len(jpeg) == row["jpeg_bytes"]
    assert po.fnv(jpeg) == row["jpeg_fnv"]
    dec = po.decode(jpeg, row["w"], row["h"], row["q"])
    assert po.fnv(dec.reshape(-1)) == row["rgb_fnv"]


@pytest.mark.parametrize("row", [r for r in ROWS if r["w"] * r["h"] > 1920 * 1080], ids=_row_id)
def test_oracle_reproduces_reference_hashes_large(po, row):
    if row.get("slow") and os.environ.get("JEA_SLOW") != "1":
        pytest.skip("16384x16384 takes about a minute and 2.5 GiB: set JEA_SLOW=1 (verified when the oracle was pinned)")
    rgb = po.gen_image(row["w"], row["h"], row["mode"], 1234)
    jpeg = po.encode(rgb, row["q"])
    assert len(jpeg) == row["jpeg_bytes"] and po.fnv(jpeg) == row["jpeg_fnv"]
    dec = po.decode(jpeg, row["w"], row["h"], row["q"])
    assert po.fnv(dec.reshape(-1)) == row["rgb_fnv"]


def test_cos_table_bits_match_this_libm(po):
    """The embedded cosine bits are what the reference's inline expression gives with glibc here."""
    a, b = np.zeros(64, np.uint32), np.zeros(64, np.uint32)
    sa, sb = ctypes.c_uint32(0), ctypes.c_uint32(0)
    po.LIB.jo_cos_table(a.ctypes.data, ctypes.byref(sa))
    po.LIB.jo_cos_table_runtime(b.ctypes.data, ctypes.byref(sb))
    assert np.array_equal(a, b)
    assert sa.value == sb.value == 0x3EB504F3
    assert a[8 * 4 + 2] == 0xBF3504F1 and a[8 * 7 + 7] == 0xBE47C614  # not symmetric to the last bit


def test_zigzag_walk_equals_table(po):
    a, b = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
    po.LIB.jo_zigzag_order(a.ctypes.data)
    po.LIB.jo_zigzag_walk(b.ctypes.data)
    assert np.array_equal(a, b)
    assert sorted(a.tolist()) == list(range(64))


def test_quant_tables_recorded_values(po):
    """Zig-zag ordered luminance tables recorded from the reference's DQT segment for q=75 and q=90."""
    zz = np.zeros(64, np.uint8)
    po.LIB.jo_zigzag_order(zz.ctypes.data)
    q75 = [8, 6, 6, 7, 6, 5, 8, 7, 7, 7, 9, 9, 8, 10, 12, 20, 13, 12, 11, 11, 12, 25, 18, 19, 15, 20, 29, 26, 31, 30,
           29, 26, 28, 28, 32, 36, 46, 39, 32, 34, 44, 35, 28, 28, 40, 55, 41, 44, 48, 49, 52, 52, 52, 31, 39, 57, 61,
           56, 50, 60, 46, 51, 52, 50]
    q90 = [3, 2, 2, 3, 2, 2, 3, 3, 3, 3, 4, 3, 3, 4, 5, 8, 5, 5, 4, 4, 5, 10, 7, 7, 6, 8, 12, 10, 12, 12, 11, 10, 11,
           11, 13, 14, 18, 16, 13, 14, 17, 14, 11, 11, 16, 22, 16, 17, 19, 20, 21, 21, 21, 12, 15, 23, 24, 22, 20, 24,
           18, 20, 21, 20]
    assert po.quant_tables(75)[0][zz].tolist() == q75
    assert po.quant_tables(90)[0][zz].tolist() == q90
    assert (po.quant_tables(100)[0] == 1).all() and (po.quant_tables(100)[1] == 1).all()
    base = po.quant_tables(50)[0]
    assert base[0] == 16 and base[63] == 99
    assert po.quant_tables(0)[0].tolist() == po.quant_tables(1)[0].tolist()       # clamp, quantiser.cpp:6-11
    assert po.quant_tables(1000)[0].tolist() == po.quant_tables(100)[0].tolist()


def test_huffman_tables_are_annex_k(po):
    code, length = np.zeros(4 * 256, np.uint16), np.zeros(4 * 256, np.uint8)
    po.LIB.jo_huffman_tables(code.ctypes.data, length.ctypes.data)
    code, length = code.reshape(4, 256), length.reshape(4, 256)
    # entropy_encoder.cpp:149-162, :378-379, :383-396, :610-611
    assert (length[0][:12].tolist(), code[0][:12].tolist()) == (
        [2, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9], [0, 2, 3, 4, 5, 6, 14, 30, 62, 126, 254, 510])
    assert (length[2][:12].tolist(), code[2][:12].tolist()) == (
        [2, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], [0, 1, 2, 6, 14, 30, 62, 126, 254, 510, 1022, 2046])
    assert (length[1][0x00], code[1][0x00]) == (4, 0b1010) and (length[1][0xF0], code[1][0xF0]) == (11, 0b11111111001)
    assert (length[3][0x00], code[3][0x00]) == (2, 0b00) and (length[3][0xF0], code[3][0xF0]) == (10, 0b1111111010)
    assert (length[1][0x01], code[1][0x01]) == (2, 0b00) and (length[1][0xFA], code[1][0xFA]) == (16, 0xFFFE)
    assert (length[3][0x01], code[3][0x01]) == (2, 0b01) and (length[3][0xFA], code[3][0xFA]) == (16, 0xFFFE)
    assert (length[1] > 0).sum() == 162 and (length[3] > 0).sum() == 162


def test_header_layout(po):
    lum, chrom = po.quant_tables(75)
    h = po.header(3840, 2160, lum, chrom)
    assert len(h) == 607 and h[:4] == b"\xff\xd8\xff\xe0" and h[20:22] == b"\xff\xdb"
    assert h[154:156] == b"\xff\xc0" and h[159:163] == bytes([2160 >> 8, 2160 & 255, 3840 >> 8, 3840 & 255])
    assert h[173:175] == b"\xff\xc4" and h[593:595] == b"\xff\xda" and h[-3:] == b"\x00\x3f\x00"
    w, hh = ctypes.c_int(0), ctypes.c_int(0)
    buf = np.frombuffer(h, np.uint8)
    assert po.LIB.jo_check_header(buf.ctypes.data, buf.size, ctypes.byref(w), ctypes.byref(hh)) == 0
    assert (w.value, hh.value) == (3840, 2160)


def test_oracle_files_open_in_an_independent_decoder(po):
    """External sanity: libjpeg (via PIL) accepts the files and sees roughly the same picture."""
    PIL = pytest.importorskip("PIL.Image")
    import io

    for mode in (0, 1):
        rgb = po.gen_image(96, 64, mode, 7)
        im = np.asarray(PIL.open(io.BytesIO(po.encode(rgb, 90))).convert("RGB"))
        assert im.shape == rgb.shape
        assert np.abs(im.astype(int) - rgb.astype(int)).mean() < (4 if mode == 0 else 12)


def test_run_length_rules(po):
    """entropy_encoder.cpp:94-120 on hand-made blocks, bit counts worked out by hand from the Annex-K codes."""
    z = np.zeros((1, 3, 64), np.int16)
    assert po.coefs_to_bit_lengths(z).tolist() == [2 + 4, 2 + 2, 2 + 2]           # DC 0 + EOB
    b = z.copy(); b[0, 0, 63] = 3                                                # run 62 = 3 ZRL + (14, size 2), no EOB
    assert po.coefs_to_bit_lengths(b)[0] == 2 + 3 * 11 + 16 + 2
    b = z.copy(); b[0, 0, 17] = 1                                                # ZRL + (0,1) + EOB
    assert po.coefs_to_bit_lengths(b)[0] == 2 + 11 + (2 + 1) + 4
    b = z.copy(); b[0, 0, 1] = 1                                                 # trailing zeros collapse into EOB
    assert po.coefs_to_bit_lengths(b)[0] == 2 + (2 + 1) + 4
    scan, bits = po.coefs_to_scan(b)
    assert bits == (2 + 3 + 4) + 4 + 4 and scan == bytes([0b00001101, 0b00000000, 0b00000000])


def test_decode_drops_last_entry_without_eob(po):
    """entropy_encoder.cpp:129: the run-length removal skips the last entry, so zz[63] is lost."""
    coefs = np.zeros((2, 3, 64), np.int16)
    coefs[0, 0, 63] = 3
    coefs[0, 0, 5] = -7
    coefs[1, 1, 1] = 2
    scan, _ = po.coefs_to_scan(coefs)
    got, used, nlen = po.scan_to_coefs(scan + b"\xff\xd9", 2)
    want = coefs.copy()
    want[0, 0, 63] = 0
    assert np.array_equal(got, want)
    assert used == nlen - 2


def test_stuffing_round_trip(po):
    coefs = np.zeros((50, 3, 64), np.int16)
    coefs[:, :, 1:] = -1023
    scan, bits = po.coefs_to_scan(coefs)
    assert scan.count(b"\xff\x00") > 100 and b"\xff\xff" not in scan
    got, used, nlen = po.scan_to_coefs(scan + b"\xff\xd9", 50)
    want = coefs.copy(); want[:, :, 63] = 0
    assert np.array_equal(got, want)


def test_edge_replication(po):
    """block_grid.cpp:16-26: a 13x7 image equals its 16x8 edge-replicated extension, coefficient for coefficient."""
    lum, chrom = po.quant_tables(75)
    rgb = po.gen_image(13, 7, 1, 5)
    ext = np.zeros((8, 16, 3), np.uint8)
    ext[:7, :13] = rgb
    ext[:7, 13:] = rgb[:, 12:13]
    ext[7:] = ext[6:7]
    assert np.array_equal(po.image_to_coefs(rgb, lum, chrom), po.image_to_coefs(ext, lum, chrom))


def test_optimised_tables_mode(po):
    """Per-image Huffman tables: this build's own procedure (the reference has none, entropy_encoder.cpp:614), so
    nothing of the reference pins it -- parity unpinned.  What can be checked on the CPU: an independent decoder
    (libjpeg behind PIL) reads such a file and gets the pixels of the file with the Annex K tables; the layout of
    the 607-byte header is unchanged; DC codes stay within 11 bits, AC codes within 16, no code is all ones; the
    oracle reads its own file back; files do not grow beyond the three bytes a worse last byte may cost."""
    import io

    Image = pytest.importorskip("PIL.Image")
    for (w, h, mode, q) in [(64, 48, 0, 75), (200, 120, 1, 90), (512, 512, 0, 50), (37, 21, 1, 100), (8, 8, 0, 10), (97, 33, 3, 60)]:
        rgb = po.gen_image(w, h, mode, 5)
        std, opt = po.encode(rgb, q), po.encode_optimised(rgb, q)
        assert opt[:177] == std[:177] and opt[593:607] == std[593:607]
        assert len(opt) <= len(std) + 3
        spec = opt[177:593]
        for at, n, limit in ((0, 12, 11), (29, 162, 16), (208, 12, 11), (237, 162, 16)):
            bits = list(spec[at + 1: at + 17])
            assert sum(bits) == n and all(b == 0 for b in bits[limit:])
            assert sum(b * 2 ** (16 - l - 1) for l, b in enumerate(bits)) < 2 ** 16       # room left for the all-ones word
            assert sorted(spec[at + 17: at + 17 + n]) == sorted(std[177 + at + 17: 177 + at + 17 + n])
        a = np.asarray(Image.open(io.BytesIO(std)).convert("RGB"))
        b = np.asarray(Image.open(io.BytesIO(opt)).convert("RGB"))
        assert np.array_equal(a, b)
        assert np.array_equal(po.decode_file_tables(opt, w, h, q), po.decode(std, w, h, q))
        assert po.encode(rgb, q) == std                                                     # the mode leaves nothing behind
