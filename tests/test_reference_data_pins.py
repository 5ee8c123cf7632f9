"""Data that the reference holds as literals in its sources, read from those sources as text (nothing of the
reference is compiled or run) and compared entry for entry with what the oracle and the library's host-side helpers
produce:

  * the four Huffman code tables, brace-initialised in HuffmanEncoder's constructor
    (jpeg/src/entropy_encoder.cpp:146-613): 2 x (12 DC codes, 16 x 10 AC codes, end-of-block, zero-run);
  * the DHT segment's payload, a second hard-coded copy of the same codes in canonical form (:639-669) and the order
    in which the four tables are written (:672-693);
  * the Annex K base quantisation matrices (jpeg/src/quantiser.cpp:14-24, :36-46);
  * the zig-zag index list (quantiser.cpp:93);
  * the marker constants (jpeg/inc/markers.hpp).

This pins the *data* of the quantiser tables, the zig-zag order, the Huffman codes and the header's table segments to
the reference itself.  The arithmetic (colour conversion, transforms, quantiser rounding, symbol generation) is not
pinned by this file.  Build container only: the reference does not travel to the GPU box, where these tests skip.
"""
import os
import re

import numpy as np
import pytest

REF = "/root/reference/jpeg"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are not present")


def _lines(path, first, last):
    with open(os.path.join(REF, path)) as f:
        return "".join(f.readlines()[first - 1: last])


def _code_pairs():
    """(length, code) pairs in source order from the constructor's initialiser lists."""
    text = _lines("src/entropy_encoder.cpp", 146, 613)
    return [(int(a), int(b, 2), len(b)) for a, b in re.findall(r"\{\s*(\d+)\s*,\s*0b([01]+)\s*\}", text)]


def _byte_vector(text, name):
    body = re.search(r"std::vector<uint8_t>\s+" + name + r"\s*\{\{(.*?)\}\}", text, re.S).group(1)
    return [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", body)]


def test_huffman_code_tables_equal_the_constructor_literals(po):
    pairs = _code_pairs()
    assert len(pairs) == 2 * (12 + 160 + 2)
    code, length = np.zeros(4 * 256, np.uint16), np.zeros(4 * 256, np.uint8)
    po.LIB.jo_huffman_tables(code.ctypes.data, length.ctypes.data)
    code, length = code.reshape(4, 256), length.reshape(4, 256)
    for cls in range(2):                       # 0 luminance, 1 chrominance
        part = pairs[cls * 174: (cls + 1) * 174]
        dc, ac, eob, zrl = part[:12], part[12:172], part[172], part[173]
        for ssss, (n, word, _) in enumerate(dc):
            assert (int(length[2 * cls][ssss]), int(code[2 * cls][ssss])) == (n, word), ("dc", cls, ssss)
        for i, (n, word, _) in enumerate(ac):
            run, size = divmod(i, 10)          # first index run length, second index size - 1
            sym = (run << 4) | (size + 1)
            assert (int(length[2 * cls + 1][sym]), int(code[2 * cls + 1][sym])) == (n, word), ("ac", cls, run, size + 1)
        assert (int(length[2 * cls + 1][0x00]), int(code[2 * cls + 1][0x00])) == eob[:2]
        assert (int(length[2 * cls + 1][0xF0]), int(code[2 * cls + 1][0xF0])) == zrl[:2]
        # and nothing else: 162 AC symbols, 12 DC symbols
        assert int((length[2 * cls + 1] > 0).sum()) == 162 and int((length[2 * cls] > 0).sum()) == 12
    # the literals are written with as many digits as the code has bits -- but for one: the chrominance DC code of
    # category 0 is written 0b000 with length 2 (entropy_encoder.cpp:384); its value, 0, is what counts
    odd = [(n, word, digits) for n, word, digits in pairs if n != digits]
    assert odd == [(2, 0, 3)]


def _dht_segment_from_source():
    text = _lines("src/entropy_encoder.cpp", 636, 694)
    v = {name: _byte_vector(text, name) for name in (
        "luminanceDcCodeLengths", "luminanceDcValues", "chrominanceDcCodeLengths", "chrominanceDcValues",
        "luminanceAcCodeLengths", "luminanceAcValues", "chrominanceAcCodeLengths", "chrominanceAcValues")}
    assert [len(v[k]) for k in v] == [16, 12, 16, 12, 16, 162, 16, 162]
    # table order and ids as pushed at :672-693
    ids = [int(x, 16) for x in re.findall(r"pushByte\(0x([0-9A-Fa-f]{2})\);\s*// table type", text)]
    assert ids == [0x00, 0x10, 0x01, 0x11]
    order = re.findall(r"for_each\((\w+),", text)
    assert order == ["luminanceDcCodeLengths", "luminanceDcValues", "luminanceAcCodeLengths", "luminanceAcValues",
                     "chrominanceDcCodeLengths", "chrominanceDcValues", "chrominanceAcCodeLengths", "chrominanceAcValues"]
    payload = []
    for i, table_id in enumerate(ids):
        payload += [table_id] + v[order[2 * i]] + v[order[2 * i + 1]]
    marker = _marker_constants()["markerDefineHuffmanTableSegmentDHT"]
    n = 2 + 4 * 17 + 12 + 12 + 162 + 162      # the length expression at :673
    return bytes([marker >> 8, marker & 255, n >> 8, n & 255] + payload), v


def _marker_constants():
    text = open(os.path.join(REF, "inc/markers.hpp")).read()
    return {name: int(value, 16) for name, value in re.findall(r"(marker\w+)\s*\{\s*0x([0-9A-Fa-f]{4})\s*\}", text)}


def test_dht_segment_equals_the_reference_literals(po, jea):
    segment, v = _dht_segment_from_source()
    assert len(segment) == 420
    lum, chrom = po.quant_tables(75)
    assert po.header(640, 480, lum, chrom)[173:593] == segment
    # the library's header writer (host code, no GPU involved)
    assert jea.binding.write_header(640, 480, lum, chrom)[173:593] == segment
    # the segment's canonical codes (T.81 Annex C) are the constructor's code words: the reference's two hard-coded
    # copies agree with each other, so either pins the codes
    pairs = _code_pairs()
    for cls, (dc_l, dc_v, ac_l, ac_v) in enumerate((
            ("luminanceDcCodeLengths", "luminanceDcValues", "luminanceAcCodeLengths", "luminanceAcValues"),
            ("chrominanceDcCodeLengths", "chrominanceDcValues", "chrominanceAcCodeLengths", "chrominanceAcValues"))):
        part = pairs[cls * 174: (cls + 1) * 174]
        for lengths, values, is_ac in ((v[dc_l], v[dc_v], False), (v[ac_l], v[ac_v], True)):
            canon, word, k = {}, 0, 0
            for n in range(1, 17):
                for _ in range(lengths[n - 1]):
                    canon[values[k]] = (n, word)
                    word += 1
                    k += 1
                word <<= 1
            if not is_ac:
                assert [canon[s] for s in range(12)] == [p[:2] for p in part[:12]]
            else:
                for i, p in enumerate(part[12:172]):
                    run, size = divmod(i, 10)
                    assert canon[(run << 4) | (size + 1)] == p[:2]
                assert canon[0x00] == part[172][:2] and canon[0xF0] == part[173][:2]


def _matrix(text):
    body = re.search(r"=\s*\{(.*?)\};", text, re.S).group(1)
    return [int(x) for x in re.findall(r"\d+", body)]


def test_quantiser_base_tables_and_zigzag_equal_the_reference_literals(po, jea):
    base_lum = _matrix(_lines("src/quantiser.cpp", 14, 24))
    base_chr = _matrix(_lines("src/quantiser.cpp", 36, 46))
    assert len(base_lum) == 64 and len(base_chr) == 64
    # quality 50 gives S = 100, so the tables are the base matrices themselves (quantiser.cpp:25-29)
    lum, chrom = po.quant_tables(50)
    assert lum.tolist() == base_lum and chrom.tolist() == base_chr
    lum_lib, chrom_lib = jea.binding.quant_tables(50)
    assert lum_lib.tolist() == base_lum and chrom_lib.tolist() == base_chr
    # the other qualities follow from the base matrices by the arithmetic of :25-32, restated here in float32
    for quality in (1, 10, 19, 49, 51, 75, 90, 100):
        s = np.float32(5000 // quality if quality < 50 else 200 - 2 * quality)
        for base, got, got_lib in ((base_lum, po.quant_tables(quality)[0], jea.binding.quant_tables(quality)[0]),
                                   (base_chr, po.quant_tables(quality)[1], jea.binding.quant_tables(quality)[1])):
            want = np.floor((s * np.array(base, np.float32) + np.float32(50)) / np.float32(100)).astype(np.uint16)
            want[want == 0] = 1
            assert got.tolist() == want.tolist() and got_lib.tolist() == want.tolist(), quality
    # zig-zag index list of quantiser.cpp:93
    line = _lines("src/quantiser.cpp", 93, 93)
    indices = [int(x) for x in re.findall(r"\d+", line.split("{", 1)[1])]
    assert sorted(indices) == list(range(64))
    zz = np.zeros(64, np.uint8)
    po.LIB.jo_zigzag_order(zz.ctypes.data)
    assert zz.tolist() == indices
    walk = np.zeros(64, np.uint8)
    po.LIB.jo_zigzag_walk(walk.ctypes.data)      # the walk of entropy_encoder.cpp:15-77, restated
    assert walk.tolist() == indices
    # and the DQT segment carries the tables in that order (quantiser.cpp:87-102)
    lum, chrom = po.quant_tables(75)
    for header in (po.header(64, 64, lum, chrom), jea.binding.write_header(64, 64, lum, chrom)):
        assert list(header[25:25 + 64]) == [int(lum[i]) & 255 for i in indices]
        assert list(header[25 + 65:25 + 65 + 64]) == [int(chrom[i]) & 255 for i in indices]


def test_marker_constants(po):
    m = _marker_constants()
    assert len(m) == 8
    lum, chrom = po.quant_tables(75)
    h = po.header(3840, 2160, lum, chrom)
    at = {"markerStartOfImageSegmentSOI": 0, "markerJFIFImageSegmentAPP0": 2, "markerDefineQuantisationTableSegmentDQT": 20,
          "markerStartOfFrame0SOF0": 154, "markerDefineHuffmanTableSegmentDHT": 173, "markerStartOfScanSegmentSOS": 593}
    assert set(at) | {"markerCommentSegmentCOM", "markerEndOfImageSegmentEOI"} == set(m)
    for name, offset in at.items():
        assert h[offset: offset + 2] == bytes([m[name] >> 8, m[name] & 255]), name
    # the end-of-image marker closes every file the oracle writes
    eoi = m["markerEndOfImageSegmentEOI"]
    assert po.encode(po.gen_image(16, 16, 0, 1), 75)[-2:] == bytes([eoi >> 8, eoi & 255])
