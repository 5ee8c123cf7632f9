"""The oracle's bit packing, stuffing and unstuffing against the reference's own BitStream class,
compiled from /root/reference/jpeg/src/bit_stream.cpp into oracle/_ref (the one translation unit of the
reference that builds in this image without a stand-in for <SDL.h>)."""
import ctypes
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(po):
    if not os.path.exists(po.REF_BITSTREAM_PATH):
        pytest.skip("oracle/_ref not built (reference sources absent)")
    lib = ctypes.CDLL(po.REF_BITSTREAM_PATH)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    lib.refbs_new.restype = vp
    lib.refbs_delete.argtypes = [vp]
    lib.refbs_push_bits.argtypes = [vp, ctypes.c_uint16, sz]
    lib.refbs_push_into_alignment.argtypes = [vp]
    lib.refbs_size.argtypes = [vp]
    lib.refbs_size.restype = sz
    lib.refbs_copy.argtypes = [vp, vp]
    lib.refbs_stuff.argtypes = [vp, sz]
    lib.refbs_unstuff.argtypes = [vp, sz]
    return lib


def _ref_bytes(ref, s):
    out = np.zeros(ref.refbs_size(s), np.uint8)
    ref.refbs_copy(s, out.ctypes.data)
    return out.tobytes()


def test_block_bits_match_reference_bitstream(ref, po):
    """Push the oracle's own (code, length) sequence through the reference BitStream: same bytes as the oracle's
    packer gives, including the zero padding of pushIntoAlignment (bit_stream.cpp:44-81,101-105)."""
    rng = np.random.RandomState(3)
    code, length = np.zeros(4 * 256, np.uint16), np.zeros(4 * 256, np.uint8)
    po.LIB.jo_huffman_tables(code.ctypes.data, length.ctypes.data)
    code, length = code.reshape(4, 256), length.reshape(4, 256)
    coefs = np.zeros((40, 3, 64), np.int16)
    for b in coefs.reshape(-1, 64):
        n = rng.randint(0, 30)
        b[rng.choice(64, n, replace=False)] = rng.randint(-200, 201, n)
    # independent symbol walk in python following entropy_encoder.cpp:94-120,715-765
    s = ref.refbs_new()
    last = [0, 0, 0]
    for m in range(coefs.shape[0]):
        for ch in range(3):
            zz = coefs[m, ch].astype(int)
            t = 0 if ch == 0 else 2
            diff = int(zz[0]) - last[ch]; last[ch] = int(zz[0])
            ss = abs(diff).bit_length()
            ref.refbs_push_bits(s, int(code[t][ss]), int(length[t][ss]))
            if ss:
                ref.refbs_push_bits(s, (diff if diff > 0 else ~(-diff)) & 0xFFFF, ss)
            entries, run = [], 0
            for v in zz[1:]:
                if run == 15 or v != 0:
                    entries.append((run, int(v))); run = 0
                else:
                    run += 1
            if zz[63] == 0:
                while entries and entries[-1][1] == 0:
                    entries.pop()
                entries.append((0, 0))
            for run, v in entries:
                ss = abs(v).bit_length()
                sym = (run << 4) | ss
                ref.refbs_push_bits(s, int(code[t + 1][sym]), int(length[t + 1][sym]))
                if ss:
                    ref.refbs_push_bits(s, (v if v > 0 else ~(-v)) & 0xFFFF, ss)
    ref.refbs_push_into_alignment(s)
    raw = _ref_bytes(ref, s)
    ref.refbs_delete(s)
    scan, bits = po.coefs_to_scan(coefs)
    assert len(raw) == (bits + 7) // 8
    assert scan.replace(b"\xff\x00", b"\xff") == raw


def test_stuffing_matches_reference(ref, po):
    """stuffBytes / removeStuffedBytes of the reference (bit_stream.cpp:127-147) on a stream sized so that no
    vector::insert reallocates (1025 bytes pushed -> capacity 2048, at most 600 insertions)."""
    rng = np.random.RandomState(11)
    data = rng.randint(0, 256, 1025).astype(np.uint8)
    data[rng.choice(1025, 500, replace=False)] = 0xFF
    data[:8] = 0xFF          # before `from`: must stay untouched
    data[-1] = 0x12
    s = ref.refbs_new()
    for b in data:
        ref.refbs_push_bits(s, int(b), 8)
    ref.refbs_stuff(s, 8)
    stuffed = _ref_bytes(ref, s)
    want = bytes(data[:8]) + bytes(data[8:]).replace(b"\xff", b"\xff\x00")
    assert stuffed == want
    # unstuff needs the two trailing marker bytes the reference skips
    ref.refbs_push_bits(s, 0xFFD9, 16)
    ref.refbs_unstuff(s, 8)
    assert _ref_bytes(ref, s) == bytes(data) + b"\xff\xd9"
    ref.refbs_delete(s)
    buf = np.frombuffer(want + b"\xff\xd9", np.uint8).copy()
    n = po.LIB.jo_remove_stuffing(buf.ctypes.data, 8, buf.size)
    assert buf[:n].tobytes() == bytes(data) + b"\xff\xd9"


def _reference_walk_stays_inside(raw, start):
    """removeStuffedBytes compares its iterator with end() - 2 by `!=`: a 0xFF three bytes before the end that is
    not followed by 0x00 carries it past that point and off the vector.  True when the walk ends on end() - 2."""
    data, i = list(raw), start
    while i != len(data) - 2:
        if i > len(data) - 2:
            return False
        if data[i] == 0xFF:
            i += 1
            if i >= len(data):
                return False
            if data[i] == 0x00:
                del data[i]
                continue
        i += 1
    return True


def test_unstuffing_of_ff_runs_matches_reference(ref, po):
    """removeStuffedBytes (bit_stream.cpp:136-147) on byte strings no encoder produces: runs of 0xFF before a
    0x00, zeros after zeros.  After a 0xFF that no 0x00 follows the reference steps over the following byte as
    well, so in FF FF 00 the zero stays and in FF FF FF 00 it goes; the oracle's restatement must leave the same
    bytes, from every starting offset.  Strings on which the reference walks off its vector are left out."""
    rng = np.random.RandomState(23)
    tail = bytes([0x05, 0xFF, 0xD9])
    cases = [bytes([0xFF, 0xFF, 0x00, 0x11, 0x22]) + tail, bytes([0xFF, 0xFF, 0xFF, 0x00, 0x11, 0x22]) + tail,
             bytes([0x01, 0xFF, 0x00, 0x00, 0xFF, 0x00]) + tail, bytes([0xFF, 0x00] * 4) + tail,
             bytes([0xFF] * 40 + [0x00]) + tail, bytes([0xFF] * 41 + [0x00]) + tail]
    for n in (2, 3, 4, 5, 8, 17, 64, 257, 1000):
        for _ in range(12):
            body = rng.choice(np.array([0xFF, 0xFF, 0x00, 0x00, 0x01], np.uint8), n).tobytes()
            cases.append(body + tail)
            cases.append(body + b"\xff\xd9")
    compared = 0
    for raw in cases:
        for start in (0, 1, 3):
            if start > len(raw) - 2 or not _reference_walk_stays_inside(raw, start):
                continue
            s = ref.refbs_new()
            for b in raw:
                ref.refbs_push_bits(s, int(b), 8)
            ref.refbs_unstuff(s, start)
            want = _ref_bytes(ref, s)
            ref.refbs_delete(s)
            buf = np.frombuffer(raw, np.uint8).copy()
            n = po.LIB.jo_remove_stuffing(buf.ctypes.data, start, buf.size)
            assert buf[:n].tobytes() == want, (raw.hex(), start)
            compared += 1
    assert compared > 400
