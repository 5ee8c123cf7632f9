"""ctypes view of the CPU oracle (oracle/libjpeg_oracle.so) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py; never by the shipped library or its bindings.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libjpeg_oracle.so")
REF_BITSTREAM_PATH = os.path.join(HERE, "_ref", "libref_bitstream.so")
RECORDED_FNV_BASIS = 0x14650FB0739D0383   # the basis the recorded reference hashes (tests/golden) were made with
STD_FNV_BASIS = 0xCBF29CE484222325
HEADER_BYTES = 607

_c = ctypes
_vp, _sz, _int = _c.c_void_p, _c.c_size_t, _c.c_int


def build():
    """(Re)builds the oracle; also oracle/_ref when the reference sources are present."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = _c.CDLL(LIB_PATH)
    lib.jo_fnv1a64.restype = _c.c_uint64
    lib.jo_fnv1a64.argtypes = [_vp, _sz, _c.c_uint64]
    lib.jo_max_jpeg_bytes.restype = _sz
    lib.jo_max_jpeg_bytes.argtypes = [_int, _int]
    lib.jo_gen_image.argtypes = [_int, _int, _int, _c.c_uint32, _vp]
    lib.jo_encode.argtypes = [_vp, _int, _int, _int, _vp, _sz, _c.POINTER(_sz)]
    lib.jo_decode.argtypes = [_vp, _sz, _int, _int, _int, _vp]
    lib.jo_encode_optimised.argtypes = [_vp, _int, _int, _int, _vp, _sz, _c.POINTER(_sz)]
    lib.jo_decode_file_tables.argtypes = [_vp, _sz, _int, _int, _int, _vp]
    lib.jo_optimal_huffman_spec.argtypes = [_vp, _sz, _vp]
    lib.jo_use_huffman_spec.argtypes = [_vp]
    lib.jo_use_huffman_spec.restype = None
    lib.jo_quant_tables.argtypes = [_int, _vp, _vp]
    lib.jo_set_mapper.argtypes = [_int]
    lib.jo_set_mapper.restype = None
    lib.jo_image_to_coefs.argtypes = [_vp, _int, _int, _vp, _vp, _vp]
    lib.jo_coefs_to_bit_lengths.argtypes = [_vp, _sz, _vp]
    lib.jo_coefs_to_scan.argtypes = [_vp, _sz, _vp, _sz, _c.POINTER(_sz), _c.POINTER(_c.c_uint64)]
    lib.jo_scan_to_coefs.argtypes = [_vp, _sz, _sz, _vp, _c.POINTER(_sz)]
    lib.jo_coefs_to_image.argtypes = [_vp, _int, _int, _vp, _vp, _vp]
    lib.jo_remove_stuffing.restype = _sz
    lib.jo_remove_stuffing.argtypes = [_vp, _sz, _sz]
    lib.jo_write_header.restype = _sz
    lib.jo_write_header.argtypes = [_vp, _int, _int, _vp, _vp]
    lib.jo_check_header.argtypes = [_vp, _sz, _c.POINTER(_int), _c.POINTER(_int)]
    lib.jo_rgb_to_ycbcr.argtypes = [_vp, _sz, _vp]
    lib.jo_ycbcr_to_rgb.argtypes = [_vp, _sz, _vp]
    lib.jo_fdct.argtypes = [_vp, _vp]
    lib.jo_idct.argtypes = [_vp, _vp]
    lib.jo_quantise.argtypes = [_vp, _vp, _vp]
    lib.jo_dequantise.argtypes = [_vp, _vp, _vp]
    lib.jo_cos_table.argtypes = [_vp, _vp]
    lib.jo_cos_table_runtime.argtypes = [_vp, _vp]
    lib.jo_zigzag_order.argtypes = [_vp]
    lib.jo_zigzag_walk.argtypes = [_vp]
    lib.jo_huffman_tables.argtypes = [_vp, _vp]
    return lib


LIB = _load()


def _p(a):
    return a.ctypes.data


def fnv(data, basis=RECORDED_FNV_BASIS):
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data)
    return "%016x" % LIB.jo_fnv1a64(_p(a), a.size, basis)


def gen_image(w, h, mode, seed=1234):
    """mode 0 = smooth gradients + small noise, mode 1 = uniform noise; (h, w, 3) uint8."""
    rgb = np.zeros((h, w, 3), np.uint8)
    LIB.jo_gen_image(w, h, mode, seed, _p(rgb))
    return rgb


def quant_tables(quality):
    lum, chrom = np.zeros(64, np.uint16), np.zeros(64, np.uint16)
    LIB.jo_quant_tables(quality, _p(lum), _p(chrom))
    return lum, chrom


def n_mcu(w, h):
    return ((w + 7) // 8) * ((h + 7) // 8)


def encode(rgb, quality):
    h, w = rgb.shape[:2]
    rgb = np.ascontiguousarray(rgb)
    cap = LIB.jo_max_jpeg_bytes(w, h)
    out = np.zeros(cap, np.uint8)
    n = _sz(0)
    rc = LIB.jo_encode(_p(rgb), w, h, quality, _p(out), cap, _c.byref(n))
    if rc:
        raise RuntimeError("jo_encode rc=%d" % rc)
    return out[: n.value].tobytes()


def encode_optimised(rgb, quality):
    """The file with the image's own Huffman tables (this build's optional mode; the reference has none)."""
    h, w = rgb.shape[:2]
    rgb = np.ascontiguousarray(rgb)
    cap = LIB.jo_max_jpeg_bytes(w, h)
    out = np.zeros(cap, np.uint8)
    n = _sz(0)
    rc = LIB.jo_encode_optimised(_p(rgb), w, h, quality, _p(out), cap, _c.byref(n))
    if rc:
        raise RuntimeError("jo_encode_optimised rc=%d" % rc)
    return out[: n.value].tobytes()


def decode_file_tables(jpeg, w, h, quality):
    """Decode with the Huffman tables in the file's header instead of the Annex K ones."""
    buf = np.frombuffer(jpeg, np.uint8)
    rgb = np.zeros((h, w, 3), np.uint8)
    rc = LIB.jo_decode_file_tables(_p(buf), buf.size, w, h, quality, _p(rgb))
    if rc:
        raise RuntimeError("jo_decode_file_tables rc=%d" % rc)
    return rgb


def set_mapper(mapper):
    """0: RGBToYCbCrMapper (default); 1: RGBToRGBMapper -- for the whole-image functions (encode, decode,
    image_to_coefs, coefs_to_scan, scan_to_coefs, coefs_to_image).  Not pinned by reference outputs."""
    LIB.jo_set_mapper(int(mapper))


def decode(jpeg, w, h, quality):
    buf = np.frombuffer(jpeg, np.uint8)
    rgb = np.zeros((h, w, 3), np.uint8)
    rc = LIB.jo_decode(_p(buf), buf.size, w, h, quality, _p(rgb))
    if rc:
        raise RuntimeError("jo_decode rc=%d" % rc)
    return rgb


def image_to_coefs(rgb, lum, chrom):
    h, w = rgb.shape[:2]
    rgb = np.ascontiguousarray(rgb)
    coefs = np.zeros((n_mcu(w, h), 3, 64), np.int16)
    rc = LIB.jo_image_to_coefs(_p(rgb), w, h, _p(lum), _p(chrom), _p(coefs))
    if rc:
        raise RuntimeError("jo_image_to_coefs rc=%d" % rc)
    return coefs


def coefs_to_bit_lengths(coefs):
    coefs = np.ascontiguousarray(coefs, np.int16)
    m = coefs.size // 192
    out = np.zeros(m * 3, np.uint32)
    rc = LIB.jo_coefs_to_bit_lengths(_p(coefs), m, _p(out))
    if rc:
        raise RuntimeError("jo_coefs_to_bit_lengths rc=%d" % rc)
    return out


def coefs_to_scan(coefs):
    """Stuffed, zero-padded scan bytes (no EOI) and the number of Huffman bits."""
    coefs = np.ascontiguousarray(coefs, np.int16)
    m = coefs.size // 192
    cap = 2 * ((m * 3 * 1660 + 7) // 8) + 16
    out = np.zeros(cap, np.uint8)
    n, bits = _sz(0), _c.c_uint64(0)
    rc = LIB.jo_coefs_to_scan(_p(coefs), m, _p(out), cap, _c.byref(n), _c.byref(bits))
    if rc:
        raise RuntimeError("jo_coefs_to_scan rc=%d" % rc)
    return out[: n.value].tobytes(), bits.value


def scan_to_coefs(scan_with_eoi, m):
    """Inverse of coefs_to_scan on stuffed bytes followed by FFD9."""
    buf = np.frombuffer(scan_with_eoi, np.uint8).copy()
    nlen = LIB.jo_remove_stuffing(_p(buf), 0, buf.size)
    coefs = np.zeros((m, 3, 64), np.int16)
    used = _sz(0)
    rc = LIB.jo_scan_to_coefs(_p(buf), nlen, m, _p(coefs), _c.byref(used))
    if rc:
        raise RuntimeError("jo_scan_to_coefs rc=%d" % rc)
    return coefs, used.value, nlen


def coefs_to_image(coefs, w, h, lum, chrom):
    coefs = np.ascontiguousarray(coefs, np.int16)
    rgb = np.zeros((h, w, 3), np.uint8)
    rc = LIB.jo_coefs_to_image(_p(coefs), w, h, _p(lum), _p(chrom), _p(rgb))
    if rc:
        raise RuntimeError("jo_coefs_to_image rc=%d" % rc)
    return rgb


def header(w, h, lum, chrom):
    out = np.zeros(HEADER_BYTES, np.uint8)
    n = LIB.jo_write_header(_p(out), w, h, _p(lum), _p(chrom))
    assert n == HEADER_BYTES
    return out.tobytes()


def rgb_to_ycbcr(rgb_triples):
    a = np.ascontiguousarray(rgb_triples, np.uint8)
    out = np.zeros_like(a)
    LIB.jo_rgb_to_ycbcr(_p(a), a.size // 3, _p(out))
    return out


def ycbcr_to_rgb(ycc_triples):
    a = np.ascontiguousarray(ycc_triples, np.uint8)
    out = np.zeros_like(a)
    LIB.jo_ycbcr_to_rgb(_p(a), a.size // 3, _p(out))
    return out


def fdct_blocks(samples_u8):
    """Level shift + forward DCT of (n, 64) u8 sample blocks -> (n, 64) float32 (index u + 8v)."""
    a = np.ascontiguousarray(samples_u8, np.uint8).reshape(-1, 64)
    shifted = (a.astype(np.int16) - 128).astype(np.int8)
    out = np.zeros(a.shape, np.float32)
    for i in range(a.shape[0]):
        LIB.jo_fdct(shifted[i].ctypes.data, out[i].ctypes.data)
    return out


def idct_blocks(dct_f32):
    """Inverse DCT + level unshift of (n, 64) float32 -> (n, 64) u8."""
    a = np.ascontiguousarray(dct_f32, np.float32).reshape(-1, 64)
    tmp = np.zeros(64, np.int8)
    out = np.zeros(a.shape, np.uint8)
    for i in range(a.shape[0]):
        LIB.jo_idct(a[i].ctypes.data, tmp.ctypes.data)
        out[i] = (tmp.astype(np.int16) + 128).astype(np.uint8)
    return out


def quantise_blocks(dct_f32, table):
    a = np.ascontiguousarray(dct_f32, np.float32).reshape(-1, 64)
    out = np.zeros(a.shape, np.int16)
    for i in range(a.shape[0]):
        LIB.jo_quantise(a[i].ctypes.data, _p(table), out[i].ctypes.data)
    return out


def dequantise_blocks(q_i16, table):
    a = np.ascontiguousarray(q_i16, np.int16).reshape(-1, 64)
    out = np.zeros(a.shape, np.float32)
    for i in range(a.shape[0]):
        LIB.jo_dequantise(a[i].ctypes.data, _p(table), out[i].ctypes.data)
    return out


def _build_tables_once():
    """The C library builds its cosine, Huffman and decode tables on first use, without a lock: one encode and one
    decode here, on the importing thread, so that callers may use the functions above from several threads."""
    tiny = gen_image(8, 8, 0, 1)
    decode(encode(tiny, 50), 8, 8, 50)


_build_tables_once()
