"""ctypes binding of include/jpegenc_amd.h."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libjpegenc_amd.so")
HEADER_BYTES = 607

ERR_NAMES = {1: "INVALID_ARG", 2: "HIP", 3: "CAPACITY", 4: "RANGE", 5: "BAD_HEADER", 6: "BAD_STREAM",
             7: "NO_EOI", 8: "NO_DEVICE"}

EXPORTS = [
    "jea_create", "jea_destroy", "jea_last_error", "jea_set_stream", "jea_set_colour_mapper", "jea_get_stream", "jea_synchronize",
    "jea_set_profiling", "jea_last_phase_ms", "jea_set_scratch_budget", "jea_set_huffman_optimised",
    "jea_quant_tables", "jea_write_header", "jea_max_jpeg_bytes", "jea_check_header", "jea_read_header", "jea_read_header_ex",
    "jea_encode_batch", "jea_encode_batch_async", "jea_encode_host",
    "jea_encode_host_batch", "jea_decode_host_batch", "jea_host_malloc", "jea_host_free",
    "jea_encode_batch_static_async", "jea_static_scratch_needed", "jea_static_scratch_bytes",
    "jea_stripe_transform", "jea_stripe_bits", "jea_stripe_finish",
    "jea_stripe_transform_async", "jea_stripe_bits_async", "jea_stripe_finish_async",
    "jea_decode_batch", "jea_decode_host",
    "jea_fdct_quant", "jea_fdct_quant_async", "jea_fdct_quant_planar_async", "jea_block_bit_lengths", "jea_entropy_encode", "jea_entropy_decode", "jea_remove_stuffing", "jea_idct_dequant",
    "jea_rgb_to_ycbcr", "jea_ycbcr_to_rgb",
    "jea_rgb_to_ycbcr_host", "jea_ycbcr_to_rgb_host", "jea_dct_forward_blocks", "jea_dct_inverse_blocks",
    "jea_quantise_blocks", "jea_dequantise_blocks", "jea_huffman_encode_blocks", "jea_stream_upload",
    "jea_huffman_decode_block", "jea_huffman_encode_rle", "jea_huffman_decode_rle",
    "jea_malloc", "jea_free", "jea_memcpy_h2d", "jea_memcpy_d2h",
]


class JeaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%s)" % (message, ERR_NAMES.get(code, code)))
        self.code = code
        self.message = message


_lib = None


def load_library():
    """Loads the HIP library; raises if it is missing -- there is no other implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `make -C jpeg-encoder_amd/csrc`" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, sz, u16, i32 = c.c_void_p, c.c_size_t, c.c_uint16, c.c_int
    lib.jea_create.argtypes = [i32, c.POINTER(vp)]
    lib.jea_destroy.argtypes = [vp]
    lib.jea_destroy.restype = None
    lib.jea_last_error.argtypes = [vp]
    lib.jea_last_error.restype = c.c_char_p
    lib.jea_set_stream.argtypes = [vp, vp]
    lib.jea_set_colour_mapper.argtypes = [vp, i32]
    lib.jea_set_huffman_optimised.argtypes = [vp, i32]
    lib.jea_get_stream.argtypes = [vp]
    lib.jea_get_stream.restype = vp
    lib.jea_synchronize.argtypes = [vp]
    lib.jea_set_scratch_budget.argtypes = [vp, sz]
    lib.jea_set_profiling.argtypes = [vp, i32]
    lib.jea_last_phase_ms.argtypes = [vp, vp, sz, c.POINTER(sz)]
    lib.jea_quant_tables.argtypes = [i32, vp, vp]
    lib.jea_quant_tables.restype = None
    lib.jea_write_header.argtypes = [vp, u16, u16, vp, vp]
    lib.jea_write_header.restype = None
    lib.jea_max_jpeg_bytes.argtypes = [u16, u16]
    lib.jea_max_jpeg_bytes.restype = sz
    lib.jea_check_header.argtypes = [vp, sz, c.POINTER(u16), c.POINTER(u16), c.POINTER(c.c_char_p)]
    lib.jea_read_header.argtypes = [vp, sz, c.POINTER(u16), c.POINTER(u16), vp, vp]
    lib.jea_read_header_ex.argtypes = [vp, sz, c.POINTER(u16), c.POINTER(u16), vp, vp, c.POINTER(i32)]
    lib.jea_encode_batch.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp, sz, vp, vp]
    lib.jea_encode_batch_async.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp, sz, vp, vp, vp]
    lib.jea_encode_host.argtypes = [vp, vp, u16, u16, vp, vp, vp, sz, c.POINTER(sz)]
    lib.jea_encode_host_batch.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp, sz, vp, vp]
    lib.jea_decode_host_batch.argtypes = [vp, vp, vp, vp, sz, u16, u16, vp, vp, vp]
    lib.jea_host_malloc.argtypes = [vp, sz, c.POINTER(vp)]
    lib.jea_host_free.argtypes = [vp, vp]
    lib.jea_encode_batch_static_async.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp, sz, vp, vp, vp]
    lib.jea_static_scratch_needed.argtypes = [sz, u16, u16]
    lib.jea_static_scratch_needed.restype = sz
    lib.jea_static_scratch_bytes.argtypes = []
    lib.jea_static_scratch_bytes.restype = sz
    lib.jea_stripe_transform.argtypes = [vp, vp, u16, u16, vp, vp, vp]
    lib.jea_stripe_bits.argtypes = [vp, vp, c.POINTER(c.c_uint64), c.POINTER(c.c_uint32)]
    lib.jea_stripe_finish.argtypes = [vp, c.c_uint64, i32, i32, c.c_uint32, u16, u16, vp, vp, vp, sz, c.POINTER(sz)]
    lib.jea_stripe_transform_async.argtypes = [vp, vp, u16, u16, vp, vp, vp]
    lib.jea_stripe_bits_async.argtypes = [vp, vp, vp]
    lib.jea_stripe_finish_async.argtypes = [vp, vp, c.c_uint32, c.c_uint32, u16, u16, vp, vp, vp, sz, vp, vp]
    lib.jea_decode_batch.argtypes = [vp, vp, vp, vp, sz, u16, u16, vp, vp, vp]
    lib.jea_decode_host.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp]
    lib.jea_fdct_quant.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp]
    lib.jea_fdct_quant_async.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp]
    lib.jea_fdct_quant_planar_async.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp, vp]
    lib.jea_block_bit_lengths.argtypes = [vp, vp, sz, sz, vp]
    lib.jea_entropy_encode.argtypes = [vp, vp, sz, sz, vp, sz, vp, vp]
    lib.jea_entropy_decode.argtypes = [vp, vp, vp, vp, sz, sz, vp]
    lib.jea_idct_dequant.argtypes = [vp, vp, sz, u16, u16, vp, vp, vp]
    lib.jea_remove_stuffing.argtypes = [vp, vp, sz, vp, c.POINTER(sz)]
    lib.jea_rgb_to_ycbcr.argtypes = [vp, vp, sz, vp]
    lib.jea_ycbcr_to_rgb.argtypes = [vp, vp, sz, vp]
    lib.jea_rgb_to_ycbcr_host.argtypes = [vp, vp, sz, vp]
    lib.jea_ycbcr_to_rgb_host.argtypes = [vp, vp, sz, vp]
    lib.jea_dct_forward_blocks.argtypes = [vp, vp, sz, vp]
    lib.jea_dct_inverse_blocks.argtypes = [vp, vp, sz, vp]
    lib.jea_quantise_blocks.argtypes = [vp, vp, sz, vp, vp]
    lib.jea_dequantise_blocks.argtypes = [vp, vp, sz, vp, vp]
    lib.jea_huffman_encode_blocks.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    lib.jea_stream_upload.argtypes = [vp, vp, sz]
    lib.jea_huffman_decode_block.argtypes = [vp, c.c_uint64, i32, c.POINTER(c.c_int16), vp, c.POINTER(c.c_uint64)]
    lib.jea_huffman_encode_rle.argtypes = [vp, c.c_int16, vp, vp, sz, i32, c.POINTER(c.c_uint32), vp, sz]
    lib.jea_huffman_decode_rle.argtypes = [vp, c.c_uint64, i32, c.POINTER(c.c_int16), vp, vp, c.POINTER(sz), c.POINTER(c.c_uint64)]
    lib.jea_malloc.argtypes = [vp, sz, c.POINTER(vp)]
    lib.jea_free.argtypes = [vp, vp]
    lib.jea_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    lib.jea_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    _lib = lib
    return lib


def quant_tables(quality):
    """Host-side helper (no GPU): Quantiser::Quantiser(quality) tables, natural order."""
    lib = load_library()
    lum, chrom = np.zeros(64, np.uint16), np.zeros(64, np.uint16)
    lib.jea_quant_tables(int(quality), lum.ctypes.data, chrom.ctypes.data)
    return lum, chrom


def write_header(w, h, lum, chrom):
    lib = load_library()
    out = np.zeros(HEADER_BYTES, np.uint8)
    lib.jea_write_header(out.ctypes.data, w, h, lum.ctypes.data, chrom.ctypes.data)
    return out.tobytes()


def max_jpeg_bytes(w, h):
    return load_library().jea_max_jpeg_bytes(w, h)


def check_header(jpeg):
    """Returns (ok, message, hdr_w, hdr_h)."""
    lib = load_library()
    buf = np.frombuffer(jpeg, np.uint8)
    w, h, msg = ctypes.c_uint16(0), ctypes.c_uint16(0), ctypes.c_char_p()
    rc = lib.jea_check_header(buf.ctypes.data, buf.size, ctypes.byref(w), ctypes.byref(h), ctypes.byref(msg))
    return rc == 0, (msg.value.decode() if msg.value else None), w.value, h.value


def read_header(jpeg):
    """(width, height, lum_q, chr_q) as the file states them; raises JeaError on a header the library rejects."""
    lib = load_library()
    buf = np.frombuffer(jpeg, np.uint8)
    w, h = ctypes.c_uint16(0), ctypes.c_uint16(0)
    lum, chrom = np.zeros(64, np.uint16), np.zeros(64, np.uint16)
    rc = lib.jea_read_header(buf.ctypes.data, buf.size, ctypes.byref(w), ctypes.byref(h), lum.ctypes.data,
                             chrom.ctypes.data)
    if rc:
        raise JeaError(rc, "jea_read_header: not a file of this format")
    return w.value, h.value, lum, chrom


def read_header_ex(jpeg):
    """(width, height, lum_q, chr_q, own_huffman_tables): read_header for files that may carry per-image Huffman tables."""
    lib = load_library()
    buf = np.frombuffer(jpeg, np.uint8)
    w, h, own = ctypes.c_uint16(0), ctypes.c_uint16(0), ctypes.c_int32(0)
    lum, chrom = np.zeros(64, np.uint16), np.zeros(64, np.uint16)
    rc = lib.jea_read_header_ex(buf.ctypes.data, buf.size, ctypes.byref(w), ctypes.byref(h), lum.ctypes.data,
                                chrom.ctypes.data, ctypes.byref(own))
    if rc:
        raise JeaError(rc, "jea_read_header_ex: not a file of this format")
    return w.value, h.value, lum, chrom, bool(own.value)


def static_scratch_bytes():
    """Size of the library's built-in scratch buffer (JEA_STATIC_SCRATCH_BYTES), read from the loaded library."""
    return load_library().jea_static_scratch_bytes()


def static_scratch_needed(n_images, w, h):
    return load_library().jea_static_scratch_needed(n_images, w, h)


def n_mcu(w, h):
    return ((w + 7) // 8) * ((h + 7) // 8)


class Context:
    """One jea_ctx on one GPU.  Device buffers are torch tensors on that GPU."""

    def __init__(self, device=0):
        import torch  # device memory only

        self.torch = torch
        self.lib = load_library()
        self.device = device
        self.handle = ctypes.c_void_p()
        rc = self.lib.jea_create(device, ctypes.byref(self.handle))
        if rc:
            raise JeaError(rc, self.lib.jea_last_error(None).decode())
        self.tdev = torch.device("cuda", device)

    def close(self):
        if self.handle:
            self.lib.jea_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _need(tensor, n_elements, what):
        """The C ABI takes raw pointers and trusts the sizes it is told; a tensor that is too short would make
        the kernels read or write past its end, so the binding refuses it here."""
        if tensor.numel() < int(n_elements):
            raise ValueError("%s holds %d elements, the call needs %d" % (what, tensor.numel(), int(n_elements)))

    def _check(self, rc):
        if rc:
            raise JeaError(rc, self.lib.jea_last_error(self.handle).decode())

    def synchronize(self):
        self._check(self.lib.jea_synchronize(self.handle))

    def set_scratch_budget(self, n_bytes):
        self._check(self.lib.jea_set_scratch_budget(self.handle, int(n_bytes)))

    def set_profiling(self, on):
        self._check(self.lib.jea_set_profiling(self.handle, 1 if on else 0))

    def last_phase_ms(self):
        """Durations (ms) of the phases of the last encode / decode call made while profiling was on."""
        ms = (ctypes.c_float * 8)()
        n = ctypes.c_size_t(0)
        self._check(self.lib.jea_last_phase_ms(self.handle, ms, 8, ctypes.byref(n)))
        return [float(ms[i]) for i in range(n.value)]

    @property
    def stream_handle(self):
        return self.lib.jea_get_stream(self.handle)

    def to_device(self, array):
        array = np.ascontiguousarray(array)
        if not array.flags.writeable:   # e.g. a view of a bytes object: torch wants memory it may write to
            array = array.copy()
        t = self.torch.from_numpy(array).to(self.tdev)
        self.torch.cuda.synchronize(self.tdev)
        return t

    def empty(self, n, dtype):
        return self.torch.empty(int(n), dtype=dtype, device=self.tdev)

    def set_colour_mapper(self, mapper):
        """0: RGB -> YCbCr (default); 1: RGB pass-through, all components on the luminance tables."""
        self._check(self.lib.jea_set_colour_mapper(self.handle, int(mapper)))

    def set_huffman_optimised(self, on):
        """Per-image Huffman tables in encode_batch / encode_host, read back from the files by decode_batch / decode_host."""
        self._check(self.lib.jea_set_huffman_optimised(self.handle, 1 if on else 0))

    # ---- stages ----
    def fdct_quant(self, d_rgb, n_images, w, h, lum, chrom):
        self._need(d_rgb, n_images * 3 * w * h, "d_rgb")
        coefs = self.empty(n_images * n_mcu(w, h) * 192, self.torch.int16)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_fdct_quant(self.handle, d_rgb.data_ptr(), n_images, w, h, lum.ctypes.data,
                                            chrom.ctypes.data, coefs.data_ptr()))
        return coefs

    def fdct_quant_async(self, d_rgb, n_images, w, h, lum, chrom, d_coefs):
        self._need(d_rgb, n_images * 3 * w * h, "d_rgb")
        self._need(d_coefs, n_images * n_mcu(w, h) * 192, "d_coefs")
        self._check(self.lib.jea_fdct_quant_async(self.handle, d_rgb.data_ptr(), n_images, w, h, lum.ctypes.data,
                                                  chrom.ctypes.data, d_coefs.data_ptr()))

    def fdct_quant_planar_async(self, d_rgb, n_images, w, h, lum, chrom, d_coefs_planar, d_masks):
        self._need(d_rgb, n_images * 3 * w * h, "d_rgb")
        self._need(d_coefs_planar, n_images * n_mcu(w, h) * 192, "d_coefs_planar")
        self._need(d_masks, n_images * n_mcu(w, h) * 3, "d_masks")
        self._check(self.lib.jea_fdct_quant_planar_async(self.handle, d_rgb.data_ptr(), n_images, w, h,
                                                         lum.ctypes.data, chrom.ctypes.data, d_coefs_planar.data_ptr(),
                                                         d_masks.data_ptr()))

    def idct_dequant(self, d_coefs, n_images, w, h, lum, chrom):
        self._need(d_coefs, n_images * n_mcu(w, h) * 192, "d_coefs")
        rgb = self.empty(n_images * 3 * w * h, self.torch.uint8)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_idct_dequant(self.handle, d_coefs.data_ptr(), n_images, w, h, lum.ctypes.data,
                                              chrom.ctypes.data, rgb.data_ptr()))
        return rgb

    def block_bit_lengths(self, d_coefs, n_images, mcus):
        self._need(d_coefs, n_images * mcus * 192, "d_coefs")
        out = self.empty(n_images * mcus * 3, self.torch.int16)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_block_bit_lengths(self.handle, d_coefs.data_ptr(), n_images, mcus, out.data_ptr()))
        return out

    def entropy_encode(self, d_coefs, n_images, mcus, capacity):
        self._need(d_coefs, n_images * mcus * 192, "d_coefs")
        out = self.empty(capacity, self.torch.uint8)
        offsets, sizes = np.zeros(n_images, np.uint64), np.zeros(n_images, np.uint64)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_entropy_encode(self.handle, d_coefs.data_ptr(), n_images, mcus, out.data_ptr(),
                                                capacity, offsets.ctypes.data, sizes.ctypes.data))
        return out, offsets, sizes

    def entropy_decode(self, d_scan, offsets, sizes, mcus):
        n = len(offsets)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        sizes = np.ascontiguousarray(sizes, np.uint64)
        coefs = self.empty(n * mcus * 192, self.torch.int16)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_entropy_decode(self.handle, d_scan.data_ptr(), offsets.ctypes.data,
                                                sizes.ctypes.data, n, mcus, coefs.data_ptr()))
        return coefs

    def remove_stuffing(self, scan_with_eoi):
        """bytes -> bytes: BitStream::removeStuffedBytes on scan data that ends with the EOI marker."""
        d_in = self.to_device(np.frombuffer(scan_with_eoi, np.uint8))
        d_out = self.empty(d_in.numel(), self.torch.uint8)
        n = ctypes.c_size_t(0)
        self._check(self.lib.jea_remove_stuffing(self.handle, d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), ctypes.byref(n)))
        return d_out[: n.value].cpu().numpy().tobytes()

    def rgb_to_ycbcr(self, d_rgb):
        out = self.torch.empty_like(d_rgb)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_rgb_to_ycbcr(self.handle, d_rgb.data_ptr(), d_rgb.numel() // 3, out.data_ptr()))
        return out

    def ycbcr_to_rgb(self, d_ycc):
        out = self.torch.empty_like(d_ycc)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_ycbcr_to_rgb(self.handle, d_ycc.data_ptr(), d_ycc.numel() // 3, out.data_ptr()))
        return out

    # ---- component level (host arrays) ----
    def dct_forward_blocks(self, samples):
        a = np.ascontiguousarray(samples, np.uint8).reshape(-1, 64)
        out = np.zeros(a.shape, np.float32)
        self._check(self.lib.jea_dct_forward_blocks(self.handle, a.ctypes.data, a.shape[0], out.ctypes.data))
        return out

    def dct_inverse_blocks(self, dct):
        a = np.ascontiguousarray(dct, np.float32).reshape(-1, 64)
        out = np.zeros(a.shape, np.uint8)
        self._check(self.lib.jea_dct_inverse_blocks(self.handle, a.ctypes.data, a.shape[0], out.ctypes.data))
        return out

    def quantise_blocks(self, dct, table):
        a = np.ascontiguousarray(dct, np.float32).reshape(-1, 64)
        out = np.zeros(a.shape, np.int16)
        self._check(self.lib.jea_quantise_blocks(self.handle, a.ctypes.data, a.shape[0], table.ctypes.data,
                                                 out.ctypes.data))
        return out

    def dequantise_blocks(self, q, table):
        a = np.ascontiguousarray(q, np.int16).reshape(-1, 64)
        out = np.zeros(a.shape, np.float32)
        self._check(self.lib.jea_dequantise_blocks(self.handle, a.ctypes.data, a.shape[0], table.ctypes.data,
                                                   out.ctypes.data))
        return out

    def huffman_encode_blocks(self, zigzag, prev_dc, is_lum):
        a = np.ascontiguousarray(zigzag, np.int16).reshape(-1, 64)
        dc = np.ascontiguousarray(prev_dc, np.int16)
        cls = np.ascontiguousarray(is_lum, np.uint8)
        lengths = np.zeros(a.shape[0], np.uint16)
        bits = np.zeros((a.shape[0], 52), np.uint32)
        self._check(self.lib.jea_huffman_encode_blocks(self.handle, a.ctypes.data, dc.ctypes.data, cls.ctypes.data,
                                                       a.shape[0], lengths.ctypes.data, bits.ctypes.data))
        return lengths, bits

    def stream_upload(self, stream):
        buf = np.frombuffer(stream, np.uint8)
        self._check(self.lib.jea_stream_upload(self.handle, buf.ctypes.data, buf.size))

    def huffman_decode_block(self, start_bit, is_lum, last_dc):
        dc = ctypes.c_int16(last_dc)
        zz = np.zeros(64, np.int16)
        end = ctypes.c_uint64(0)
        self._check(self.lib.jea_huffman_decode_block(self.handle, start_bit, int(is_lum), ctypes.byref(dc),
                                                      zz.ctypes.data, ctypes.byref(end)))
        return zz, end.value, dc.value

    def huffman_encode_rle(self, dc_difference, entries, is_lum):
        """entries: [(run, value), ...] -> (bit length, bits as uint32 words, MSB first)."""
        runs = np.array([e[0] for e in entries], np.uint16)
        values = np.array([e[1] for e in entries], np.int16)
        words = (22 + 26 * len(entries) + 31) // 32
        bits = np.zeros(words, np.uint32)
        length = ctypes.c_uint32(0)
        self._check(self.lib.jea_huffman_encode_rle(self.handle, int(dc_difference), runs.ctypes.data, values.ctypes.data,
                                                    len(entries), int(is_lum), ctypes.byref(length), bits.ctypes.data, words))
        return length.value, bits

    def huffman_decode_rle(self, start_bit, is_lum):
        """-> (dc difference, [(run, value), ...], end bit) from the uploaded stream."""
        runs, values = np.zeros(64, np.uint16), np.zeros(64, np.int16)
        dc, n, end = ctypes.c_int16(0), ctypes.c_size_t(0), ctypes.c_uint64(0)
        self._check(self.lib.jea_huffman_decode_rle(self.handle, start_bit, int(is_lum), ctypes.byref(dc), runs.ctypes.data,
                                                    values.ctypes.data, ctypes.byref(n), ctypes.byref(end)))
        return dc.value, [(int(runs[i]), int(values[i])) for i in range(n.value)], end.value

    # ---- stripes of one image ----
    def stripe_transform(self, d_rgb_stripe, w, stripe_h, lum, chrom):
        self._need(d_rgb_stripe, 3 * w * stripe_h, "d_rgb_stripe")
        dc = np.zeros(3, np.int16)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_stripe_transform(self.handle, d_rgb_stripe.data_ptr(), w, stripe_h, lum.ctypes.data,
                                                  chrom.ctypes.data, dc.ctypes.data))
        return [int(v) for v in dc]

    def stripe_bits(self, prev_dc):
        dc = np.asarray(prev_dc, np.int16)
        bits, head = ctypes.c_uint64(0), ctypes.c_uint32(0)
        self._check(self.lib.jea_stripe_bits(self.handle, dc.ctypes.data, ctypes.byref(bits), ctypes.byref(head)))
        return bits.value, head.value

    def stripe_finish(self, bits_before, is_first, has_next, next_head, image_w, image_h, lum, chrom, capacity, d_segment=None):
        seg = d_segment if d_segment is not None else self.empty(capacity, self.torch.uint8)
        capacity = min(capacity, seg.numel())
        n = ctypes.c_size_t(0)
        self._check(self.lib.jea_stripe_finish(self.handle, bits_before, int(is_first), int(has_next), next_head, image_w,
                                               image_h, lum.ctypes.data, chrom.ctypes.data, seg.data_ptr(), capacity,
                                               ctypes.byref(n)))
        return seg[: n.value]

    # the same steps with the facts in device memory (tensors on this GPU); nothing waits
    def stripe_transform_async(self, d_rgb_stripe, w, stripe_h, lum, chrom, d_last_dc):
        self._need(d_rgb_stripe, 3 * w * stripe_h, "d_rgb_stripe")
        self._need(d_last_dc, 4, "d_last_dc")
        self._check(self.lib.jea_stripe_transform_async(self.handle, d_rgb_stripe.data_ptr(), w, stripe_h, lum.ctypes.data,
                                                        chrom.ctypes.data, d_last_dc.data_ptr()))

    def stripe_bits_async(self, d_prev_dc, d_facts):
        self._need(d_facts, 2, "d_facts")
        self._check(self.lib.jea_stripe_bits_async(self.handle, d_prev_dc.data_ptr() if d_prev_dc is not None else None,
                                                   d_facts.data_ptr()))

    def stripe_finish_async(self, d_all_facts, index, n_stripes, image_w, image_h, lum, chrom, d_segment, d_segment_bytes, d_status):
        self._need(d_all_facts, 2 * n_stripes, "d_all_facts")
        self._check(self.lib.jea_stripe_finish_async(self.handle, d_all_facts.data_ptr(), index, n_stripes, image_w, image_h,
                                                     lum.ctypes.data, chrom.ctypes.data, d_segment.data_ptr(), d_segment.numel(),
                                                     d_segment_bytes.data_ptr(), d_status.data_ptr()))

    # ---- whole path ----
    def encode_batch(self, d_rgb, n_images, w, h, lum, chrom, capacity=None, d_out=None):
        if capacity is None:
            capacity = n_images * max_jpeg_bytes(w, h)
        if d_out is None:
            d_out = self.empty(capacity, self.torch.uint8)
        self._need(d_rgb, n_images * 3 * w * h, "d_rgb")
        self._need(d_out, capacity, "d_out")
        offsets, sizes = np.zeros(n_images, np.uint64), np.zeros(n_images, np.uint64)
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_encode_batch(self.handle, d_rgb.data_ptr(), n_images, w, h, lum.ctypes.data,
                                              chrom.ctypes.data, d_out.data_ptr(), capacity, offsets.ctypes.data,
                                              sizes.ctypes.data))
        return d_out, offsets, sizes

    def _need_encode_buffers(self, d_rgb, n_images, w, h, d_offsets, d_sizes, d_status):
        self._need(d_rgb, n_images * 3 * w * h, "d_rgb")
        self._need(d_offsets, n_images, "d_offsets")
        self._need(d_sizes, n_images, "d_sizes")
        self._need(d_status, 1, "d_status")

    def encode_batch_async(self, d_rgb, n_images, w, h, lum, chrom, d_out, d_offsets, d_sizes, d_status):
        self._need_encode_buffers(d_rgb, n_images, w, h, d_offsets, d_sizes, d_status)
        self._check(self.lib.jea_encode_batch_async(self.handle, d_rgb.data_ptr(), n_images, w, h, lum.ctypes.data,
                                                    chrom.ctypes.data, d_out.data_ptr(), d_out.numel(),
                                                    d_offsets.data_ptr(), d_sizes.data_ptr(), d_status.data_ptr()))

    def encode_batch_static_async(self, d_rgb, n_images, w, h, lum, chrom, d_out, d_offsets, d_sizes, d_status):
        """jea_encode_batch_async on the library's built-in scratch: no allocation inside the call."""
        self._need_encode_buffers(d_rgb, n_images, w, h, d_offsets, d_sizes, d_status)
        self._check(self.lib.jea_encode_batch_static_async(self.handle, d_rgb.data_ptr(), n_images, w, h,
                                                           lum.ctypes.data, chrom.ctypes.data, d_out.data_ptr(),
                                                           d_out.numel(), d_offsets.data_ptr(), d_sizes.data_ptr(),
                                                           d_status.data_ptr()))

    def encode_host(self, rgb, lum, chrom, capacity=None):
        h, w = rgb.shape[:2]
        rgb = np.ascontiguousarray(rgb, np.uint8)
        if capacity is None:
            capacity = max_jpeg_bytes(w, h)
        out = np.zeros(capacity, np.uint8)
        n = ctypes.c_size_t(0)
        self._check(self.lib.jea_encode_host(self.handle, rgb.ctypes.data, w, h, lum.ctypes.data, chrom.ctypes.data,
                                             out.ctypes.data, capacity, ctypes.byref(n)))
        return out[: n.value].tobytes()

    def host_array(self, n_bytes):
        """uint8 numpy array of n_bytes in pinned host memory (jea_host_malloc), which the GPU reads and writes directly;
        free with host_free(array) when done."""
        p = ctypes.c_void_p()
        self._check(self.lib.jea_host_malloc(self.handle, int(n_bytes), ctypes.byref(p)))
        a = np.ctypeslib.as_array((ctypes.c_uint8 * int(n_bytes)).from_address(p.value))
        self._host_arrays = getattr(self, "_host_arrays", {})
        self._host_arrays[a.ctypes.data] = p
        return a

    def host_free(self, array):
        p = self._host_arrays.pop(array.ctypes.data)
        self._check(self.lib.jea_host_free(self.handle, p))

    def encode_host_batch(self, rgb, lum, chrom, out=None, capacity=None):
        """rgb: [n, h, w, 3] uint8 in host memory (pageable or pinned) -> (out array, offsets, sizes)."""
        n, h, w = rgb.shape[:3]
        assert rgb.flags.c_contiguous and rgb.dtype == np.uint8
        if out is None:
            out = np.zeros(capacity if capacity is not None else n * ((max_jpeg_bytes(w, h) + 63) // 64 * 64), np.uint8)
        offsets, sizes = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        self._check(self.lib.jea_encode_host_batch(self.handle, rgb.ctypes.data, n, w, h, lum.ctypes.data, chrom.ctypes.data,
                                                   out.ctypes.data, out.size, offsets.ctypes.data, sizes.ctypes.data))
        return out, offsets, sizes

    def decode_host_batch(self, jpeg, offsets, sizes, w, h, lum, chrom, out=None):
        """jpeg: uint8 array in host memory holding the files -> [n, h, w, 3] uint8."""
        n = len(offsets)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        sizes = np.ascontiguousarray(sizes, np.uint64)
        if out is None:
            out = np.zeros(n * h * w * 3, np.uint8)
        self._check(self.lib.jea_decode_host_batch(self.handle, jpeg.ctypes.data, offsets.ctypes.data, sizes.ctypes.data, n, w, h,
                                                   lum.ctypes.data, chrom.ctypes.data, out.ctypes.data))
        return out.reshape(-1)[: n * h * w * 3].reshape(n, h, w, 3)

    def decode_batch(self, d_jpeg, offsets, sizes, w, h, lum, chrom):
        n = len(offsets)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        sizes = np.ascontiguousarray(sizes, np.uint64)
        rgb = self.empty(n * 3 * w * h, self.torch.uint8)
        if n:
            self._need(d_jpeg, int((offsets + sizes).max()), "d_jpeg")
        self.torch.cuda.synchronize(self.tdev)
        self._check(self.lib.jea_decode_batch(self.handle, d_jpeg.data_ptr(), offsets.ctypes.data, sizes.ctypes.data,
                                              n, w, h, lum.ctypes.data, chrom.ctypes.data, rgb.data_ptr()))
        return rgb

    def decode_host(self, jpeg, w, h, lum, chrom):
        buf = np.frombuffer(jpeg, np.uint8)
        rgb = np.zeros((h, w, 3), np.uint8)
        self._check(self.lib.jea_decode_host(self.handle, buf.ctypes.data, buf.size, w, h, lum.ctypes.data,
                                             chrom.ctypes.data, rgb.ctypes.data))
        return rgb
