"""GPU parity of the component-level entry points (one 8x8 block-channel per element) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_dct_forward_blocks(ctx, po):
    rng = np.random.RandomState(1)
    blocks = np.concatenate([
        rng.randint(0, 256, (1500, 64)),
        np.full((1, 64), 128), np.zeros((1, 64)), np.full((1, 64), 255),
        (np.indices((8, 8)).sum(0) % 2 * 255).reshape(1, 64),            # checkerboard
        np.tile(np.arange(8) * 36, 8).reshape(1, 64),                    # horizontal ramp
    ]).astype(np.uint8)
    got = ctx.dct_forward_blocks(blocks)
    want = po.fdct_blocks(blocks)
    assert np.array_equal(_bits(got), _bits(want))                       # bit patterns, not values


def test_dct_inverse_blocks(ctx, po):
    rng = np.random.RandomState(2)
    dct = np.concatenate([
        rng.uniform(-300, 300, (800, 64)),
        rng.uniform(-2000, 2000, (400, 64)),                             # drives the clamp branches
        po.fdct_blocks(rng.randint(0, 256, (300, 64)).astype(np.uint8)),
        np.zeros((1, 64)),
    ]).astype(np.float32)
    dct[0, 0] = 127.5 * 8; dct[1, 0] = -128.5 * 8                        # values right at the clamp edges
    assert np.array_equal(ctx.dct_inverse_blocks(dct), po.idct_blocks(dct))


def test_quantise_blocks_random(ctx, po, jea):
    rng = np.random.RandomState(3)
    for quality in (10, 50, 75, 90, 100):
        lum, chrom = jea.binding.quant_tables(quality)
        for table in (lum, chrom):
            dct = np.concatenate([rng.uniform(-1100, 1100, (300, 64)), rng.normal(0, 8, (300, 64))]).astype(np.float32)
            assert np.array_equal(ctx.quantise_blocks(dct, table), po.quantise_blocks(dct, table)), quality


def test_quantise_blocks_adversarial(ctx, po):
    """Coefficients a few ulps either side of Q*(n + 0.5): the cases where a reciprocal multiply and the
    reference's rounded division can disagree, and where the +0.5 / floor step is decided by the last bit."""
    rng = np.random.RandomState(4)
    for table_val in list(range(1, 256)) + [257, 300, 1000, 4999, 65535]:
        table = np.full(64, table_val, np.uint16)
        n = rng.randint(-1100 // max(1, table_val) - 2, 1100 // max(1, table_val) + 3, 64 * 6)
        half = (n.astype(np.float64) + 0.5) * table_val
        x = half.astype(np.float32)
        ulps = rng.randint(-3, 4, x.shape).astype(np.int32)   # int32: the sum must stay 32 bits wide for the view
        x = (x.view(np.int32) + ulps).view(np.float32).reshape(-1, 64)
        assert x.shape == (6, 64) and np.isfinite(x).all()
        assert np.array_equal(ctx.quantise_blocks(x, table), po.quantise_blocks(x, table)), table_val
    # exact integers and exact halves
    table = np.arange(1, 65, dtype=np.uint16)
    x = (np.arange(-32, 32)[:, None] * table[None, :] * 0.5).astype(np.float32)
    assert np.array_equal(ctx.quantise_blocks(x, table), po.quantise_blocks(x, table))


def test_dequantise_blocks(ctx, po, jea):
    rng = np.random.RandomState(5)
    lum, chrom = jea.binding.quant_tables(30)
    q = rng.randint(-1024, 1025, (500, 64)).astype(np.int16)
    assert np.array_equal(_bits(ctx.dequantise_blocks(q, chrom)), _bits(po.dequantise_blocks(q, chrom)))


def test_colour_host_entry_points(ctx, po):
    rng = np.random.RandomState(6)
    px = rng.randint(0, 256, (5000, 3)).astype(np.uint8)
    out = np.zeros_like(px)
    ctx._check(ctx.lib.jea_rgb_to_ycbcr_host(ctx.handle, px.ctypes.data, 5000, out.ctypes.data))
    assert np.array_equal(out, po.rgb_to_ycbcr(px))
    ctx._check(ctx.lib.jea_ycbcr_to_rgb_host(ctx.handle, px.ctypes.data, 5000, out.ctypes.data))
    assert np.array_equal(out, po.ycbcr_to_rgb(px))


def _bitstring(lengths, bits):
    """Concatenates per-block MSB-first strings into bytes (zero padded)."""
    acc, n = 0, 0
    for ln, words in zip(lengths, bits):
        v = 0
        for w in words[: (int(ln) + 31) // 32]:
            v = (v << 32) | int(w)
        nw = (int(ln) + 31) // 32
        v >>= nw * 32 - int(ln)
        acc = (acc << int(ln)) | v
        n += int(ln)
    pad = -n % 8
    return (acc << pad).to_bytes((n + pad) // 8, "big"), n


def test_huffman_encode_blocks(ctx, po, jea):
    lum, chrom = jea.binding.quant_tables(90)
    rgb = po.gen_image(72, 40, 1, 21)
    coefs = po.image_to_coefs(rgb, lum, chrom)                           # (m, 3, 64)
    m = coefs.shape[0]
    prev = np.zeros((m, 3), np.int16)
    prev[1:] = coefs[:-1, :, 0]
    is_lum = np.tile(np.array([1, 0, 0], np.uint8), m)
    lengths, bits = ctx.huffman_encode_blocks(coefs.reshape(-1, 64), prev.reshape(-1), is_lum)
    assert np.array_equal(lengths.astype(np.uint32), po.coefs_to_bit_lengths(coefs))
    got, nbits = _bitstring(lengths, bits)
    scan, want_bits = po.coefs_to_scan(coefs)
    assert nbits == want_bits
    assert got == scan.replace(b"\xff\x00", b"\xff")


def test_huffman_decode_block_walk(ctx, po, jea):
    """Decode a small unstuffed stream block by block, as Encoder::decode drives EntropyEncoder::decode."""
    lum, chrom = jea.binding.quant_tables(75)
    rgb = po.gen_image(40, 24, 1, 9)
    coefs = po.image_to_coefs(rgb, lum, chrom)
    coefs[3, 0, 63] = 5                                                  # a block without EOB
    scan, _ = po.coefs_to_scan(coefs)
    seg = scan + b"\xff\xd9"
    want, used, nlen = po.scan_to_coefs(seg, coefs.shape[0])
    buf = np.frombuffer(seg, np.uint8).copy()
    n = po.LIB.jo_remove_stuffing(buf.ctypes.data, 0, buf.size)
    ctx.stream_upload(buf[:n].tobytes())
    bit, last = 0, [0, 0, 0]
    for mcu in range(coefs.shape[0]):
        for ch in range(3):
            zz, bit, last[ch] = ctx.huffman_decode_block(bit, ch == 0, last[ch])
            assert np.array_equal(zz, want[mcu, ch]), (mcu, ch)
    assert (bit + 7) // 8 == used
    assert want[3, 0, 63] == 0


def _run_length_form(zz, prev_dc):
    """entropy_encoder.cpp:94-120 restated for the test: (dc difference, [(run, value), ...])."""
    entries, zeros = [], 0
    for v in zz[1:]:
        if v == 0 and zeros < 15:
            zeros += 1
            continue
        entries.append((zeros, int(v)))
        zeros = 0
    if zz[63] == 0:
        while entries and entries[-1][1] == 0:
            entries.pop()
        entries.append((0, 0))
    return int(zz[0]) - int(prev_dc), entries


def test_huffman_final_encoding_on_run_length_form(ctx, po, jea):
    """jea_huffman_encode_rle / jea_huffman_decode_rle: the step a subclass of EntropyEncoder may replace
    (entropy_encoder.hpp:50-51).  The strings must be those of the block-level entry point, which the oracle pins."""
    lum, chrom = jea.binding.quant_tables(85)
    rgb = po.gen_image(48, 24, 1, 33)
    coefs = po.image_to_coefs(rgb, lum, chrom)
    coefs[2, 1, 63] = -7                                                 # a block without EOB
    coefs[4, 0, 1:] = 0
    coefs[4, 0, 40] = 3                                                  # zero runs of more than 16
    m = coefs.shape[0]
    prev = np.zeros((m, 3), np.int16)
    prev[1:] = coefs[:-1, :, 0]
    is_lum = np.tile(np.array([1, 0, 0], np.uint8), m)
    lengths, bits = ctx.huffman_encode_blocks(coefs.reshape(-1, 64), prev.reshape(-1), is_lum)
    flat, prevs = coefs.reshape(-1, 64), prev.reshape(-1)
    for i in range(flat.shape[0]):
        diff, entries = _run_length_form(flat[i], prevs[i])
        n, words = ctx.huffman_encode_rle(diff, entries, is_lum[i])
        assert n == lengths[i], i
        assert np.array_equal(words[: (n + 31) // 32], bits[i][: (n + 31) // 32]), i
    # and back: the stream of all blocks, entry lists as they were encoded
    scan, _ = po.coefs_to_scan(coefs)
    ctx.stream_upload(scan.replace(b"\xff\x00", b"\xff") + b"\x00\x00")
    bit = 0
    for i in range(flat.shape[0]):
        diff, entries = _run_length_form(flat[i], prevs[i])
        got_diff, got_entries, bit = ctx.huffman_decode_rle(bit, is_lum[i])
        assert (got_diff, got_entries) == (diff, entries), i
    assert bit == int(lengths.sum())
    # entries are taken as given: a zero value with a run other than 0 or 15 is the reference's runtime error
    with pytest.raises(jea.binding.JeaError) as e:
        ctx.huffman_encode_rle(0, [(3, 0), (0, 0)], True)
    assert e.value.code == 4 and e.value.message == "Invalid runtime encoding encountered."
    with pytest.raises(jea.binding.JeaError):
        ctx.huffman_encode_rle(0, [(16, 5)], True)
    # explicit zero-run entries in the tail and no end-of-block entry: encoded all the same
    n, words = ctx.huffman_encode_rle(-3, [(15, 0), (15, 0)], False)
    assert n == 2 + 2 + 10 + 10   # chrominance: DC size-2 code (2 bits) + 2 amplitude bits, two 10-bit ZRL codes
