"""The allocation-free encode entry point against the entry point it is a variant of.

jea_encode_batch_static_async must write exactly what jea_encode_batch_async writes -- files, offsets,
sizes, status -- on two calls in a row, the second with a smaller input (so that whatever the first call
left in the shared scratch buffer lies beyond and between what the second one uses), and must refuse a
job that does not fit its buffer without enqueuing anything.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _encode(ctx, entry, d_rgb, n, w, h, lum, chrom, cap):
    import torch

    out = torch.full((cap,), 0xA5, dtype=torch.uint8, device=ctx.tdev)
    offsets = torch.full((n,), -1, dtype=torch.int64, device=ctx.tdev)
    sizes = torch.full((n,), -1, dtype=torch.int64, device=ctx.tdev)
    status = torch.full((16,), 77, dtype=torch.int32, device=ctx.tdev)
    torch.cuda.synchronize(ctx.tdev)
    entry(d_rgb, n, w, h, lum, chrom, out, offsets, sizes, status)
    ctx.synchronize()
    return out.cpu().numpy(), offsets.cpu().numpy(), sizes.cpu().numpy(), int(status[0].item())


def test_static_scratch_encode_equals_allocating_encode(ctx, po, jea):
    b = jea.binding
    # first call: three noisy pictures (long bit strings everywhere in the scratch); second call: one small smooth one
    jobs = [(3, 328, 200, 90, 1), (1, 136, 72, 75, 0), (2, 1920, 1080, 75, 0)]
    for n, w, h, quality, mode in jobs[:2]:
        assert b.static_scratch_needed(n, w, h) <= b.static_scratch_bytes()
    assert b.static_scratch_needed(1, 1920, 1080) <= b.static_scratch_bytes()
    for n, w, h, quality, mode in jobs[:2]:
        lum, chrom = b.quant_tables(quality)
        imgs = np.stack([po.gen_image(w, h, mode, 500 + i) for i in range(n)])
        d_rgb = ctx.to_device(imgs.reshape(-1))
        cap = n * (b.max_jpeg_bytes(w, h) + 64)
        want = _encode(ctx, ctx.encode_batch_async, d_rgb, n, w, h, lum, chrom, cap)
        got = _encode(ctx, ctx.encode_batch_static_async, d_rgb, n, w, h, lum, chrom, cap)
        assert want[3] == 0 and got[3] == 0
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        assert np.array_equal(got[0], want[0])      # every byte of the buffer, written or not
        # and both are the oracle's files
        for i in range(n):
            off, size = int(got[1][i]), int(got[2][i])
            assert got[0][off: off + size].tobytes() == po.encode(imgs[i], quality)


def test_static_scratch_encode_refuses_what_does_not_fit(ctx, po, jea):
    import torch

    b = jea.binding
    n, w, h = 2, 1920, 1080
    assert b.static_scratch_needed(n, w, h) > b.static_scratch_bytes()
    lum, chrom = b.quant_tables(75)
    d_rgb = torch.zeros(n * 3 * w * h, dtype=torch.uint8, device=ctx.tdev)
    out = torch.full((1 << 20,), 0xA5, dtype=torch.uint8, device=ctx.tdev)
    offsets = torch.zeros(n, dtype=torch.int64, device=ctx.tdev)
    sizes = torch.zeros(n, dtype=torch.int64, device=ctx.tdev)
    status = torch.full((16,), 77, dtype=torch.int32, device=ctx.tdev)
    with pytest.raises(b.JeaError) as e:
        ctx.encode_batch_static_async(d_rgb, n, w, h, lum, chrom, out, offsets, sizes, status)
    assert e.value.code == 3 and "scratch" in e.value.message and str(b.static_scratch_bytes()) in e.value.message
    ctx.synchronize()
    assert int(status[0].item()) == 77 and bool((out == 0xA5).all())   # nothing ran


def test_static_scratch_encode_full_hd(ctx, po, jea):
    """The largest single picture the buffer is sized for."""
    b = jea.binding
    w, h, quality = 1920, 1080, 90
    lum, chrom = b.quant_tables(quality)
    img = po.gen_image(w, h, 0, 1234)
    d_rgb = ctx.to_device(img.reshape(-1))
    cap = 3 * w * h
    got = _encode(ctx, ctx.encode_batch_static_async, d_rgb, 1, w, h, lum, chrom, cap)
    want = _encode(ctx, ctx.encode_batch_async, d_rgb, 1, w, h, lum, chrom, cap)
    assert got[3] == 0 and np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    assert po.fnv(got[0][: int(got[2][0])].tobytes()) == "1f272a4bcc2f03f5"   # the recorded hash of this picture


def test_static_scratch_calls_from_two_contexts_queue(po, jea):
    """Two contexts on streams of their own, driven by two host threads, share the GPU's one buffer: the library
    orders the calls (each waits for the one that used the buffer before), so every file is still the oracle's."""
    import threading

    import torch

    b = jea.binding
    w, h, quality, n, rounds = 320, 240, 85, 2, 12
    lum, chrom = b.quant_tables(quality)
    imgs = [np.stack([po.gen_image(w, h, (t + i) & 1, 700 + 10 * t + i) for i in range(n)]) for t in range(2)]
    want = [[po.encode(im, quality) for im in group] for group in imgs]
    ctxs = [jea.Context(0) for _ in range(2)]
    cap = n * (b.max_jpeg_bytes(w, h) + 64)
    bufs = []
    for t, c in enumerate(ctxs):
        bufs.append([dict(rgb=c.to_device(imgs[t].reshape(-1)), out=torch.zeros(cap, dtype=torch.uint8, device=c.tdev),
                          offsets=torch.zeros(n, dtype=torch.int64, device=c.tdev), sizes=torch.zeros(n, dtype=torch.int64, device=c.tdev),
                          status=torch.zeros(16, dtype=torch.int32, device=c.tdev)) for _ in range(rounds)])
    torch.cuda.synchronize()
    errors = []

    def work(t):
        try:
            for r in range(rounds):      # all enqueued without a wait in between: the GPU work of the two threads interleaves
                o = bufs[t][r]
                ctxs[t].encode_batch_static_async(o["rgb"], n, w, h, lum, chrom, o["out"], o["offsets"], o["sizes"], o["status"])
            ctxs[t].synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=100)
    try:
        assert not errors, errors
        for t in range(2):
            for r in range(rounds):
                o = bufs[t][r]
                assert int(o["status"][0].item()) == 0
                host, off, size = o["out"].cpu().numpy(), o["offsets"].cpu().numpy(), o["sizes"].cpu().numpy()
                for i in range(n):
                    assert host[int(off[i]): int(off[i] + size[i])].tobytes() == want[t][i], (t, r, i)
    finally:
        for c in ctxs:
            c.close()
