"""One large image encoded as MCU-row stripes, one stripe per rank / GPU.

The reference entropy-codes an image as one serial chain (jpeg/src/encoder.cpp:20-28): DC predictors and the
bit position carry from MCU to MCU.  Stripes of whole MCU rows break that chain in only two places, so three
steps with two tiny exchanges reproduce the serial file byte for byte:

    step 1  transform the stripe                   -> quantised DC of its last MCU
            exchange: DC of the stripe above       (3 x int16 per rank)
    step 2  Huffman-code the stripe                -> bit count, first 32 bits of its stream
            exchange: bit counts, head bits        (1 x uint64 + 1 x uint32 per rank)
    step 3  pack at the global bit offset, stuff   -> the stripe's segment of the file

The byte shared by two stripes belongs to the upper one, which completes it with the lower stripe's head
bits.  The file is the concatenation of the segments in stripe order.  The exchanges are all-gathers of a
few bytes (latency-bound; never a ring reduction); the segment gather is the only bulk transfer.

`Backend` is what a rank does locally (the HIP library on GPUs); `Group` is how ranks talk
(torch.distributed, or a loop inside one process for tests).
"""


def stripe_rows(image_height, n_stripes):
    """Pixel-row ranges [(y0, y1), ...]: whole MCU rows, as even as possible; empty stripes are dropped."""
    mcu_rows = (image_height + 7) // 8
    n = max(1, min(n_stripes, mcu_rows))
    base, extra = divmod(mcu_rows, n)
    out, row = [], 0
    for i in range(n):
        rows = base + (1 if i < extra else 0)
        out.append((row * 8, min(image_height, (row + rows) * 8)))
        row += rows
    return out


class HipBackend:
    """Stripe steps on one GPU through the C-ABI (jea_stripe_*)."""

    def __init__(self, ctx, d_rgb_stripe, width, stripe_height, lum, chrom, capacity=None, d_segment=None):
        self.ctx, self.d_rgb, self.w, self.h, self.lum, self.chrom = ctx, d_rgb_stripe, width, stripe_height, lum, chrom
        self.capacity, self.d_segment = capacity, d_segment   # optional: segment size bound, caller's output buffer

    def transform(self):
        return self.ctx.stripe_transform(self.d_rgb, self.w, self.h, self.lum, self.chrom)

    def bits(self, prev_dc):
        return self.ctx.stripe_bits(prev_dc)

    def finish(self, bits_before, is_first, has_next, next_head, image_w, image_h, on_device=False):
        """The stripe's segment of the file: bytes, or (on_device) a uint8 tensor on the GPU."""
        n_mcu = ((self.w + 7) // 8) * ((self.h + 7) // 8)
        # The worst case (every block at 1660 bits, every byte stuffed) is 19 bytes per pixel; a caller that
        # knows its pictures may bound the segment more tightly with `capacity`.
        capacity = self.capacity or 607 + 2 * ((n_mcu * 3 * 1660 + 7) // 8) + 16
        seg = self.ctx.stripe_finish(bits_before, is_first, has_next, next_head, image_w, image_h, self.lum,
                                     self.chrom, capacity, self.d_segment)
        return seg if on_device else bytes(seg.cpu().numpy().tobytes())


def encode_stripe(backend, rank, world, all_gather, image_w, image_h, on_device=False):
    """What one rank runs.  all_gather(list_of_ints) -> list over ranks of such lists.  Returns the segment."""
    last_dc = backend.transform()
    dcs = all_gather(list(last_dc))
    prev_dc = dcs[rank - 1] if rank > 0 else [0, 0, 0]
    bits, head = backend.bits(prev_dc)
    facts = all_gather([bits, head])
    bits_before = sum(f[0] for f in facts[:rank])
    has_next = rank + 1 < world
    next_head = facts[rank + 1][1] if has_next else 0
    if on_device:
        return backend.finish(bits_before, rank == 0, has_next, next_head, image_w, image_h, on_device=True)
    return backend.finish(bits_before, rank == 0, has_next, next_head, image_w, image_h)


def encode_in_process(backends, image_w, image_h):
    """All stripes in one process (one or several GPUs driven from one host thread): the exchanges are loops."""
    world = len(backends)
    dcs = [list(b.transform()) for b in backends]
    facts = [list(b.bits(dcs[r - 1] if r > 0 else [0, 0, 0])) for r, b in enumerate(backends)]
    segments = []
    for r, b in enumerate(backends):
        has_next = r + 1 < world
        segments.append(b.finish(sum(f[0] for f in facts[:r]), r == 0, has_next, facts[r + 1][1] if has_next else 0,
                                 image_w, image_h))
    return b"".join(segments)


def dist_all_gather(dist, device="cpu"):
    """all_gather of short integer lists over torch.distributed (RCCL on GPUs, gloo on CPUs)."""
    import torch

    def gather(values):
        mine = torch.tensor(values, dtype=torch.int64, device=device)
        out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(out, mine)
        return [t.tolist() for t in out]

    return gather


def gather_segments(segment, dist, device="cpu", dst=0):
    """Rank `dst` receives the whole file (segments in rank order); other ranks get None."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = dist_all_gather(dist, device)([len(segment)])
    longest = max(s[0] for s in sizes)
    mine = torch.zeros(longest, dtype=torch.uint8, device=device)
    mine[: len(segment)] = torch.frombuffer(bytearray(segment), dtype=torch.uint8).to(device)
    parts = [torch.zeros_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, parts, dst=dst)
    if rank != dst:
        return None
    return b"".join(bytes(parts[r][: sizes[r][0]].cpu().numpy().tobytes()) for r in range(world))


def gather_segments_device(segment, dist, device, dst=0):
    """gather_segments for segments that are uint8 tensors on the GPU: rank `dst` receives the file as one
    tensor on `device` (the GPU with RCCL; "cpu" with gloo), other ranks get None."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = dist_all_gather(dist, device)([int(segment.numel())])
    longest = max(s[0] for s in sizes)
    mine = torch.zeros(longest, dtype=torch.uint8, device=device)
    mine[: segment.numel()] = segment.to(device)
    parts = [torch.zeros_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, parts, dst=dst)
    if rank != dst:
        return None
    return torch.cat([parts[r][: sizes[r][0]] for r in range(world)])


# ---------------------------------------------------------------------------------------------------------
# The same protocol with the facts kept in device memory: a rank enqueues all three steps and the two exchanges
# between them on its context's stream and waits once, when it needs the segments' sizes on the host.
# ---------------------------------------------------------------------------------------------------------

class DeviceStripe:
    """One rank's stripe through the device-argument entry points (jea_stripe_*_async).  Holds the small device
    buffers the steps write: the last MCU's DC values, the stripe's {bit count, head bits}, the segment's size, the
    status word."""

    def __init__(self, ctx, d_rgb_stripe, width, stripe_height, lum, chrom, d_segment):
        import torch

        self.ctx, self.d_rgb, self.w, self.h, self.lum, self.chrom, self.d_segment = ctx, d_rgb_stripe, width, stripe_height, lum, chrom, d_segment
        dev = ctx.tdev
        self.d_dc = torch.zeros(4, dtype=torch.int16, device=dev)
        self.d_facts = torch.zeros(2, dtype=torch.int64, device=dev)
        self.d_size = torch.zeros(1, dtype=torch.int64, device=dev)
        self.d_status = torch.zeros(1, dtype=torch.int32, device=dev)
        # torch's view of the context's stream: collectives issued under it are ordered with the library's kernels
        self.stream = torch.cuda.ExternalStream(ctx.stream_handle, device=dev)
        self.keep = []   # gathered tensors that enqueued kernels still read
        torch.cuda.synchronize(dev)


def _gather_on_device(t, dist, on_device):
    """[world, t.numel()] on t's device: all ranks' copies of the small tensor t, in rank order.  With RCCL the
    exchange is device to device on the current stream; with another backend (gloo in the rehearsals) it goes
    through the host."""
    import torch

    if dist is None:
        return t.reshape(1, -1)
    world = dist.get_world_size()
    raw = t.reshape(-1).view(torch.uint8)      # bytes on the wire: not every backend knows 16-bit integers
    if on_device:
        out = torch.empty(world * raw.numel(), dtype=torch.uint8, device=t.device)
        dist.all_gather_into_tensor(out, raw)
        return out.view(t.dtype).reshape(world, -1)
    mine = raw.cpu()
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return torch.stack(parts).to(t.device).view(t.dtype).reshape(world, -1)


def encode_stripe_on_stream(stripe, rank, world, dist, image_w, image_h, on_device=True):
    """What one rank enqueues: transform, all-gather of the DC values, bit strings, all-gather of {bit count, head
    bits}, pack and stuff at the bit offset worked out on the GPU, all-gather of the segment sizes.  Returns the
    sizes ([world] int64, device) without waiting for anything (on_device: the exchanges are RCCL's)."""
    import torch

    ctx = stripe.ctx
    with torch.cuda.stream(stripe.stream):
        ctx.stripe_transform_async(stripe.d_rgb, stripe.w, stripe.h, stripe.lum, stripe.chrom, stripe.d_dc)
        all_dc = _gather_on_device(stripe.d_dc, dist, on_device)
        ctx.stripe_bits_async(all_dc[rank - 1] if rank > 0 else None, stripe.d_facts)
        all_facts = _gather_on_device(stripe.d_facts, dist, on_device)
        ctx.stripe_finish_async(all_facts, rank, world, image_w, image_h, stripe.lum, stripe.chrom, stripe.d_segment,
                                stripe.d_size, stripe.d_status)
        sizes = _gather_on_device(stripe.d_size, dist, on_device).reshape(-1)
    stripe.keep = [all_dc, all_facts, sizes]
    return sizes


def collect_segments(stripe, sizes, rank, world, dist, on_device=True, dst=0):
    """The one wait of a step: the sizes come to the host (and with them the status), then every segment travels
    to rank `dst` at its exact size, straight to its place in the file.  Returns the file (uint8 tensor; on the GPU
    with RCCL) on rank `dst`, None elsewhere."""
    import torch

    with torch.cuda.stream(stripe.stream):
        host_sizes = [int(v) for v in sizes.cpu()]          # waits for the stream
        status = int(stripe.d_status.item())
    if status != 0:
        raise RuntimeError("stripe encode reported status %d" % status)
    mine = stripe.d_segment[: host_sizes[rank]]
    if dist is None or world == 1:
        return mine
    if not on_device:
        mine = mine.cpu()
    if rank != dst:
        if on_device:   # RCCL: point-to-point operations go out as one group
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, dst)]):
                req.wait()
        else:
            dist.send(mine, dst=dst)
        return None
    file = torch.empty(sum(host_sizes), dtype=torch.uint8, device=mine.device)
    at = 0
    receives = []
    for r in range(world):
        part = file[at: at + host_sizes[r]]
        if r == dst:
            part.copy_(mine)
        else:
            receives.append((part, r))
        at += host_sizes[r]
    if on_device:
        pending = dist.batch_isend_irecv([dist.P2POp(dist.irecv, part, r) for part, r in receives])
    else:
        pending = [dist.irecv(part, src=r) for part, r in receives]
    for p in pending:
        p.wait()
    return file


def encode_in_process_on_device(stripes_of_image, image_w, image_h):
    """All stripes in one process through the device-argument entry points (DeviceStripe objects on contexts of one
    GPU): the exchanges are tensor copies on that GPU.  Returns the file as bytes."""
    import torch

    world = len(stripes_of_image)

    def wait_all():
        for s in stripes_of_image:
            s.ctx.synchronize()
        torch.cuda.synchronize()

    for s in stripes_of_image:
        s.ctx.stripe_transform_async(s.d_rgb, s.w, s.h, s.lum, s.chrom, s.d_dc)
    wait_all()
    all_dc = torch.stack([s.d_dc for s in stripes_of_image])
    torch.cuda.synchronize()
    for r, s in enumerate(stripes_of_image):
        s.ctx.stripe_bits_async(all_dc[r - 1] if r > 0 else None, s.d_facts)
    wait_all()
    all_facts = torch.stack([s.d_facts for s in stripes_of_image]).reshape(-1)
    torch.cuda.synchronize()
    for r, s in enumerate(stripes_of_image):
        s.ctx.stripe_finish_async(all_facts, r, world, image_w, image_h, s.lum, s.chrom, s.d_segment, s.d_size, s.d_status)
    wait_all()
    out = []
    for s in stripes_of_image:
        if int(s.d_status.item()) != 0:
            raise RuntimeError("stripe encode reported status %d" % int(s.d_status.item()))
        out.append(s.d_segment[: int(s.d_size.item())].cpu().numpy().tobytes())
    return b"".join(out)
