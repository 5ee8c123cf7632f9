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
