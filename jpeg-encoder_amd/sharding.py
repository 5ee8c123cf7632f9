"""Image-batch sharding over ranks (one process per GPU).

Images are self-contained JPEG streams (the DC predictors and the bit stream start fresh for every
`encode` call of the reference, jpeg/src/encoder.cpp:16-20), so a batch shards with no data-path
collective: rank r takes a contiguous block of images.  The only communication is bookkeeping:
an all-gather of the per-image file sizes (optional) and the max/sum that turns per-rank timings
into one whole-job throughput.  Works with any torch.distributed backend (RCCL on GPUs, gloo on CPUs).
"""


def image_range(n_images, world_size, rank):
    """Contiguous block of image indices for `rank`; blocks differ in size by at most one."""
    if n_images < 0 or world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad sharding arguments")
    base, extra = divmod(n_images, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def owner_of(image, n_images, world_size):
    """Rank that image_range() gives `image` to."""
    base, extra = divmod(n_images, world_size)
    split = extra * (base + 1)
    if image < split:
        return image // (base + 1)
    return extra + (image - split) // base


def gather_file_sizes(local_sizes, n_images, dist, device="cpu"):
    """All ranks learn every image's file size.  local_sizes: sizes of this rank's images, in order."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (n_images + world - 1) // world
    mine = torch.zeros(per_rank, dtype=torch.int64, device=device)
    mine[: len(local_sizes)] = torch.as_tensor(list(local_sizes), dtype=torch.int64)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    sizes = []
    for r in range(world):
        sizes.extend(gathered[r][: len(image_range(n_images, world, r))].tolist())
    assert len(sizes) == n_images
    return sizes


def job_throughput(local_pixels, local_seconds, dist=None, device="cpu"):
    """Whole-job pixels per second: all ranks' pixels over the slowest rank's time."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_pixels / local_seconds, local_seconds
    import torch

    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    p = torch.tensor([float(local_pixels)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return float(p.item()) / float(t.item()), float(t.item())
