#!/usr/bin/env python3
"""Benchmarks of the baseline-JPEG hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5]

--config names the workload (the numbering of BASELINE.json's configs; c1 is the reference's own CPU case):
  c2 (default)  one 3840x2160 image encoded at q=75 per step -- the configuration the metric is quoted on.
                Steps alternate over `--streams` contexts and over a pool of distinct input images that is
                larger than the 256 MB Infinity Cache, so every step reads its picture from HBM.
  c3            1024 frames of 1920x1080 at q=90, sharded by image over the ranks; a step encodes the rank's
                share in one call.
  c4            one 16384x16384 image at q=75 as MCU-row stripes, one stripe per rank; a step is the whole
                protocol: transform, exchange of the DC predictors, bit strings, exchange of the bit counts,
                pack and stuff, gather of the segments on rank 0 (RCCL between GPUs).
  c5            1024 JPEG files of 1920x1080 at q=50 decoded, sharded by image; a step decodes the rank's share.
N > 1: one rank per GPU.  Under a launcher (torch.distributed.run sets RANK / WORLD_SIZE / MASTER_*) this process
is one rank.  Without one, `--gpus N` starts the N ranks itself: before anything touches a GPU the process runs
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child, lets rank 0's
JSON line through and exits with the child's code.  Every line says how many ranks took part (`ranks_seen`, a sum of
ones over the process group) and over what (`comm_backend`).  There is no data-path collective except c4's
exchanges; the reported value is all ranks' pixels over the slowest rank's time (weak scaling for c2, strong for
c3-c5, whose total work is fixed).

The default run (c2) also carries, under `other_measurements`, the other configurations at their full sizes: with one
rank c3 and c5 (1024 frames each), with several ranks c4 (the stripe protocol with its collectives, file hash
checked) -- each with the steps and warm-up of its own defaults; none of them is `value`.  The c4 measurement of a
multi-rank run and the closing barrier run under a time limit (JEA_C4_LIMIT_SECONDS, default 240): whatever becomes of
its collectives, rank 0 prints the headline line, with the measurement or with what happened to it.

Timing: after an untimed ramp and W warm-up steps, regions of exactly K steps each are timed, every one
bracketed by a barrier and a device synchronisation on both sides; as many regions as fit 50 ms (one, when K
steps take that long).  The value is taken from the median region (slowest rank of each region).

The JSON line also carries
  roofline      the dominant kernel's algorithmic bytes over its measured duration against the 8 TB/s HBM peak,
                what actually binds it (`bound`), and for the forward kernel the vector-instruction issue figures
  cpu_baseline  the CPU restatement of the reference encoder (oracle/), one thread, on a bounded sample, plus
                the same work image-parallel on the host's cores (rank 0, N=1)
"""
import argparse
import concurrent.futures
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0
K1_BYTES_PER_PIXEL = 9.0      # forward kernel: 3 B RGB read + 3 x int16 coefficients written
INVERSE_BYTES_PER_PIXEL = 9.0  # inverse kernel: 6 B coefficients read + 3 B RGB written
CPU_THREADS = 16              # the GPU box gives a one-GPU job 16 host cores
MIN_TIMED_SECONDS = 0.05
# vector-instruction issue cost on this chip (tools/valu_rates.hip, best of five to eight waves per SIMD): f32 add / mul / fma on
# vector-register or literal operands; everything else (conversions, fract, compares, perm, scalar operands)
VALU_FULL_RATE_NS = 1.03
VALU_HALF_RATE_NS = 1.72
SIMDS = 1024


def photo_like_image(width, height, seed):
    """Noise with a 1/f amplitude spectrum per channel, correlated across channels: the statistics of a
    photograph rather than of a gradient (mode 0) or of white noise (mode 1)."""
    import numpy as np

    rng = np.random.default_rng(seed)
    fy, fx = np.fft.fftfreq(height)[:, None], np.fft.fftfreq(width)[None, :]
    amp = 1.0 / np.maximum(np.hypot(fx, fy), 1.0 / max(width, height)) ** 1.1
    shaped = lambda: np.fft.ifft2(np.fft.fft2(rng.standard_normal((height, width))) * amp).real
    base = shaped()
    chans = [base + 0.3 * shaped() for _ in range(3)]
    return np.stack([np.clip(128 + 48 * c / c.std(), 0, 255) for c in chans], axis=-1).astype(np.uint8)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class Job:
    """Process group, device and the timing protocol shared by all configurations."""

    def __init__(self, args):
        import torch

        self.args = args
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            # a launcher's environment decides; n_gpus and ranks_seen in the line say what actually ran
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, self.world), file=sys.stderr)
        self.dist = None
        n_devices = torch.cuda.device_count()
        # One rank per GPU.  JEA_DIST_BACKEND=gloo rehearses the multi-process flow on a box with fewer GPUs
        # than ranks (ranks then share devices); the measured runs use RCCL ("nccl").
        self.backend = os.environ.get("JEA_DIST_BACKEND", "nccl")
        if self.backend == "nccl" and self.local_rank >= n_devices:
            raise SystemExit("bench: rank %d has no GPU of its own (%d visible)" % (self.local_rank, n_devices))
        self.device_index = self.local_rank % max(1, n_devices)
        # (JEA_FORCE_DIST: a process group also for a single rank, which lets one GPU rehearse the RCCL calls of c4)
        if self.world > 1 or os.environ.get("JEA_FORCE_DIST"):
            import torch.distributed as dist

            torch.cuda.set_device(self.device_index)
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device_index))
            else:
                dist.init_process_group(self.backend)
            self.dist = dist
        self.pkg = importlib.import_module("jpeg-encoder_amd")
        self.binding = self.pkg.binding
        self.dev = torch.device("cuda", self.device_index)
        self.comm_device = self.dev if self.backend == "nccl" else "cpu"
        # how many ranks really take part: every rank adds a one
        self.ranks_seen = int(round(self.sum_over_ranks(1)))
        self.comm_name = "none (one process)" if self.dist is None else \
            ("nccl (RCCL)" if self.backend == "nccl" else self.backend)

    def barrier(self, sync):
        if self.dist is not None:
            self.dist.barrier()
        sync()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.comm_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return value
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.comm_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, step, sync, steps=None, warmup=None, ramp=None):
        """Returns (seconds of the median region of `steps` steps, all region times)."""
        steps = self.args.steps if steps is None else steps
        warmup = self.args.warmup if warmup is None else warmup
        ramp = self.args.ramp_seconds if ramp is None else ramp
        # A parity gate before this left the GPU idle while the CPU encoded; its clocks take some milliseconds
        # of load to come back up.  Untimed steps until the ramp time has passed, then the W warm-up steps.
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < ramp:
            for _ in range(min(50, max(1, steps))):
                step()
            sync()
        for _ in range(warmup):
            step()
        regions = []
        want = 1
        while len(regions) < want:
            self.barrier(sync)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            self.torch.cuda.synchronize(self.dev)   # the device, i.e. every context's stream
            seconds = time.perf_counter() - t0
            sync()                                  # nothing left to wait for: the contexts report their errors
            regions.append(self.max_over_ranks(seconds))   # the job's time: the slowest rank's
            if len(regions) == 1:
                want = int(min(60, max(1, math.ceil(MIN_TIMED_SECONDS / max(regions[0], 1e-9)))))
        self.barrier(sync)
        ordered = sorted(regions)
        return ordered[len(ordered) // 2], regions

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None

    def emit(self, out):
        out["ranks_seen"] = self.ranks_seen
        out["comm_backend"] = self.comm_name
        if self.rank == 0:
            print(json.dumps(out), flush=True)

    def finish(self, out):
        self.close()
        self.emit(out)

    def finish_after(self, out, name, measure, seconds):
        """Runs measure() -- a measurement with collectives in it that this build could not run on several GPUs -- and
        the closing barrier on a thread with a time limit, so that the headline line is printed whatever becomes of it:
        out["other_measurements"][name] is its line, or says what failed.  A rank whose thread is still stuck when the
        limit passes prints (rank 0) and leaves without the barrier; every rank has the same limit."""
        import threading

        box = {}

        def work():
            try:
                self.torch.cuda.set_device(self.device_index)
                box["line"] = embedded(measure())
            except Exception as e:   # noqa: BLE001  (an operation the installed RCCL / torch refuses, on every rank alike)
                box["line"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                self.close()
                box["closed"] = True
            except Exception as e:   # noqa: BLE001
                box["close_error"] = "%s: %s" % (type(e).__name__, e)

        t = threading.Thread(target=work, daemon=True)
        t.start()
        t.join(seconds)
        extra = out.setdefault("other_measurements", {})
        if "line" in box:
            extra[name] = box["line"]
        else:
            extra[name] = {"error": "did not finish within %d s" % seconds}
        if t.is_alive() or "closed" not in box:
            if "line" in box and "error" not in box["line"]:
                extra[name]["note"] = "the closing barrier did not finish within the limit"
            self.emit(out)
            sys.stdout.flush()
            os._exit(0)
        self.emit(out)


def region_stats(regions):
    ordered = sorted(regions)
    return {"timed_regions": len(regions), "region_ms_min_median_max":
            [round(ordered[0] * 1e3, 4), round(ordered[len(ordered) // 2] * 1e3, 4), round(ordered[-1] * 1e3, 4)]}


def pmc_profile(name):
    """profiles/rNN_<name> of the latest round that has one (counter passes are taken with tools/refresh_profiles.sh)."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + name)))
    return json.load(open(found[-1])) if found else None


def cpu_encode_baseline(po, images, quality, n_single, threads):
    """The oracle on this host: one thread (what the reference does), then one image per thread."""
    w, h = images[0].shape[1], images[0].shape[0]
    t0 = time.perf_counter()
    for i in range(n_single):
        po.encode(images[i % len(images)], quality)
    single = n_single * w * h / (time.perf_counter() - t0) / 1e6
    n_all = 8 * threads
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda i: po.encode(images[i % len(images)], quality), range(n_all)))
    allc = n_all * w * h / (time.perf_counter() - t0) / 1e6
    return {"value": round(single, 2), "unit": "MPix/s", "cores": 1, "kind": "port",
            "sample": "%d encodes of %dx%d q=%d synthetic image(s), one thread, oracle/jpeg_oracle.c (a restatement of "
                      "the reference that tabulates its cosines; the reference itself is single-threaded)"
                      % (n_single, w, h, quality),
            "all_cores": {"value": round(allc, 2), "unit": "MPix/s", "cores": threads, "cpu_model": cpu_model(),
                          "host_cpus_visible": os.cpu_count(),
                          "sample": "%d encodes, one image per thread on %d threads" % (n_all, threads)}}


def forward_kernel_roofline(job, ctx, groups, whole, w, h, lum, chrom, images_per_launch=1, marginal_images=8):
    """Forward kernel alone, back to back on the library's stream, timed with HIP events on that stream.
    `groups`: device tensors of `images_per_launch` images each, launched in rotation; `whole`: one tensor
    that holds at least `marginal_images` images, or None."""
    torch, binding = job.torch, job.binding
    ext = torch.cuda.ExternalStream(ctx.stream_handle, device=job.dev)
    n = images_per_launch
    mcus = binding.n_mcu(w, h)
    d_rgb_pool = list(groups)
    for g in d_rgb_pool:
        assert g.numel() >= n * 3 * w * h
    many = marginal_images if (whole is not None and whole.numel() >= marginal_images * 3 * w * h and marginal_images > n) else 0
    d_coefs = torch.empty(max(n, many) * mcus * 192, dtype=torch.int16, device=job.dev)
    d_masks = torch.empty(max(n, many) * mcus * 3, dtype=torch.uint8, device=job.dev)

    def launches(images, reps):
        # the clocks drop while the GPU waits for host work (the parity checks before this): untimed launches first
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < job.args.ramp_seconds:
            for i in range(20):
                ctx.fdct_quant_planar_async(d_rgb_pool[i % len(d_rgb_pool)], images, w, h, lum, chrom, d_coefs, d_masks)
            ctx.synchronize()
        with torch.cuda.stream(ext):
            for i in range(3):
                ctx.fdct_quant_planar_async(d_rgb_pool[i % len(d_rgb_pool)], images, w, h, lum, chrom, d_coefs, d_masks)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                ctx.fdct_quant_planar_async(d_rgb_pool[i % len(d_rgb_pool)], images, w, h, lum, chrom, d_coefs, d_masks)
            e1.record()
        ctx.synchronize()
        return e0.elapsed_time(e1) / reps   # ms

    one = launches(n, 60)
    k_bytes = K1_BYTES_PER_PIXEL * n * w * h
    roof = {"kernel": "fdct_quant_rows_kernel", "achieved": round(k_bytes / (one * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(k_bytes / (one * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "avg_launch_us": round(one * 1e3, 2), "algorithmic_bytes_per_launch": int(k_bytes)}
    # The same kernel with several times the work per launch (images of the pool side by side): the difference is
    # the cost of one more image once the launch's fixed latency (first and last workgroups running on a
    # half-empty GPU) is paid.
    if many:
        with torch.cuda.stream(ext):
            e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2):
                ctx.fdct_quant_planar_async(whole, many, w, h, lum, chrom, d_coefs, d_masks)
            e2.record()
            for _ in range(12):
                ctx.fdct_quant_planar_async(whole, many, w, h, lum, chrom, d_coefs, d_masks)
            e3.record()
        ctx.synchronize()
        big = e2.elapsed_time(e3) / 12
        steady = max((big - one) / (many - n), 1e-6)      # per image
        roof["marginal_launch_images"] = many
        roof["marginal_us_per_step_of_work"] = round(steady * n * 1e3, 2)
        roof["marginal_frac"] = round(K1_BYTES_PER_PIXEL * w * h / (steady * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return roof, one


def host_batch_rates(job, ctx, imgs, w, h, quality, lum, chrom, po, n=64):
    """jea_encode_host_batch / jea_decode_host_batch on n pictures held in host memory, beside what the link itself does
    (pinned memory to device, measured here).  PCIe both ways inside the timed calls; never `value`."""
    import numpy as np

    torch = job.torch
    # the link: pinned host memory -> device, 256 MB at a time
    pin = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
    dev = torch.empty(256 << 20, dtype=torch.uint8, device=job.dev)
    dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize(job.dev)
    t0 = time.perf_counter()
    for _ in range(8):
        dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize(job.dev)
    h2d = 8 * pin.numel() / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    for _ in range(8):
        pin.copy_(dev, non_blocking=True)
    torch.cuda.synchronize(job.dev)
    d2h = 8 * pin.numel() / (time.perf_counter() - t0) / 1e9
    del pin, dev
    res = {"images": n, "link_pinned_h2d_GBps": round(h2d, 1), "link_pinned_d2h_GBps": round(d2h, 1),
           "link_ceiling_MPix/s_at_3_bytes_per_pixel": round(h2d * 1e3 / 3, 1)}
    per = 3 * w * h
    cap = n * (per // 8)           # smooth pictures: far below 3/8 byte per pixel
    for name, pinned in (("pageable", False), ("pinned", True)):
        if pinned:
            rgb, out, back = ctx.host_array(n * per), ctx.host_array(cap), ctx.host_array(n * per)
        else:
            rgb, out, back = np.empty(n * per, np.uint8), np.empty(cap, np.uint8), np.zeros(n * per, np.uint8)
            back[:] = 1      # the caller's pages exist before the timed call (first touches would be timed as copies)
        rgb = rgb.reshape(n, h, w, 3)
        for i in range(n):
            rgb[i] = imgs[i % len(imgs)]
        ctx.encode_host_batch(rgb[:4], lum, chrom, out=out)      # staging buffers and scratch exist after this
        t0 = time.perf_counter()
        _, off, size = ctx.encode_host_batch(rgb, lum, chrom, out=out)
        t_enc = time.perf_counter() - t0
        ok = all(out[int(off[i]): int(off[i] + size[i])].tobytes() == po.encode(imgs[i % len(imgs)], quality) for i in (0, n - 1))
        ctx.decode_host_batch(out, off[:4], size[:4], w, h, lum, chrom, out=back)
        t0 = time.perf_counter()
        got = ctx.decode_host_batch(out, off, size, w, h, lum, chrom, out=back)
        t_dec = time.perf_counter() - t0
        ok = ok and np.array_equal(got[n - 1], po.decode(out[int(off[n - 1]): int(off[n - 1] + size[n - 1])].tobytes(), w, h, quality))
        res[name + "_caller_memory"] = {
            "encode_MPix/s": round(n * w * h / t_enc / 1e6, 1), "encode_input_GBps": round(n * per / t_enc / 1e9, 1),
            "encode_fraction_of_link": round(n * per / t_enc / 1e9 / h2d, 3),
            "decode_MPix/s": round(n * w * h / t_dec / 1e6, 1), "decode_output_GBps": round(n * per / t_dec / 1e9, 1),
            "decode_fraction_of_link": round(n * per / t_dec / 1e9 / d2h, 3), "identical_to_oracle_sampled": bool(ok)}
        if pinned:
            for a in (rgb.reshape(-1), out, back):
                ctx.host_free(a)
    res["note"] = ("%d pictures of %dx%d in host memory per call; pieces of about 32 MB on two staging slots, link, kernels and the "
                   "way back overlapped; pinned = memory from jea_host_malloc, which the GPU reads and writes directly" % (n, w, h))
    return res


class ImagePool(list):
    """Device-resident images: pool[i] is image i; pool.whole is the one buffer that holds them back to back."""


def make_pool(ctx, images):
    import numpy as np

    whole = ctx.to_device(np.ascontiguousarray(images).reshape(-1))
    per = images[0].size
    pool = ImagePool(whole[i * per: (i + 1) * per] for i in range(len(images)))
    pool.whole = whole
    return pool


# ---------------------------------------------------------------------------------------------------------
def config_c2(job):
    import numpy as np
    import pyoracle as po  # generator + checker only; nothing timed on the GPU side calls it

    args, torch, binding = job.args, job.torch, job.binding
    W, H, QUALITY = 3840, 2160, 75
    ctxs = [job.pkg.Context(job.device_index) for _ in range(max(1, args.streams))]
    ctx = ctxs[0]
    lum, chrom = binding.quant_tables(QUALITY)
    batch = args.batch

    def generator(mode):
        return (lambda seed: photo_like_image(W, H, seed)) if mode == 2 else (lambda seed: po.gen_image(W, H, mode, seed))

    def capacity(mode):
        return batch * (3 * W * H // 2 + (1 << 20)) if mode == 0 else batch * binding.max_jpeg_bytes(W, H)

    def measure(mode, pool_images, steps, warmup, ramp, check):
        """Encode rate (MPix/s) with steps alternating over the contexts and over the pool of input images."""
        gen = generator(mode)
        n_pool = max(1, pool_images // batch) * batch
        with concurrent.futures.ThreadPoolExecutor(8) as tp:
            imgs = np.stack(list(tp.map(lambda i: gen(1234 + job.rank * 1000 + i), range(n_pool))))
        pool = make_pool(ctx, imgs)
        groups = [pool.whole[g * batch * 3 * W * H: (g + 1) * batch * 3 * W * H] for g in range(n_pool // batch)]
        cap = capacity(mode)
        outs = [dict(out=torch.empty(cap, dtype=torch.uint8, device=job.dev),
                     offsets=torch.zeros(batch, dtype=torch.int64, device=job.dev),
                     sizes=torch.zeros(batch, dtype=torch.int64, device=job.dev),
                     status=torch.zeros(16, dtype=torch.int32, device=job.dev)) for _ in ctxs]
        torch.cuda.synchronize(job.dev)
        turn = [0]

        def step():
            i = turn[0]
            turn[0] += 1
            o = outs[i % len(ctxs)]
            ctxs[i % len(ctxs)].encode_batch_async(groups[i % len(groups)], batch, W, H, lum, chrom, o["out"], o["offsets"],
                                                   o["sizes"], o["status"])

        def sync_all():
            for c in ctxs:
                c.synchronize()

        size0 = None
        if check:
            # parity gate before timing: files of pool images must equal the oracle's, byte for byte
            for g in sorted(set([0, len(groups) - 1])):
                turn[0] = g
                step()
                sync_all()
                o = outs[g % len(ctxs)]
                assert int(o["status"][0].item()) == 0, "encode reported status %d" % int(o["status"][0].item())
                off, size = int(o["offsets"][0].item()), int(o["sizes"][0].item())
                if o["out"][off: off + size].cpu().numpy().tobytes() != po.encode(imgs[g * batch], QUALITY):
                    raise SystemExit("bench: GPU JPEG differs from the oracle's; refusing to report a number")
                if g == 0:
                    size0 = size
            turn[0] = 0
        seconds, regions = job.timed(step, sync_all, steps, warmup, ramp)
        assert all(int(o["status"][0].item()) == 0 for o in outs)
        rate = job.sum_over_ranks(steps * batch * W * H) / seconds / 1e6
        return rate, seconds, regions, pool, imgs, size0, groups

    value, seconds, regions, pool, imgs, size0, groups = measure(args.mode, args.pool, args.steps, args.warmup, args.ramp_seconds, True)

    out = {
        "metric": "encode MPix/s (q=75)", "value": round(value, 1), "unit": "MPix/s", "n_gpus": job.world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(seconds / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic %s RGB8 (seeds 1234+), %d distinct images per rank resident in HBM (%d MB; the Infinity Cache "
                "holds 256 MB), read in rotation" % (("smooth", "noise", "photo-like")[args.mode], len(pool),
                                                     len(pool) * 3 * W * H >> 20),
        "config": {"workload": "single 3840x2160 RGB image encode at q=75 per step" if batch == 1 else
                   "%d x 3840x2160 RGB images encoded at q=75 per step" % batch,
                   "images_per_step": batch, "width": W, "height": H, "quality": QUALITY,
                   "jpeg_bytes_image0": size0, "byte_identical_to_oracle": True,
                   "streams": len(ctxs), "input_pool_images": len(pool), "sharding": "images by rank, no collective",
                   "untimed_ramp_seconds_before_warmup": args.ramp_seconds, **region_stats(regions)},
    }

    if job.world == 1 and not args.quick:
        # ---- the same measurement in other settings, each labelled; none of them is `value` ----
        others = {}
        lat_ctxs, ctxs[:] = ctxs[:], ctxs[:1]                     # one context: images strictly one after the other
        r, s = measure(args.mode, args.pool, max(50, args.steps // 4), 20, 0.05, False)[:2]
        others["one_stream_latency"] = {"MPix/s": round(r, 1), "us_per_image": round(s / max(50, args.steps // 4) * 1e6 / batch, 2),
                                        "note": "one context, every image waits for the one before"}
        ctxs[:] = lat_ctxs
        r = measure(args.mode, batch, max(50, args.steps // 4), 20, 0.05, False)[0]
        others["input_cached"] = {"MPix/s": round(r, 1), "note": "every step encodes the same image (it stays in the Infinity Cache)"}
        for mode, name, n_pool in ((2, "photo_like", 2), (1, "noise", 4)):
            if mode != args.mode:
                res = measure(mode, n_pool, max(50, args.steps // 4), 20, 0.05, True)
                r, sz = res[0], res[5]
                others[name] = {"MPix/s": round(r, 1), "jpeg_bytes_image0": sz, "input_pool_images": n_pool}
        # per-image Huffman tables (the optional mode; the reference has only the Annex K tables): two more passes over
        # the coefficients per image, smaller files
        for c in ctxs:
            c.set_huffman_optimised(True)
        try:
            res = measure(args.mode, args.pool, max(50, args.steps // 4), 20, 0.05, False)
            o = dict(out=torch.empty(capacity(args.mode), dtype=torch.uint8, device=job.dev), offsets=torch.zeros(batch, dtype=torch.int64, device=job.dev),
                     sizes=torch.zeros(batch, dtype=torch.int64, device=job.dev), status=torch.zeros(16, dtype=torch.int32, device=job.dev))
            ctx.encode_batch_async(res[6][0], batch, W, H, lum, chrom, o["out"], o["offsets"], o["sizes"], o["status"])
            ctx.synchronize()
            size_opt = int(o["sizes"][0].item())
            same = o["out"][: size_opt].cpu().numpy().tobytes() == po.encode_optimised(res[4][0], QUALITY)
            others["per_image_huffman_tables"] = {"MPix/s": round(res[0], 1), "jpeg_bytes_image0": size_opt,
                                                  "byte_identical_to_oracle": bool(same)}
        finally:
            for c in ctxs:
                c.set_huffman_optimised(False)
        out["other_measurements"] = others

    # ---- roofline of the fused colour+FDCT+quantise+zigzag kernel, live, on the library's stream ----
    roofline, one_ms = forward_kernel_roofline(job, ctx, groups, pool.whole, W, H, lum, chrom, batch, args.marginal_images)
    # HBM bytes and instruction counts per launch from the PMC counters, collected in separate rocprofv3 passes
    # over this same command (tools/refresh_profiles.sh); valid for the workload they were taken on
    pmc = pmc_profile("k1_issue.json") if (batch == 1 and args.mode == 0) else None
    roofline["traffic"] = pmc.get("hbm_bytes_per_launch") if pmc else None
    roofline["bound"] = "valu"
    roofline["bound_note"] = ("vector-instruction issue, not memory: the reference's operation order (every product and "
                              "sum of the transform rounded on its own, no FMA) fixes most of the instruction count; "
                              "`frac` is the 9 B/px of algorithmic traffic over the HBM peak all the same")
    if pmc:
        insts, waves = pmc["valu_insts_per_wave"], pmc["waves_per_launch"]
        per_inst = one_ms * 1e6 * SIMDS / (insts * waves)
        roofline["valu"] = {"insts_per_wave": insts, "waves_per_launch": waves, "simds": SIMDS,
                            "launch_ns_per_wave_instruction_per_simd": round(per_inst, 3),
                            "measured_issue_ns_full_rate": VALU_FULL_RATE_NS, "measured_issue_ns_half_rate": VALU_HALF_RATE_NS,
                            "half_rate_share_of_insts": pmc.get("half_rate_share"),
                            "issue_frac_single_launch": round(pmc["issue_ns_per_wave"] * waves / SIMDS / (one_ms * 1e6), 4),
                            "shader_clock_mhz_under_load": pmc.get("shader_clock_mhz"),
                            "issue_frac_note": "issue time by the per-instruction costs of a microbenchmark (tools/valu_rates.hip, "
                                               "taken at 2.30 GHz) over the measured time; the shader clock moves between 1.8 and "
                                               "2.3 GHz with the load, so a value near 1, or a little above, reads: at the issue bound"}
        if "marginal_us_per_step_of_work" in roofline:
            # a wave of a large launch does several units of work and pays its set-up once: fewer instructions per
            # image than in a single-image launch (counted in the 64-image pass of the same profile)
            many = pmc.get("counters_many_image_launch", {}).get("SQ_INSTS_VALU")
            images = pmc.get("many_image_launch", {}).get("images")
            insts_image = many / images if many and images else insts * waves
            roofline["valu"]["insts_per_image_in_large_launches"] = round(insts_image)
            roofline["valu"]["issue_frac_marginal"] = round(
                pmc["issue_ns_per_wave"] / insts * insts_image / SIMDS / (roofline["marginal_us_per_step_of_work"] * 1e3), 4)
    # in the pipeline: phase durations of whole encodes (events between the kernels of one context)
    ctx.set_profiling(True)
    o = dict(out=torch.empty(capacity(args.mode), dtype=torch.uint8, device=job.dev), offsets=torch.zeros(batch, dtype=torch.int64, device=job.dev),
             sizes=torch.zeros(batch, dtype=torch.int64, device=job.dev), status=torch.zeros(16, dtype=torch.int32, device=job.dev))
    acc = [0.0, 0.0, 0.0]
    reps = 40
    for i in range(reps + 5):
        ctx.encode_batch_async(groups[i % len(groups)], batch, W, H, lum, chrom, o["out"], o["offsets"], o["sizes"], o["status"])
        ph = ctx.last_phase_ms()
        if i >= 5:
            acc = [a + p for a, p in zip(acc, ph)]
    ctx.set_profiling(False)
    roofline["pipeline_phase_us"] = {"forward_transform": round(acc[0] / reps * 1e3, 2), "bit_strings": round(acc[1] / reps * 1e3, 2),
                                     "scans_pack_stuff": round(acc[2] / reps * 1e3, 2),
                                     "note": "one context, events between its kernels, one encode at a time"}
    out["roofline"] = roofline

    cpu = None
    if job.rank == 0 and job.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_encode_baseline(po, imgs, QUALITY, max(1, args.cpu_images), CPU_THREADS)
        # host to host: pageable caller memory in, file in caller memory out (PCIe both ways); never `value`
        t0 = time.perf_counter()
        n_h2h = 8
        for i in range(n_h2h):
            ctx.encode_host(imgs[i % len(imgs)], lum, chrom)
        out["host_to_host"] = {"MPix/s": round(n_h2h * W * H / (time.perf_counter() - t0) / 1e6, 1),
                               "note": "jea_encode_host: 24.9 MB of pageable host memory in over PCIe, file back out; one context"}
        out["host_to_host"]["batch"] = host_batch_rates(job, ctx, imgs, W, H, QUALITY, lum, chrom, po)
    out["cpu_baseline"] = cpu
    for c in ctxs:
        c.close()
    return out


# ---------------------------------------------------------------------------------------------------------
FRAMES, FW, FH = 1024, 1920, 1080


def rank_frames(job, po):
    """This rank's share of the 1024 distinct frames (smooth pictures with their own noise; every 64th is noise)."""
    import numpy as np

    sharding = importlib.import_module("jpeg-encoder_amd.sharding")
    mine = sharding.image_range(FRAMES if not job.args.frames else job.args.frames, job.world, job.rank)
    with concurrent.futures.ThreadPoolExecutor(CPU_THREADS) as tp:
        return np.stack(list(tp.map(lambda i: po.gen_image(FW, FH, 1 if i % 64 == 63 else 0, 1234 + i), mine))), mine


def config_c3(job, steps=None, warmup=None, with_cpu=True):
    import numpy as np
    import pyoracle as po

    args, torch, binding = job.args, job.torch, job.binding
    steps, warmup = (args.steps if steps is None else steps), (args.warmup if warmup is None else warmup)
    QUALITY = 90
    ctx = job.pkg.Context(job.device_index)
    lum, chrom = binding.quant_tables(QUALITY)
    frames, mine = rank_frames(job, po)
    n = len(mine)
    pool = make_pool(ctx, frames)
    sample = sorted(set([0, n // 2, n - 1]))
    want = {i: po.encode(frames[i], QUALITY) for i in sample}
    cap = n * (1 << 20) + (n // 64 + 1) * (4 << 20)
    o = dict(out=torch.empty(cap, dtype=torch.uint8, device=job.dev), offsets=torch.zeros(n, dtype=torch.int64, device=job.dev),
             sizes=torch.zeros(n, dtype=torch.int64, device=job.dev), status=torch.zeros(16, dtype=torch.int32, device=job.dev))

    def step():
        ctx.encode_batch_async(pool.whole, n, FW, FH, lum, chrom, o["out"], o["offsets"], o["sizes"], o["status"])

    step()
    ctx.synchronize()
    assert int(o["status"][0].item()) == 0
    offs, sizes = o["offsets"].cpu().numpy(), o["sizes"].cpu().numpy()
    for i in sample:
        if o["out"][int(offs[i]): int(offs[i] + sizes[i])].cpu().numpy().tobytes() != want[i]:
            raise SystemExit("bench: GPU JPEG %d differs from the oracle's; refusing to report a number" % i)
    seconds, regions = job.timed(step, ctx.synchronize, steps, warmup)
    value = job.sum_over_ranks(steps * n * FW * FH) / seconds / 1e6
    roofline, _ = forward_kernel_roofline(job, ctx, [pool.whole], None, FW, FH, lum, chrom, n)
    roofline["bound"] = "valu"
    roofline["traffic"] = None
    ctx.set_profiling(True)
    step()
    ph = ctx.last_phase_ms()
    ctx.set_profiling(False)
    roofline["pipeline_phase_us"] = {"forward_transform": round(ph[0] * 1e3, 1), "bit_strings": round(ph[1] * 1e3, 1),
                                     "scans_pack_stuff": round(ph[2] * 1e3, 1),
                                     "note": "the same batch as one pass on one stream, which is how a profiled call runs; the timed "
                                             "calls run as pieces of 64 frames on two streams of the context, transform beside entropy coding"}
    cpu = None
    if job.rank == 0 and job.world == 1 and with_cpu and not args.no_cpu_baseline:
        cpu = cpu_encode_baseline(po, frames[:64], QUALITY, 192, CPU_THREADS)
    out = {"metric": "encode MPix/s (q=90)", "value": round(value, 1), "unit": "MPix/s", "n_gpus": job.world,
           "steps": steps, "warmup": warmup, "ms_per_step": round(seconds / steps * 1e3, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic RGB8 frames (seeds 1234+i; every 64th uniform noise), resident in HBM",
           "config": {"workload": "batch of %d 1920x1080 frames at q=90, sharded by image, %d per rank, one encode call per step"
                      % (job.sum_over_ranks(n), n), "width": FW, "height": FH, "quality": QUALITY,
                      "byte_identical_to_oracle_sampled": len(sample), "sharding": "images by rank, no collective",
                      **region_stats(regions)},
           "roofline": roofline, "cpu_baseline": cpu}
    ctx.close()
    return out


def config_c5(job, steps=None, warmup=None, with_cpu=True):
    import numpy as np
    import pyoracle as po

    args, torch, binding = job.args, job.torch, job.binding
    steps, warmup = (args.steps if steps is None else steps), (args.warmup if warmup is None else warmup)
    QUALITY = 50
    ctx = job.pkg.Context(job.device_index)
    lum, chrom = binding.quant_tables(QUALITY)
    frames, mine = rank_frames(job, po)
    n = len(mine)
    d_rgb = ctx.to_device(frames.reshape(-1))
    d_jpeg, offs, sizes = ctx.encode_batch(d_rgb, n, FW, FH, lum, chrom, capacity=n * (1 << 19) + (n // 64 + 1) * (3 << 20))
    del d_rgb
    sample = sorted(set([0, n // 2, n - 1]))
    host = d_jpeg.cpu().numpy()
    files = {i: host[int(offs[i]): int(offs[i] + sizes[i])].tobytes() for i in sample}
    d_out = ctx.empty(n * 3 * FW * FH, torch.uint8)
    lib, handle = ctx.lib, ctx.handle
    offs_c, sizes_c = np.ascontiguousarray(offs, np.uint64), np.ascontiguousarray(sizes, np.uint64)

    def step():   # jea_decode_batch waits for the GPU itself (it reports the stream's errors)
        ctx._check(lib.jea_decode_batch(handle, d_jpeg.data_ptr(), offs_c.ctypes.data, sizes_c.ctypes.data, n, FW, FH,
                                        lum.ctypes.data, chrom.ctypes.data, d_out.data_ptr()))

    step()
    got = d_out.view(n, FH, FW, 3)
    for i in sample:
        if not np.array_equal(got[i].cpu().numpy(), po.decode(files[i], FW, FH, QUALITY)):
            raise SystemExit("bench: GPU decode of file %d differs from the oracle's; refusing to report a number" % i)
    seconds, regions = job.timed(step, ctx.synchronize, steps, warmup)
    value = job.sum_over_ranks(steps * n * FW * FH) / seconds / 1e6
    ctx.set_profiling(True)
    acc = [0.0] * 4
    for _ in range(5):
        step()
        acc = [a + p for a, p in zip(acc, ctx.last_phase_ms())]
    ctx.set_profiling(False)
    ph = [a / 5 for a in acc]
    stream_bytes = float(sizes.sum())
    px = n * FW * FH
    # HBM bytes of the inverse transform's kernels from the PMC passes over this configuration (tools/refresh_profiles.sh);
    # valid for the full 1024-frame call on one GPU
    pmc5 = pmc_profile("pmc_traffic_c5.json") if (job.world == 1 and n == 1024) else None
    inverse_traffic = None
    if pmc5:
        inverse_traffic = sum(pmc5["kernels"].get(k, {}).get("hbm_bytes_per_launch", 0) for k in ("idct_sparse_kernel", "idct_dequant_kernel"))
    # The inverse transform inside jea_decode_batch reads one 4-byte record per block (12 B per MCU) and the chunks
    # that exist (content dependent, not counted) and writes 3 B/px: that is what it must move at the least.
    inv_bytes = 3.0 * px + 12.0 * n * binding.n_mcu(FW, FH)
    std_bytes = INVERSE_BYTES_PER_PIXEL * px
    roofline = {"kernel": "idct_sparse_kernel (+ idct_dequant_kernel for the groups it lists)", "bound": "hbm",
                "achieved": round(inv_bytes / (ph[3] * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(inv_bytes / (ph[3] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": inverse_traffic, "avg_launch_us": round(ph[3] * 1e3, 1), "algorithmic_bytes_per_launch": int(inv_bytes),
                "standard_layout_equivalent_GBps": round(std_bytes / (ph[3] * 1e-3) / 1e9, 1),
                "bound_note": "3 B/px written + 12 B/MCU of block records read; the chunks of blocks with detail come on top. "
                              "The stores bind it, with the record reads mixed into them, not arithmetic (most MCUs of these frames "
                              "are one colour): the same rows written after one record load per MCU and nothing else run at "
                              "4.75 TB/s on this GPU, without the loads at 5.55, a linear fill at 6.85 (tools/store_patterns.hip). "
                              "6 B/px of coefficients read + 3 B/px written, the "
                              "figure for the standard layout, would read as `standard_layout_equivalent_GBps`",
                "other_kernels": {
                    "decode_write_kernel(+scan)": {"us": round(ph[2] * 1e3, 1), "algorithmic_bytes": int(stream_bytes + 6 * px),
                                                   "standard_layout_equivalent_GBps": round((stream_bytes + 6 * px) / (ph[2] * 1e-3) / 1e9, 1),
                                                   "note": "stream bytes read + 6 B/px of coefficients written is the standard layout's figure; "
                                                           "inside jea_decode_batch the pass stores one 4-byte record per block plus the chunks "
                                                           "that hold a non-zero AC coefficient, far fewer bytes, so the equivalent rate may exceed "
                                                           "the HBM peak; bound by the latency of the serial walks and by the L2's write transactions (a 16-byte "
                                                           "chunk or four records each), not by bytes"},
                    "decode_sync_kernel x2": {"us": round(ph[1] * 1e3, 1), "algorithmic_bytes": int(2 * stream_bytes),
                                              "note": "the stuffed bytes are dropped while the raw scan is staged: no unstuffing pass"}}}
    cpu = None
    if job.rank == 0 and job.world == 1 and with_cpu and not args.no_cpu_baseline:
        t0 = time.perf_counter()
        for i in sample:
            po.decode(files[i], FW, FH, QUALITY)
        single = len(sample) * FW * FH / (time.perf_counter() - t0) / 1e6
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(CPU_THREADS) as tp:
            list(tp.map(lambda k: po.decode(files[sample[k % len(sample)]], FW, FH, QUALITY), range(2 * CPU_THREADS)))
        allc = 2 * CPU_THREADS * FW * FH / (time.perf_counter() - t0) / 1e6
        cpu = {"value": round(single, 2), "unit": "MPix/s", "cores": 1, "kind": "port",
               "sample": "%d decodes of 1920x1080 q=50 files, one thread, oracle/jpeg_oracle.c" % len(sample),
               "all_cores": {"value": round(allc, 2), "unit": "MPix/s", "cores": CPU_THREADS, "cpu_model": cpu_model()}}
    out = {"metric": "decode MPix/s (q=50)", "value": round(value, 1), "unit": "MPix/s", "n_gpus": job.world,
           "steps": steps, "warmup": warmup, "ms_per_step": round(seconds / steps * 1e3, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "JPEG files of synthetic frames (seeds 1234+i; every 64th uniform noise) encoded at q=50 by this library, resident in HBM",
           "config": {"workload": "batched decode of %d 1920x1080 q=50 JPEG files, sharded by image, %d per rank, one decode call per step"
                      % (job.sum_over_ranks(n), n), "width": FW, "height": FH, "quality": QUALITY,
                      "jpeg_bytes_per_rank": int(stream_bytes), "bit_exact_with_oracle_sampled": len(sample),
                      "sharding": "images by rank, no collective", **region_stats(regions)},
           "roofline": roofline, "cpu_baseline": cpu}
    ctx.close()
    return out


# ---------------------------------------------------------------------------------------------------------
def config_c4(job, steps=None, warmup=None, with_cpu=True):
    import numpy as np
    import pyoracle as po

    args, torch, binding = job.args, job.torch, job.binding
    steps, warmup = (args.steps if steps is None else steps), (args.warmup if warmup is None else warmup)
    stripes = importlib.import_module("jpeg-encoder_amd.stripes")
    W = H = args.c4_size
    QUALITY = 75
    ctx = job.pkg.Context(job.device_index)
    lum, chrom = binding.quant_tables(QUALITY)
    rows = stripes.stripe_rows(H, job.world)
    if len(rows) != job.world:
        raise SystemExit("bench: %d ranks but only %d MCU rows" % (job.world, len(rows)))
    y0, y1 = rows[job.rank]
    image = po.gen_image(W, H, 0, 1234)            # every rank generates the picture and keeps its stripe
    d_rgb = ctx.to_device(image[y0:y1].reshape(-1))
    del image
    # the segment buffer is the caller's: smooth content stays far below 1.5 bytes per pixel (the worst case is 19)
    d_segment = torch.empty(3 * W * (y1 - y0) // 2 + (1 << 20), dtype=torch.uint8, device=job.dev)
    stripe = stripes.DeviceStripe(ctx, d_rgb, W, y1 - y0, lum, chrom, d_segment)
    on_device = job.backend == "nccl"
    result = {}

    def step():
        # everything up to the sizes is enqueued on the context's stream without a wait: three steps, the exchanges
        # between them (RCCL, device to device), the bit offset worked out on the GPU; collect_segments waits once
        sizes = stripes.encode_stripe_on_stream(stripe, job.rank, job.world, job.dist, W, H, on_device)
        result["file"] = stripes.collect_segments(stripe, sizes, job.rank, job.world, job.dist, on_device)

    step()
    checked = None
    if job.rank == 0:
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_hashes.json")))["rows"]
        row = [r for r in golden if (r["w"], r["h"], r["q"], r["mode"]) == (W, H, QUALITY, 0)]
        jpeg = result["file"].cpu().numpy().tobytes()
        if row:
            if len(jpeg) != row[0]["jpeg_bytes"] or po.fnv(jpeg) != row[0]["jpeg_fnv"]:
                raise SystemExit("bench: striped file differs from the recorded reference hash; refusing to report a number")
            checked = "recorded reference hash %s" % row[0]["jpeg_fnv"]
        else:
            if jpeg != po.encode(po.gen_image(W, H, 0, 1234), QUALITY):
                raise SystemExit("bench: striped file differs from the oracle's; refusing to report a number")
            checked = "oracle, byte for byte"
    # no time-based ramp here: a step contains collectives, so every rank must run the same number of steps
    seconds, regions = job.timed(step, ctx.synchronize, steps, warmup, ramp=0.0)
    value = steps * W * H / seconds / 1e6
    ctx.set_profiling(False)
    mcus = binding.n_mcu(W, y1 - y0)
    roofline, _ = forward_kernel_roofline(job, ctx, [d_rgb], None, W, y1 - y0, lum, chrom, 1)
    roofline["bound"] = "valu"
    roofline["traffic"] = None
    cpu = None
    if job.rank == 0 and job.world == 1 and with_cpu and not args.no_cpu_baseline:
        sample = po.gen_image(W, min(H, 1024), 0, 1234)
        t0 = time.perf_counter()
        po.encode(sample, QUALITY)
        cpu = {"value": round(sample.shape[0] * W / (time.perf_counter() - t0) / 1e6, 2), "unit": "MPix/s", "cores": 1,
               "kind": "port", "sample": "one encode of the top %d rows of the image, one thread, oracle/jpeg_oracle.c "
                                         "(one image: only the single-thread figure is meaningful)" % sample.shape[0]}
    out = {"metric": "encode MPix/s (q=75)", "value": round(value, 1), "unit": "MPix/s", "n_gpus": job.world,
           "steps": steps, "warmup": warmup, "ms_per_step": round(seconds / steps * 1e3, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic smooth RGB8 (seed 1234), each rank's stripe resident in HBM",
           "config": {"workload": "single %dx%d image at q=75 as %d MCU-row stripe(s), one per rank; per step: transform, "
                                  "all-gather of DC predictors, bit strings, all-gather of bit counts and head bits, pack and "
                                  "stuff at the bit offset computed on the GPU, all-gather of the segment sizes -- all enqueued on "
                                  "one stream --, one host wait, segments sent to rank 0 at their exact sizes" % (W, H, job.world),
                      "width": W, "height": H, "quality": QUALITY, "stripe_rows": y1 - y0, "file_checked_against": checked,
                      "collectives": "3 all-gathers of a few bytes + point-to-point sends of the segments per step (%s)" % job.comm_name,
                      "host_waits_per_step": 1,
                      **region_stats(regions)},
           "roofline": roofline, "cpu_baseline": cpu}
    ctx.close()
    return out


def embedded(line):
    """A configuration's own line as an entry of another line's other_measurements."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "data", "config", "roofline")
    return {k: line[k] for k in keep if k in line}


def run_as_ranks(n):
    """`--gpus n` outside a launcher: the n ranks run as a child job under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1).  This process has made no GPU call and makes none: it waits and hands on the
    child's exit code; the ranks inherit stdout, so rank 0's JSON line appears as if printed here."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--batch", type=int, default=1, help="c2: images per step")
    ap.add_argument("--streams", type=int, default=3,
                    help="c2: contexts (HIP streams) the steps alternate over; independent images overlap")
    ap.add_argument("--pool", type=int, default=16, help="c2: distinct input images the steps rotate over")
    ap.add_argument("--marginal-images", type=int, default=8,
                    help="c2: images of the large forward-kernel launch that gives the marginal cost per image (at most --pool)")
    ap.add_argument("--mode", type=int, default=0,
                    help="c2: 0 smooth gradients + noise (the recorded workload), 1 uniform noise, 2 photo-like 1/f noise")
    ap.add_argument("--frames", type=int, default=0, help="c3/c5: frames in the whole job (default 1024)")
    ap.add_argument("--c4-size", type=int, default=16384, help="c4: width = height of the image")
    ap.add_argument("--ramp-seconds", type=float, default=0.1,
                    help="untimed steps before the warm-up, for the GPU clocks (0 for counter-collection runs)")
    ap.add_argument("--quick", action="store_true", help="c2: only the headline measurement and the roofline")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="c2: leave out the c3/c5 (one rank) and c4 (several ranks) measurements under other_measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=48, help="c2: images of the single-thread CPU sample")
    args = ap.parse_args()
    defaults_of = {"c2": (2000, 200), "c3": (10, 2), "c4": (10, 2), "c5": (10, 2)}
    defaults = defaults_of[args.config]
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(run_as_ranks(args.gpus))      # nothing has touched a GPU yet; this process only relays
    job = Job(args)
    out = {"c2": config_c2, "c3": config_c3, "c4": config_c4, "c5": config_c5}[args.config](job)
    if args.config == "c2" and not args.quick and not args.no_other_configs:
        # the other configurations at full size, each at its own default steps / warm-up; never `value`
        extra = out.setdefault("other_measurements", {})
        if job.world == 1:
            extra["c3_batch_of_1024_1080p_q90"] = embedded(config_c3(job, *defaults_of["c3"], with_cpu=False))
            extra["c5_decode_of_1024_1080p_q50"] = embedded(config_c5(job, *defaults_of["c5"], with_cpu=False))
        else:
            limit = int(os.environ.get("JEA_C4_LIMIT_SECONDS", "240"))
            job.finish_after(out, "c4_16384x16384_as_stripes",
                             lambda: config_c4(job, *defaults_of["c4"], with_cpu=False), limit)
            return
    job.finish(out)


if __name__ == "__main__":
    main()
