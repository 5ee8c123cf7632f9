#!/usr/bin/env python3
"""Headline benchmark: baseline-JPEG encode throughput (MPix/s, q=75) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one encode of `--batch` (default 1) device-resident 3840x2160 RGB
images at q=75 through the whole hot path (fused colour+FDCT+quantise+zigzag
kernel, bit-length pass, scans, pack, byte stuffing, header and EOI); input and
output stay in HBM.  For N > 1 the driver launches one rank per GPU with
torch.distributed.run; images are sharded by rank, there is no data-path
collective, and the reported value is all ranks' pixels over the slowest rank's
time (weak scaling).

The JSON line also carries
  roofline      the fused FDCT kernel's algorithmic bytes (9 B/px) over its
                average launch duration, measured with HIP events on the
                library's stream, against the 8 TB/s HBM peak
  cpu_baseline  the CPU restatement of the reference encoder (oracle/), one
                thread, timed on this host on a bounded sample (rank 0, N=1)
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, QUALITY = 3840, 2160, 75
HBM_PEAK_GBS = 8000.0
K1_BYTES_PER_PIXEL = 9.0  # 3 B RGB read + 3 x int16 coefficient write


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1, help="images per step")
    ap.add_argument("--streams", type=int, default=3,
                    help="contexts (HIP streams) the steps alternate over; independent images overlap")
    ap.add_argument("--mode", type=int, default=0,
                    help="0 smooth gradients + noise (the default workload), 1 uniform noise, 2 photo-like 1/f noise")
    ap.add_argument("--ramp-seconds", type=float, default=0.1,
                    help="untimed steps before the warm-up, for the GPU clocks (0 for counter-collection runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=10, help="images of the CPU sample")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    dist = None
    n_devices = torch.cuda.device_count()
    # One rank per GPU.  JEA_DIST_BACKEND=gloo rehearses the multi-process flow on a box with fewer
    # GPUs than ranks (ranks then share devices); the measured runs use RCCL ("nccl").
    backend = os.environ.get("JEA_DIST_BACKEND", "nccl")
    if backend == "nccl" and local_rank >= n_devices:
        raise SystemExit("bench: rank %d has no GPU of its own (%d visible)" % (local_rank, n_devices))
    device_index = local_rank % max(1, n_devices)
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("jpeg-encoder_amd")
    binding = pkg.binding
    ctxs = [pkg.Context(device_index) for _ in range(max(1, args.streams))]
    ctx = ctxs[0]
    dev = ctx.tdev
    lum, chrom = binding.quant_tables(QUALITY)

    # synthetic input, resident in HBM before the timed region; every rank gets its own images
    import pyoracle as po  # generator + checker only; nothing timed on the GPU side calls it

    gen = (lambda seed: photo_like_image(W, H, seed)) if args.mode == 2 else (lambda seed: po.gen_image(W, H, args.mode, seed))
    imgs = np.stack([gen(1234 + rank * args.batch + i) for i in range(args.batch)])
    d_rgb = ctx.to_device(imgs.reshape(-1))
    cap = args.batch * (3 * W * H // 2 + (1 << 20)) if args.mode == 0 else args.batch * binding.max_jpeg_bytes(W, H)
    # every context has its own output and bookkeeping buffers; the input images are shared (read only)
    outs = [dict(out=torch.empty(cap, dtype=torch.uint8, device=dev),
                 offsets=torch.zeros(args.batch, dtype=torch.int64, device=dev),
                 sizes=torch.zeros(args.batch, dtype=torch.int64, device=dev),
                 status=torch.zeros(16, dtype=torch.int32, device=dev)) for _ in ctxs]
    d_out, d_offsets, d_sizes, d_status = (outs[0][k] for k in ("out", "offsets", "sizes", "status"))
    torch.cuda.synchronize(dev)
    turn = [0]

    def step():
        i = turn[0] % len(ctxs)
        turn[0] += 1
        o = outs[i]
        ctxs[i].encode_batch_async(d_rgb, args.batch, W, H, lum, chrom, o["out"], o["offsets"], o["sizes"], o["status"])

    def sync_all():
        for c in ctxs:
            c.synchronize()

    # parity gate before timing: the first image's file must equal the oracle's, byte for byte
    for _ in ctxs:
        step()
    sync_all()
    for o in outs:
        assert int(o["status"][0].item()) == 0, "encode reported status %d" % int(o["status"][0].item())
        assert torch.equal(o["out"][: int(o["sizes"][0].item())], d_out[: int(d_sizes[0].item())])
    off0, size0 = int(d_offsets[0].item()), int(d_sizes[0].item())
    gpu_file = d_out[off0: off0 + size0].cpu().numpy().tobytes()
    parity = gpu_file == po.encode(imgs[0], QUALITY)
    if not parity:
        raise SystemExit("bench: GPU JPEG differs from the oracle's; refusing to report a number")

    def barrier():
        if dist is not None:
            dist.barrier()
        sync_all()
        torch.cuda.synchronize(dev)

    # The parity gate above left the GPU idle while the CPU encoded; its clocks take some milliseconds of
    # load to come back up.  Untimed steps until 0.1 s has passed, then the W warm-up steps as asked.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(50):
            step()
        sync_all()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0   # this rank's time; the job's is the maximum over ranks (below)
    barrier()
    assert all(int(o["status"][0].item()) == 0 for o in outs)

    # whole-job throughput: all ranks' pixels over the slowest rank's time
    pixels_per_step = args.batch * W * H
    sharding = importlib.import_module("jpeg-encoder_amd.sharding")
    rate, elapsed = sharding.job_throughput(args.steps * pixels_per_step, elapsed, dist,
                                            dev if backend == "nccl" else "cpu")
    value = rate / 1e6

    # ---- roofline of the fused colour+FDCT+quantise+zigzag kernel, live, on the library's stream ----
    ext = torch.cuda.ExternalStream(ctx.stream_handle, device=dev)
    # the variant the fused encode runs: chunk-planar coefficients, all-zero chunks not stored
    d_coefs = torch.empty(args.batch * binding.n_mcu(W, H) * 192, dtype=torch.int16, device=dev)
    d_masks = torch.empty(args.batch * binding.n_mcu(W, H) * 3, dtype=torch.uint8, device=dev)
    reps = 50
    with torch.cuda.stream(ext):
        for _ in range(5):
            ctx.fdct_quant_planar_async(d_rgb, args.batch, W, H, lum, chrom, d_coefs, d_masks)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx.fdct_quant_planar_async(d_rgb, args.batch, W, H, lum, chrom, d_coefs, d_masks)
        e1.record()
    ctx.synchronize()
    k1_ms = e0.elapsed_time(e1) / reps
    k1_bytes = K1_BYTES_PER_PIXEL * pixels_per_step
    achieved = k1_bytes / (k1_ms * 1e-3) / 1e9
    # The same kernel with four times the work per launch: the difference is the cost of one more
    # step's worth of work once the launch's fixed latency (first and last workgroups running on a
    # half-empty GPU) is paid -- what a stream of back-to-back images sees.
    steady = None
    if args.batch * W * H <= 64 << 20:
        d_rgb4 = d_rgb.repeat(4)
        d_coefs4 = torch.empty(4 * d_coefs.numel(), dtype=torch.int16, device=dev)
        d_masks4 = torch.empty(4 * d_masks.numel(), dtype=torch.uint8, device=dev)
        with torch.cuda.stream(ext):
            for _ in range(3):
                ctx.fdct_quant_planar_async(d_rgb4, 4 * args.batch, W, H, lum, chrom, d_coefs4, d_masks4)
            e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e2.record()
            for _ in range(20):
                ctx.fdct_quant_planar_async(d_rgb4, 4 * args.batch, W, H, lum, chrom, d_coefs4, d_masks4)
            e3.record()
        ctx.synchronize()
        k4_ms = e2.elapsed_time(e3) / 20
        steady = max((k4_ms - k1_ms) / 3, 1e-6)
        del d_rgb4, d_coefs4, d_masks4
    # HBM bytes per launch from the PMC counters, collected in separate rocprofv3 passes on this same command
    # and corrected for gfx950 as profiles/r01_pmc_traffic.json describes; only valid for the workload it was taken on
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if args.batch == 1 and args.mode == 0 and os.path.exists(pmc):
        traffic = json.load(open(pmc))["kernels"].get("fdct_quant_kernel", {}).get("hbm_bytes_per_launch")
    roofline = {"kernel": "fdct_quant_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_us": round(k1_ms * 1e3, 2), "algorithmic_bytes_per_launch": int(k1_bytes)}
    if steady is not None:
        roofline["marginal_us_per_step_of_work"] = round(steady * 1e3, 2)
        roofline["marginal_frac"] = round(k1_bytes / (steady * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n = max(1, args.cpu_images)
        t0 = time.perf_counter()
        for i in range(n):
            po.encode(imgs[i % args.batch], QUALITY)
        dt = time.perf_counter() - t0
        cpu_baseline = {"value": round(n * W * H / dt / 1e6, 2), "unit": "MPix/s", "cores": 1, "kind": "port",
                        "sample": "%d x %dx%d q=%d encodes of the same synthetic image(s), single thread, "
                                  "oracle/jpeg_oracle.c (byte-identical to the reference)" % (n, W, H, QUALITY)}

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {
            "metric": "encode MPix/s (q=75)", "value": round(value, 1), "unit": "MPix/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic %s RGB8 (seed 1234+), resident in HBM" % ("smooth", "noise", "photo-like")[args.mode],
            "config": {"workload": "single 3840x2160 RGB image encode at q=75 per step" if args.batch == 1 else
                       "%d x 3840x2160 RGB images encoded at q=75 per step" % args.batch,
                       "images_per_step": args.batch, "width": W, "height": H, "quality": QUALITY,
                       "jpeg_bytes_image0": size0, "byte_identical_to_oracle": parity,
                       "streams": len(ctxs), "sharding": "images by rank, no collective",
                       "untimed_ramp_seconds_before_warmup": args.ramp_seconds},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
