"""What rate plain stores reach on this GPU: torch fills and a copy of a buffer the size of 1024 decoded 1920x1080 frames."""
import torch
n = 1024 * 1920 * 1080 * 3
x = torch.empty(n // 4, dtype=torch.int32, device="cuda")
y = torch.empty(n // 4, dtype=torch.int32, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, f, nbytes in (("fill", lambda: x.fill_(7), n), ("zero", lambda: x.zero_(), n), ("copy", lambda: y.copy_(x), 2 * n)):
    ms = timed(f)
    print(f"{name}: {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s", flush=True)
