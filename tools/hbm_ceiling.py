"""Practical HBM ceiling of the box: device-to-device copy, fill and reduction rates (what a streaming kernel can hope for)."""
import time, torch
dev = torch.device("cuda", 0)
n = 1 << 30
x = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
y = torch.empty_like(x)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
c = t(lambda: y.copy_(x)); print("copy  1 GiB: %.1f us -> %.2f TB/s (read + write)" % (c * 1e6, 2 * n / c / 1e12))
f = t(lambda: y.fill_(7)); print("fill  1 GiB: %.1f us -> %.2f TB/s (write)" % (f * 1e6, n / f / 1e12))
xi = x.view(torch.int32)
r = t(lambda: xi.sum()); print("sum   1 GiB: %.1f us -> %.2f TB/s (read)" % (r * 1e6, n / r / 1e12))
