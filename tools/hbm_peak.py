"""Reference points for the HBM-bound kernels: device-to-device copy, fill and read-reduce rates of this box."""
import torch
dev = "cuda:0"
n = 1 << 28  # 1 GiB of float32
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


ms = timeit(lambda: b.copy_(a))
print(f"copy   1 GiB: {ms:.3f} ms  {2 * n * 4 / ms / 1e9:.2f} TB/s (read + write)")
ms = timeit(lambda: b.fill_(1.0))
print(f"fill   1 GiB: {ms:.3f} ms  {n * 4 / ms / 1e9:.2f} TB/s (write)")
ms = timeit(lambda: a.sum())
print(f"reduce 1 GiB: {ms:.3f} ms  {n * 4 / ms / 1e9:.2f} TB/s (read)")
