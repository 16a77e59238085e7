"""Achievable HBM bandwidth on the box (SURVEY 8d asks for a stream-copy figure): device-to-device copies, a read-only
reduction and a fill at several sizes, torch kernels, event-timed."""
import torch
dev = torch.device('cuda:0')
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (41, 123, 246, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    tr = timeit(lambda: a.sum())
    tw = timeit(lambda: b.fill_(1.0))
    print('%5d MB: copy %.2f TB/s (read+write)   read-only sum %.2f TB/s   fill %.2f TB/s   (copy %.1f us)'
          % (mb, 2 * n * 4 / t / 1e12, n * 4 / tr / 1e12, n * 4 / tw / 1e12, t * 1e6))
