"""Reference points for the roofline: raw fill (write-only) and copy (read+write) bandwidth of this MI355X via torch."""
import time
import torch
n = 2 * 1024 ** 3  # 8 GiB of float32
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn, nbytes in (("fill (write only)", lambda: x.fill_(1.0), 4 * n), ("copy (read + write)", lambda: y.copy_(x), 8 * n),
                         ("read-reduce (read only)", lambda: x.sum(), 4 * n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {nbytes / dt / 1e9:.0f} GB/s")
