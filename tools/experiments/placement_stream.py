#!/usr/bin/env python3
"""Does the speed of a plain streaming kernel depend on WHEN in a process its buffer was allocated?  N buffers of 1 GiB allocated one
after another (each its own hipMalloc), an in-place scale over each timed with events; then the same after the conv workspaces
of a HandsLight forward exist.  Companion of placement_variance.py.   usage (GPU box): python tools/placement_stream.py [n_buffers]"""
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
bufs = [torch.empty(1 << 28, dtype=torch.float32, device="cuda") for _ in range(n)]     # 1 GiB each
for b in bufs:
    b.zero_()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rates = []
for rnd in range(3):
    row = []
    for b in bufs:
        b.mul_(1.0)
        e0.record()
        for _ in range(10):
            b.mul_(1.0)
        e1.record()
        torch.cuda.synchronize()
        row.append(10 * 2 * b.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    rates.append(row)
best = [max(r[i] for r in rates) for i in range(n)]
print("TB/s (read + write) per 1 GiB buffer in allocation order, best of 3 rounds:")
for i in range(0, n, 8):
    print(f"  buffers {i:2d}-{i + 7:2d}: " + " ".join(f"{x:5.2f}" for x in best[i:i + 8]))
print(f"min {min(best):.2f}  max {max(best):.2f}  first 8 mean {sum(best[:8]) / 8:.2f}  last 8 mean {sum(best[-8:]) / 8:.2f}")
# pairs: read buffer i, write buffer j (the conv pattern: two different buffers), early pair vs late pair
def pair(i, j):
    torch.mul(bufs[i], 1.0, out=bufs[j])
    e0.record()
    for _ in range(10):
        torch.mul(bufs[i], 1.0, out=bufs[j])
    e1.record()
    torch.cuda.synchronize()
    return 10 * 2 * bufs[i].numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12
for i, j in ((0, 1), (2, 3), (n // 2, n // 2 + 1), (n - 2, n - 1), (0, n - 1)):
    print(f"copy {i:2d} -> {j:2d}: {max(pair(i, j) for _ in range(3)):.2f} TB/s")
