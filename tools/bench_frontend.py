#!/usr/bin/env python3
"""Time the (f2) crop / KPE front-end on one GPU: samples/s and achieved HBM bandwidth.
Algorithmic bytes per sample: one read of the (3,224,224) fp32 image + three (3,224,224) fp32 writes
(img, r_img, l_img) = 2.408 MB; the joints / boxes / angles are < 1 KB."""
import sys

import torch

sys.path.insert(0, ".")
from hands_amd import HandsFrontEnd  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    img = torch.rand(B, 3, 224, 224, generator=g).to(dev)
    ctr = 0.6 * (torch.rand(B, 2, 1, 2, generator=g) * 2 - 1)
    j = (ctr + 0.3 * (torch.rand(B, 2, 21, 2, generator=g) * 2 - 1)).to(dev)
    K = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]]).repeat(B, 1, 1).to(dev)
    fe = HandsFrontEnd()
    jr, jl = j[:, 0].contiguous(), j[:, 1].contiguous()
    for _ in range(5):
        fe(img, jr, jl, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fe(img, jr, jl, K)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_ = B * 4 * 3 * 224 * 224 * 4
    print(f"front-end bz={B}: {ms:.3f} ms  {B / ms * 1e3:.0f} samples/s ({2 * B / ms * 1e3:.0f} hands/s)  "
          f"{bytes_ / ms / 1e6:.1f} GB/s algorithmic ({bytes_ / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
