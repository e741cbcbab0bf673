#!/usr/bin/env python3
"""Stem micro-benchmark: conv_igemm stem + max-pool kernel vs the fused stem_pool kernel (B images 224x224)."""
import sys
import torch
sys.path.insert(0, ".")
from hands_amd import _lib
from hands_amd._lib import check, ptr
from hands_amd.hands_light import HandsLight
from hands_amd.packing import pack_conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
L = _lib.lib()
g = torch.Generator().manual_seed(0)
pc = pack_conv(torch.randn(64, 3, 7, 7, generator=g) / 12, torch.randn(64, generator=g), 2, 3, dev, cin_pad_to=4)
x4 = torch.randn(B, 224, 224, 4, generator=g).to(dev)
a = torch.empty(B, 112, 112, 64, device=dev)
o1 = torch.empty(B, 56, 56, 64, device=dev)
o2 = torch.empty(B, 56, 56, 64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def unfused():
    HandsLight._conv(L, pc, x4, B, 224, 224, a, 1, st)
    check(L.hands_maxpool3x3s2_nhwc_f32(ptr(a), ptr(o1), B, 112, 112, 64, st), "pool")


def fused():
    check(L.hands_stem_conv_maxpool_nhwc_f32(ptr(x4), ptr(pc.w), ptr(pc.bias), ptr(o2), B, 224, 224, 1, st), "fused")


for name, fn in (("stem + maxpool", unfused), ("fused stem_pool", fused)) * 2:
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:16s} B={B}: {ms:.3f} ms  ({2 * 64 * 147 * B * 112 * 112 / ms / 1e9:.1f} TFLOP/s of the useful conv FLOPs)")
print("equal:", torch.equal(o1, o2))
