#!/usr/bin/env python3
"""conv2 (3x3 Winograd) + conv3 (1x1 expand + identity + ReLU) of a layer1 bottleneck: the fused launch
(hands_bottleneck_wino_expand_f32) against the two launches (dev tool).  usage: python tools/bench_expand.py [B] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch

from hands_amd import _lib
from hands_amd.engine import ConvEngine
from hands_amd.packing import add_operand_form, pack_conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = _lib.lib()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(1)
c2 = pack_conv(torch.randn(64, 64, 3, 3, generator=g) / 24.0, torch.randn(64, generator=g), 1, 1, dev)
c3 = add_operand_form(pack_conv(torch.randn(256, 64, 1, 1, generator=g) / 8.0, torch.randn(256, generator=g), 1, 0, dev))
H = 56
t1 = torch.randn(B, H, H, 64, device=dev)
ident = torch.randn(B, H, H, 256, device=dev)
t2 = torch.empty(B, H, H, 64, device=dev)
out = torch.empty(B, H, H, 256, device=dev)
eng = ConvEngine()
eng.fuse_expand = True


def fused():
    eng.bottleneck_wino_expand(L, c2, c3, t1, t2, ident, out, B, H, H, st)


def separate():
    eng.conv(L, c2, t1, B, H, H, t2, True, st)
    eng.conv(L, c3, t2, B, H, H, out, True, st, res=ident)


for name, fn in (("fused", fused), ("two launches", separate), ("fused", fused), ("two launches", separate)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per bottleneck half at {B} images")
