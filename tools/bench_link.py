#!/usr/bin/env python3
"""Micro-benchmark of hands_bottleneck_link_f32 against the two hands_conv2d_nhwc_f32 launches it replaces (dev tool).
usage: python tools/bench_link.py [images=512] [C1=64] [reps=20] [mode=both|fused|plain]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
from hands_amd import _lib
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
C1 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mode = sys.argv[4] if len(sys.argv) > 4 else "both"
H = 56
M = B * H * H
dev = "cuda"
L = _lib.lib()
eng = ConvEngine()
g = torch.Generator().manual_seed(0)
c3 = pack_conv(torch.randn(256, 64, 1, 1, generator=g) / 8, torch.randn(256, generator=g), 1, 0, dev)
c1 = pack_conv(torch.randn(C1, 256, 1, 1, generator=g) / 16, torch.randn(C1, generator=g), 1, 0, dev)
t2 = torch.randn(M, 64, device=dev)
ident = torch.randn(M, 256, device=dev)
out, t1 = torch.empty(M, 256, device=dev), torch.empty(M, C1, device=dev)
st = torch.cuda.current_stream().cuda_stream


def plain():
    eng.conv(L, c3, t2, B, H, H, out, True, st, res=ident)
    eng.conv(L, c1, out, B, H, H, t1, True, st)


def fused():
    eng.bottleneck_link(L, c3, c1, t2, ident, out, t1, M, st)


for name, fn in (("plain", plain), ("fused", fused)):
    if mode not in ("both", name):
        continue
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = 4.0 * M * (64 + 256 + 256 + C1 + (256 if name == "plain" else 0)) / 1e9
    fl = 2.0 * M * (64 * 256 + 256 * C1)
    print(f"{name}: {B} images M={M} C1={C1}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s  {gb / ms:.2f} TB/s of algorithmic traffic ({gb:.2f} GB)")
