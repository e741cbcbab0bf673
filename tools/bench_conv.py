#!/usr/bin/env python3
"""Micro-benchmark of hands_conv2d_nhwc_f32 on representative hands_light layer shapes (dev tool).
usage: python tools/bench_conv.py [reps] [fp32|bf16x3]   (also prints the max error vs an fp64 convolution of a slice)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
from hands_amd import _lib
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_conv
import torch.nn.functional as F

SHAPES = [  # B, Cin, H, Cout, k, stride, pad, residual
    (512, 256, 14, 256, 3, 1, 1, False),
    (512, 128, 28, 128, 3, 1, 1, False),
    (512, 64, 56, 64, 3, 1, 1, False),
    (512, 1024, 14, 256, 1, 1, 0, False),
    (512, 256, 14, 1024, 1, 1, 0, True),
    (512, 128, 28, 512, 1, 1, 0, True),
    (512, 64, 56, 256, 1, 1, 0, True),
    (256, 256, 14, 256, 3, 1, 1, False),
    (256, 512, 7, 512, 3, 1, 1, False),
    (512, 512, 7, 2048, 1, 1, 0, True),
    (256, 2160, 1, 1024, 1, 1, 0, False),
]
if os.environ.get('HANDS_BENCH_ONE'):
    SHAPES = SHAPES[:1]
if os.environ.get('HANDS_BENCH_SHAPES'):      # "B,Cin,H,Cout,k,stride,pad,res;..."
    SHAPES = [tuple(int(v) for v in t.split(',')[:7]) + (t.split(',')[7] == '1',)
              for t in os.environ['HANDS_BENCH_SHAPES'].split(';') if t]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = ConvEngine()
eng.math = sys.argv[2] if len(sys.argv) > 2 else "fp32"
eng.stream_k = os.environ.get("HANDS_STREAMK") == "1"
eng.winograd = os.environ.get("HANDS_WINOGRAD", "1") == "1"
L = _lib.lib()
dev = "cuda"
stream = torch.cuda.current_stream().cuda_stream
tot_f = tot_t = 0.0
for (B, Cin, H, Cout, k, st, pad, use_res) in SHAPES:
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), st, pad, dev)
    x = torch.randn(B, H, H, Cin, device=dev)
    Ho = (H + 2 * pad - k) // st + 1
    out = torch.empty(B, Ho, Ho, Cout, device=dev)
    res = torch.randn(B, Ho, Ho, Cout, device=dev) if use_res else None
    for _ in range(2):
        eng.conv(L, pc, x, B, H, H, out, True, stream, res=res)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.conv(L, pc, x, B, H, H, out, True, stream, res=res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    tot_f += fl
    tot_t += ms
    nb = -(-B * Ho * Ho // 128) * -(-Cout // 128)
    nchk = min(B, 2)
    ref = F.conv2d(x[:nchk].permute(0, 3, 1, 2).double().cpu(), w.double(), pc.bias[:Cout].double().cpu(), stride=st, padding=pad)
    if res is not None:
        ref = ref + res[:nchk].permute(0, 3, 1, 2).double().cpu()
    err = (out[:nchk].permute(0, 3, 1, 2).double().cpu() - F.relu(ref)).abs().max().item()
    print(f"B{B:4d} {Cin:5d}->{Cout:5d} k{k} H{H:3d} res={int(use_res)} blocks={nb:6d} {ms * 1e3:9.1f} us {fl / ms / 1e9:7.1f} TF/s  max err vs fp64 {err:.2e}")
print(f"sum {tot_t:.3f} ms  {tot_f / tot_t / 1e9:.1f} TF/s")
