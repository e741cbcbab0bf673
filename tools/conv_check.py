#!/usr/bin/env python3
"""One convolution, EVERY output compared with an fp32 torch convolution on the GPU, output pre-filled with NaN (dev tool).
usage: python tools/conv_check.py B,Cin,H,Cout,k,stride,pad,res [...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch, torch.nn.functional as F
from hands_amd import _lib
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_conv
L = _lib.lib(); eng = ConvEngine(); eng.stream_k = os.environ.get("HANDS_STREAMK") == "1"
dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[1:]:
    B, Cin, H, Cout, k, s, pad, use_res = [int(v) for v in spec.split(",")]
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pc = pack_conv(w, bias, s, pad, dev)
    x = torch.randn(B, H, H, Cin, device=dev)
    Ho = (H + 2 * pad - k) // s + 1
    res = torch.randn(B, Ho, Ho, Cout, device=dev) if use_res else None
    outs = []
    for fill in (float("nan"), 7.0):
        out = torch.full((B, Ho, Ho, Cout), fill, device=dev)
        eng.conv(L, pc, x, B, H, H, out, True, st, res=res)
        torch.cuda.synchronize()
        outs.append(out)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double().to(dev), bias.double().to(dev), stride=s, padding=pad)
    if res is not None: ref = ref + res.permute(0, 3, 1, 2).double()
    ref = F.relu(ref).permute(0, 2, 3, 1)
    err = (outs[0].double() - ref).abs()
    bad = torch.isnan(outs[0]).sum().item()
    print(f"{spec}: max err {torch.nan_to_num(err, nan=0.0).max().item():.2e}  NaN left {bad}  same for both fills {torch.equal(torch.nan_to_num(outs[0]), torch.nan_to_num(outs[1]))}")
    wrong = (torch.nan_to_num(err, nan=9.0) > 1e-3)
    if wrong.any():
        idx = wrong.view(-1, Cout).nonzero()
        m, c = idx[:, 0], idx[:, 1]
        tiles = (m // 128).unique().tolist()
        print("   wrong outputs:", wrong.sum().item(), " m-tiles:", tiles[:20], " rows in tile:", (m % 128).unique().tolist()[:40], " channels:", c.unique().tolist()[:40])
    if bad:
        idx = torch.isnan(outs[0]).nonzero()
        print("   first NaN at (b, y, x, c)", idx[0].tolist(), " last", idx[-1].tolist(), " distinct b", idx[:, 0].unique().numel(), "distinct c", idx[:, 3].unique().tolist()[:8])
