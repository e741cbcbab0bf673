#!/usr/bin/env python3
"""Linear layers through hands_conv2d_nhwc_f32 at head sizes, every row checked against fp64 (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
from hands_amd import _lib
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_linear
L = _lib.lib(); eng = ConvEngine(); eng.use_splitk = False
dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
for (M, K, N, relu, use_res, in_ps, out_ps) in [(128, 2160, 1024, True, False, 2160, 1024), (128, 1024, 1024, True, False, 1024, 1024),
                                                (256, 1024, 256, False, True, 1024, 2160), (128, 1024, 128, False, True, 1024, 2160),
                                                (512, 2048, 1024, True, False, 2048, 1024), (128, 64, 128, True, False, 64, 128)]:
    g = torch.Generator().manual_seed(M + K)
    w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    pc = pack_linear(w, b, dev)
    x = torch.randn(M, in_ps, generator=g).to(dev)
    out = torch.randn(M, out_ps, generator=g).to(dev)          # residual aliases the output slice (in-place update)
    before = out.clone()
    off = 16 if out_ps > N else 0
    eng.conv(L, pc, x, M, 1, 1, out, relu, st, res=out if use_res else None, in_ps=in_ps, out_ps=out_ps, res_ps=out_ps if use_res else None,
             out_off=off, res_off=off)
    torch.cuda.synchronize()
    ref = x[:, :K].double().cpu() @ w.double().t() + b.double()
    if use_res: ref = ref + before[:, off:off + N].double().cpu()
    if relu: ref = ref.clamp_min(0)
    got = out[:, off:off + N].double().cpu()
    untouched = torch.equal(out[:, :off], before[:, :off]) and torch.equal(out[:, off + N:], before[:, off + N:])
    print(f"M={M} K={K} N={N} relu={relu} res={use_res} out_ps={out_ps}: max err {(got - ref).abs().max().item():.2e}  rest untouched {untouched}")
