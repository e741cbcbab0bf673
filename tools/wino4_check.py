#!/usr/bin/env python3
"""hands_conv3x3_winograd4_f32 (F(4x4,3x3), csrc/conv_wino4.hip): every output against an fp64 convolution on a list of geometries,
beside the F(2x2,3x3) kernel's error, then device time on the trunk shapes against F(2x2) and the direct kernel (dev tool, GPU box).
usage: python tools/wino4_check.py [--no-time] [--images N]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from hands_amd import _lib
from hands_amd._lib import ConvDesc, check, ptr
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_conv

DEV = "cuda"
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream


def run(kind, x, pc, act, out=None):
    B, H, W, _ = x.shape
    out = out if out is not None else torch.full((B, H, W, pc.Cout), float("nan"), device=DEV)
    d = ConvDesc(B, H, W, pc.Cin, H, W, pc.Cout, 3, 3, 1, 1, pc.Cin, pc.Cout, 0, pc.Kpad, int(act))
    if kind == "w4":
        assert L.hands_conv3x3_winograd4_supported(C.byref(d)) == 1
        check(L.hands_conv3x3_winograd4_f32(C.byref(d), ptr(x), ptr(pc.wino4), ptr(pc.bias), ptr(out), st()), "wino4")
    elif kind == "w2":
        check(L.hands_conv3x3_winograd_f32(C.byref(d), ptr(x), ptr(pc.wino), ptr(pc.bias), ptr(out), st()), "wino")
    else:
        check(L.hands_conv2d_nhwc_f32(C.byref(d), ptr(x), ptr(pc.w), ptr(pc.bias), None, ptr(out), st()), "direct")
    return out


CASES = [(3, 64, 56, 56, 64, 1), (5, 128, 28, 28, 128, 1), (7, 256, 14, 14, 256, 1), (9, 512, 7, 7, 512, 1), (2, 16, 1, 1, 32, 0),
         (3, 32, 2, 3, 32, 3), (2, 16, 5, 5, 64, 1), (1, 48, 9, 11, 96, 0), (2, 32, 20, 19, 32, 3), (4, 64, 13, 14, 32, 1),
         (1, 32, 64, 64, 64, 3), (33, 64, 8, 8, 64, 1), (2, 24, 28, 28, 32, 1), (3, 64, 55, 57, 64, 1), (40, 64, 56, 56, 64, 1)]
bad = 0
for case in (CASES if "--time-only" not in sys.argv else []):
    B, Cin, H, W, Cout, act = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pc = pack_conv(w, bias, 1, 1, DEV, winograd4=True)
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1)
    y = F.relu(y) if act == 1 else (F.leaky_relu(y, 0.01) if act == 3 else y)
    ref = y.permute(0, 2, 3, 1)
    xd = x.to(DEV)
    errs = {}
    for kind in ("w4", "w2", "direct"):
        if kind != "w4" and (pc.wino is None or Cin % 16):
            continue
        got = run(kind, xd, pc, act)
        torch.cuda.synchronize()
        got = got.cpu().double()
        errs[kind] = float("nan") if not torch.isfinite(got).all() else (got - ref).abs().max().item()
    import zlib
    crc = zlib.crc32(run("w4", xd, pc, act).cpu().numpy().tobytes())
    again = run("w4", xd, pc, act)
    first2 = run("w4", xd[:2].contiguous(), pc, act) if B > 2 else None
    torch.cuda.synchronize()
    det = torch.equal(again.cpu().double(), run("w4", xd, pc, act).cpu().double())
    inv = first2 is None or torch.equal(first2.cpu(), again[:2].cpu())
    scale = ref.abs().max().item()
    ok = errs["w4"] == errs["w4"] and errs["w4"] <= 4e-4 * scale and det and inv
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {case}: scale {scale:.2f}  err/scale  F(4x4) {errs['w4'] / scale:.2e}"
          + (f"  F(2x2) {errs['w2'] / scale:.2e}" if "w2" in errs else "") + (f"  direct {errs['direct'] / scale:.2e}" if "direct" in errs else "") +
          f"  deterministic {det}  batch-invariant {inv}  crc {crc:08x}", flush=True)
print("failures:", bad)

if "--no-time" not in sys.argv:
    n_img = int(sys.argv[sys.argv.index("--images") + 1]) if "--images" in sys.argv else 512
    for Cch, H in ((64, 56), (128, 28), (256, 14), (512, 7)):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n_img, H, H, Cch, generator=g).to(DEV)
        w = torch.randn(Cch, Cch, 3, 3, generator=g) / (Cch * 9) ** 0.5
        pc = pack_conv(w, torch.zeros(Cch), 1, 1, DEV, winograd4=True)
        out = torch.empty(n_img, H, H, Cch, device=DEV)
        flop = 2.0 * n_img * H * H * Cch * Cch * 9
        line = f"{n_img} x {Cch}ch {H}x{H}: "
        for kind in (("w4", "w2", "direct") if "--time-only" not in sys.argv else ("w4",)):
            for _ in range(3):
                run(kind, x, pc, 1, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(5):
                e0.record()
                for _ in range(4):
                    run(kind, x, pc, 1, out)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4)
            ms = sorted(ts)[len(ts) // 2]
            line += f" {kind} {ms * 1e3:7.1f} us ({flop / ms / 1e9:6.1f} TF/s alg)"
        print(line, flush=True)
sys.exit(1 if bad else 0)
