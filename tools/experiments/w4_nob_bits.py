#!/usr/bin/env python3
"""Are the 64-channel-per-workgroup conv_wino4 launches (round 6) bit-identical to the 32-channel form (-DHANDS_W4_NOB1)?  Runs the
four trunk shapes and a hands_light forward with the library named by $HANDS_HIP_LIB (or the in-tree one) and prints CRCs; run it
twice (with and without the variant) and compare the lines.  usage: [HANDS_HIP_LIB=build_ab/w4nob1.so] python tools/experiments/w4_nob_bits.py"""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch

import hands_amd
from hands_amd import _lib
from hands_amd._lib import ConvDesc, check, ptr
from hands_amd.packing import pack_conv

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for Cch, H, B in ((64, 56, 6), (128, 28, 9), (256, 14, 17), (512, 7, 33)):
    g = torch.Generator().manual_seed(Cch)
    x = torch.randn(B, H, H, Cch, generator=g).cuda()
    pc = pack_conv(torch.randn(Cch, Cch, 3, 3, generator=g) / (Cch * 9) ** 0.5, torch.randn(Cch, generator=g), 1, 1, "cuda", winograd4=True)
    out = torch.empty(B, H, H, Cch, device="cuda")
    d = ConvDesc(B, H, H, Cch, H, H, Cch, 3, 3, 1, 1, Cch, Cch, 0, pc.Kpad, 1)
    check(L.hands_conv3x3_winograd4_f32(C.byref(d), ptr(x), ptr(pc.wino4), ptr(pc.bias), ptr(out), st), "w4")
    torch.cuda.synchronize()
    print(f"layer {Cch}ch {H}x{H}: crc {zlib.crc32(out.cpu().numpy().tobytes()):08x}")
m = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
inputs, meta = hands_amd.synthetic_inputs(3, seed=5, device="cuda")
o = m(inputs, meta)
torch.cuda.synchronize()
print("hands_light forward: crc", " ".join(f"{zlib.crc32(o[k].cpu().numpy().tobytes()):08x}" for k in ("mano.vertices.r", "mano.vertices.l", "grasp.r")))
