#!/usr/bin/env python3
"""conv_wino4 workgroup order (round 6): band = 1 (round 5: channel block outermost) against the banded order, per trunk layer.
Device time alternating in one process (HANDS_W4_BAND is read per launch), and -- under `rocprofv3 --pmc FETCH_SIZE` with
--sequence -- ONE launch per (layer, band) in a fixed order so the counter rows can be matched.
Needs a library built with -DHANDS_W4_BAND_ENV (the shipped one reads no environment):
    EXTRA_FLAGS=-DHANDS_W4_BAND_ENV bash tools/build_variant.sh w4env -     (dev container)
    HANDS_HIP_LIB=build_ab/w4env.so python tools/experiments/w4_band.py [--sequence] [--images N]     (GPU box)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from hands_amd import _lib
from hands_amd._lib import ConvDesc, check, ptr
from hands_amd.packing import pack_conv

L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
n_img = int(sys.argv[sys.argv.index("--images") + 1]) if "--images" in sys.argv else 512
LAYERS = ((64, 56), (128, 28), (256, 14), (512, 7))
BANDS = ("1", "0")      # "0": the library's default band


def launch(x, pc, out, H):
    d = ConvDesc(x.shape[0], H, H, pc.Cin, H, H, pc.Cout, 3, 3, 1, 1, pc.Cin, pc.Cout, 0, pc.Kpad, 1)
    check(L.hands_conv3x3_winograd4_f32(C.byref(d), ptr(x), ptr(pc.wino4), ptr(pc.bias), ptr(out), st()), "wino4")


for Cch, H in LAYERS:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n_img, H, H, Cch, generator=g).to("cuda")
    w = torch.randn(Cch, Cch, 3, 3, generator=g) / (Cch * 9) ** 0.5
    pc = pack_conv(w, torch.zeros(Cch), 1, 1, "cuda", winograd4=True)
    out = torch.empty(n_img, H, H, Cch, device="cuda")
    if "--sequence" in sys.argv:
        for band in BANDS:
            os.environ["HANDS_W4_BAND"] = band
            launch(x, pc, out, H)
            torch.cuda.synchronize()
        continue
    ref = None
    res = {b: [] for b in BANDS}
    for rep in range(6):
        for band in BANDS:
            os.environ["HANDS_W4_BAND"] = band
            for _ in range(2):
                launch(x, pc, out, H)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                launch(x, pc, out, H)
            e1.record()
            torch.cuda.synchronize()
            res[band].append(e0.elapsed_time(e1) / 8 * 1e3)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), "the order of the workgroups changed an output bit"
    flop = 2.0 * n_img * H * H * Cch * Cch * 9
    print(f"{n_img} x {Cch}ch {H}x{H}: " + "  ".join(
        f"band {'old(1)' if b == '1' else 'new'} {sorted(v)[len(v) // 2]:7.1f} us ({flop / sorted(v)[len(v) // 2] / 1e6:6.1f} TF/s alg)"
        for b, v in res.items()), flush=True)
