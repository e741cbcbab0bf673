#!/usr/bin/env python3
"""Per-workgroup phases of ONE conv_wino launch in shader cycles (dev tool; needs python tools/instrument.py wino).
usage: HANDS_HIP_LIB=build_ab/prof_wino.so python tools/prof_wino.py B,Cin,H,Cout [...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
from hands_amd import _lib
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_conv

L = _lib.lib()
raw = C.CDLL(os.environ["HANDS_HIP_LIB"])
eng = ConvEngine()
dev = "cuda"
stream = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[1:]:
    vals = [int(v) for v in spec.split(",")]
    B, Cin, H, Cout = vals[:4]
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), 1, 1, dev)
    x = torch.randn(B, H, H, Cin, device=dev)
    out = torch.empty(B, H, H, Cout, device=dev)
    run = lambda: eng.conv(L, pc, x, B, H, H, out, True, stream)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    raw.hands_debug_wprof_clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    prof = np.zeros(32768 * 8, dtype=np.uint64)
    raw.hands_debug_wprof(C.c_void_p(prof.ctypes.data))
    raw.hands_debug_wprof_clear()
    p = prof.reshape(-1, 8).astype(np.int64)
    p = p[p[:, 5] > 0]
    d = lambda a, b: (p[:, b] - p[:, a]).astype(np.float64)
    nch = Cin // 16
    life, setup, fill, epi = d(0, 5), d(0, 1), d(1, 2), p[:, 4].astype(np.float64)
    loop = d(2, 5) - epi
    nbw = int(os.environ.get("HANDS_WINO_NBW", "0"))
    print(f"B{B} {Cin}->{Cout} H{H}: {e0.elapsed_time(e1) * 1e3:.1f} us, {len(p)} workgroups, life {life.mean():.0f} cycles "
          f"(p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
    for name, v in (("setup + first DMA issue", setup), ("first barrier (DMA latency)", fill), ("channel stages", loop), ("epilogues", epi)):
        print(f"   {name:28s} {v.mean():9.0f}  ({100 * v.mean() / life.mean():4.1f} %)")
    print(f"   stages take {loop.mean():.0f} cycles per workgroup; a stage is 32 MFMAs = 2048 matrix-pipe cycles per wave, x3 waves per SIMD = 6144")
