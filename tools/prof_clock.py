#!/usr/bin/env python3
"""Shader clock over the life of ONE conv_igemm launch (dev tool): every workgroup stamps s_memrealtime (100 MHz) and
s_memtime (shader clock) at entry and exit; needs python tools/instrument.py clock.
usage: HANDS_HIP_LIB=build_ab/prof_clock.so python tools/prof_clock.py B,Cin,H,Cout,k,stride,pad,res [...]"""
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
eng.stream_k = False
dev = "cuda"
stream = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("PROF_REPS", "10"))
for spec in sys.argv[1:]:
    B, Cin, H, Cout, k, st, pad, use_res = [int(v) for v in spec.split(",")]
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), st, pad, dev)
    x = torch.randn(B, H, H, Cin, device=dev)
    Ho = (H + 2 * pad - k) // st + 1
    out = torch.empty(B, Ho, Ho, Cout, device=dev)
    torch.cuda.synchronize()
    for _ in range(reps):       # back to back, like the benchmark: the LAST launch's stamps survive
        eng.conv(L, pc, x, B, H, H, out, True, stream)
    torch.cuda.synchronize()
    prof = np.zeros(32768 * 4, dtype=np.uint64)
    raw.hands_debug_prof(C.c_void_p(prof.ctypes.data), None)
    bm, bn = (256, 64) if Cout <= 64 else (128, 128)
    nt = min(-(-B * Ho * Ho // bm) * -(-Cout // bn), 32768)
    p = prof.reshape(-1, 4)[:nt].astype(np.int64)
    t0 = p[:, 0].min()
    start = (p[:, 0] - t0) * 0.01
    end = (p[:, 2] - t0) * 0.01
    dur = end - start
    mhz = (p[:, 3] - p[:, 1]) / np.maximum(dur, 1e-3)
    print(f"== {spec}: tiles {nt}, span {end.max():.1f} us, tile duration {dur.mean():.1f} us, s_memtime rate {np.median(mhz):.0f} MHz")
    order = np.argsort(start)
    nb = 8
    for i in range(nb):
        sel = order[i * nt // nb:(i + 1) * nt // nb]
        print(f"   tiles starting {start[sel].min():8.1f}..{start[sel].max():8.1f} us: duration {dur[sel].mean():7.1f} us   s_memtime rate {np.median(mhz[sel]):.0f} MHz")
