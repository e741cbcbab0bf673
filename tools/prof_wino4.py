#!/usr/bin/env python3
"""Phase profile of ONE hands_conv3x3_winograd4_f32 launch (dev tool, GPU box; needs build_ab/w4prof.so:
EXTRA_FLAGS=-DW4_PROF bash tools/build_variant.sh w4prof -).  s_memtime stamps (100 MHz constant clock on gfx950) of wave 0 (consumer) and
wave 12 (producer) of the first 512 workgroups: per stage, work done -> barrier passed.
usage: HANDS_HIP_LIB=build_ab/w4prof.so python tools/prof_wino4.py [Cch] [H] [images]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hands_amd import _lib
from hands_amd._lib import ConvDesc, check, ptr
from hands_amd.packing import pack_conv

Cch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 56
n_img = int(sys.argv[3]) if len(sys.argv) > 3 else 512
L = _lib.lib()
g = torch.Generator().manual_seed(1)
x = torch.randn(n_img, H, H, Cch, generator=g).to("cuda")
w = torch.randn(Cch, Cch, 3, 3, generator=g) / (Cch * 9) ** 0.5
pc = pack_conv(w, torch.zeros(Cch), 1, 1, "cuda", winograd4=True)
out = torch.empty(n_img, H, H, Cch, device="cuda")
d = ConvDesc(n_img, H, H, Cch, H, H, Cch, 3, 3, 1, 1, Cch, Cch, 0, pc.Kpad, 1)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    check(L.hands_conv3x3_winograd4_f32(C.byref(d), ptr(x), ptr(pc.wino4), ptr(pc.bias), ptr(out), st), "w4")
L.hands_debug_w4prof.argtypes = [C.c_void_p, C.c_int]
L.hands_debug_w4prof(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
check(L.hands_conv3x3_winograd4_f32(C.byref(d), ptr(x), ptr(pc.wino4), ptr(pc.bias), ptr(out), st), "w4")
e1.record()
torch.cuda.synchronize()
buf = np.zeros((512, 2, 64), np.uint64)
L.hands_debug_w4prof(buf.ctypes.data_as(C.c_void_p), 0)
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us")
t = buf.astype(np.int64)
ok = t[:, 0, 0] > 0
t = t[ok]
print("workgroups stamped:", len(t), " (units: s_memtime ticks; 100 MHz -> 10 ns)")
c, p = t[:, 0], t[:, 1]
steps = [g_ for g_ in range(19) if c[0, 4 + 3 * g_] > 0]
med = lambda v: float(np.median(v))
print(f"consumer wave 0: entry -> first stage {med(c[:, 1] - c[:, 0]):.0f}; loop {med(c[:, 4 + 3 * steps[-1]] - c[:, 1]):.0f}; "
      f"fold + hand-over {med(c[:, 60] - c[:, 4 + 3 * steps[-1]]):.0f}; output pass {med(c[:, 61] - c[:, 60]):.0f}; life {med(c[:, 61] - c[:, 0]):.0f}")
print(f"producer wave 12: entry -> first stage {med(p[:, 1] - p[:, 0]):.0f}; output pass {med(p[:, 61] - p[:, 60]):.0f}")
for g_ in steps:
    prev = c[:, 4 + 3 * (g_ - 1)] if g_ else c[:, 1]
    pprev = p[:, 4 + 3 * (g_ - 1)] if g_ else p[:, 1]
    print(f"stage {g_:2d}: consumer mfma-issue {med(c[:, 2 + 3 * g_] - prev):6.0f}  wait+barrier {med(c[:, 4 + 3 * g_] - c[:, 2 + 3 * g_]):6.0f} | "
          f"producer fill+transform {med(p[:, 3 + 3 * g_] - p[:, 2 + 3 * g_]):6.0f}  wait+barrier {med(p[:, 4 + 3 * g_] - p[:, 3 + 3 * g_]):6.0f}")
