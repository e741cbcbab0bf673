#!/usr/bin/env python3
"""Phase profile of ONE hands_attention_f32 launch at hamer_light's size (128 crops x 16 heads, T = 192, D = 80): where a
wave spends its life (dev tool; needs `python tools/instrument.py attn`).
usage: HANDS_HIP_LIB=build_ab/prof_attn.so python tools/prof_attn.py [B]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
from hands_amd import _lib
from hands_amd._lib import check, ptr

L = _lib.lib()
raw = C.CDLL(os.environ["HANDS_HIP_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T, heads, D = 192, 16, 80
dev = "cuda"
qkv = torch.randn(B, T, 3 * heads * D, device=dev)
out = torch.empty(B, T, heads * D, device=dev)
st = torch.cuda.current_stream().cuda_stream
run = lambda: check(L.hands_attention_f32(ptr(qkv), ptr(out), B, T, heads, D, float(D ** -0.5), st), "attention")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
prof = np.zeros(32768 * 8, dtype=np.uint64)
raw.hands_debug_aprof(C.c_void_p(prof.ctypes.data))
p = prof.reshape(-1, 8).astype(np.int64)
p = p[p[:, 6] > 0]
hw = p[:, 7].copy()                              # HW_ID (low 32 bits) | XCC_ID << 32, read at the end of the wave
p[:, 7] = p[:, 6]
tick_ns = 10.0                                   # s_memrealtime: 100 MHz
d = lambda a, b: (p[:, b] - p[:, a]).astype(np.float64) * tick_ns / 1e3      # microseconds
life = d(0, 7)
print(f"B={B}: {e0.elapsed_time(e1) * 1e3:.1f} us per launch, {len(p)} waves stamped (of {B * heads * 12}), "
      f"wave life {life.mean():.1f} us (p10 {np.percentile(life, 10):.1f}, p90 {np.percentile(life, 90):.1f}); "
      f"launch span {(p[:, 7].max() - p[:, 0].min()) * tick_ns / 1e3:.1f} us")
names = ("K fill + Q load + barrier", "Q.K^T (240 MFMAs 16x16x4)", "barrier (K free) + V^T registers -> LDS", "softmax (48 exps)",
         "barrier: V^T complete", "P.V (240 MFMAs) + stores")
for i, nm in enumerate(names):
    v = d(i, i + 1)
    print(f"   {nm:34s} {v.mean():7.2f} us  ({100 * v.mean() / life.mean():4.1f} %)   p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f}")
# which CU ran which wave: HW_ID bits [11:8] cu, [12] sh, [15:13] se (gfx9 layout), XCC_ID bits [3:0]
cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf)
simd = (hw >> 4) & 3
ncu = len(np.unique(cu))
t0, t1 = p[:, 0].min(), p[:, 6].max()
probe = np.linspace(t0 + 0.1 * (t1 - t0), t0 + 0.8 * (t1 - t0), 40)
alive = [np.sum((p[:, 0] <= t) & (p[:, 6] > t)) for t in probe]
per_cu = []
for t in probe[::8]:
    m = (p[:, 0] <= t) & (p[:, 6] > t)
    _, cnt = np.unique(cu[m], return_counts=True)
    per_cu.append(np.bincount(cnt, minlength=26)[:26])
print(f"   {ncu} distinct CUs seen; waves alive in the steady part: mean {np.mean(alive):.0f} = {np.mean(alive) / ncu:.1f} per CU "
      f"({np.mean(alive) / ncu / 12:.2f} workgroups of 12 waves)")
print("   CUs holding k waves (k = 0..25), sampled at 5 instants:")
for row in per_cu:
    print("     ", " ".join(f"{v:3d}" for v in row))
mf = 480 * 32 / 2.4e3
print(f"   a wave's own MFMA issue time: 480 x 32 cycles = {mf:.1f} us at 2.4 GHz; 3 waves per SIMD -> {3 * mf:.1f} us of pipe time per "
      f"wave life of {life.mean():.1f} us = {100 * 3 * mf / life.mean():.0f} % busy")
