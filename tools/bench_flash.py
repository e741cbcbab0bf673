#!/usr/bin/env python3
"""Device time of hands_flash_attention_f32 at handoccnet_light's size (dev tool).  usage: python tools/bench_flash.py [crops]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
from hands_amd import _lib
from hands_amd._lib import check, ptr
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N, heads, D = 1024, 4, 64
C = heads * D
q, k, v, res = (torch.randn(B, N, C, device="cuda") for _ in range(4))
out = torch.empty(B, N, C, device="cuda")
st = torch.cuda.current_stream().cuda_stream
run = lambda: check(L.hands_flash_attention_f32(ptr(q), ptr(k), ptr(v), None, None, ptr(res), ptr(out), B, N, heads, D, float(D ** -0.5), st), "flash")
for _ in range(5):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 30
print(f"crops={B}: {us:.1f} us per launch, {B * heads * 4.0 * N * N * D / us / 1e6:.1f} TFLOP/s")
