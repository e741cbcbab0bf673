#!/usr/bin/env python3
"""Device time of hands_attention_f32 at hamer_light's size (dev tool).  usage: python tools/bench_attn.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
from hands_amd import _lib
from hands_amd._lib import check, ptr
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T, heads, D = 192, 16, 80
qkv = torch.randn(B, T, 3 * heads * D, device="cuda")
out = torch.empty(B, T, heads * D, device="cuda")
st = torch.cuda.current_stream().cuda_stream
run = lambda: check(L.hands_attention_f32(ptr(qkv), ptr(out), B, T, heads, D, float(D ** -0.5), st), "attention")
for _ in range(5):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
flop = B * heads * (2 * T * T * D * 2)
print(f"B={B}: {us:.1f} us per launch, {flop / us / 1e6:.1f} TFLOP/s algorithmic (D = 80, unpadded)")
