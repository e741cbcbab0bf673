#!/usr/bin/env python3
"""Shader clock seen by the conv_igemm launches of whole forwards (dev tool; library built from a copy of conv_igemm.hip
in which the middle workgroup of every plain launch stamps s_memrealtime / s_memtime at entry and exit into a ring).
usage: HANDS_HIP_LIB=build_ab/prof_ring.so python tools/prof_ring.py [serial|overlap] [bz]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import hands_amd

mode = sys.argv[1] if len(sys.argv) > 1 else "serial"
bz = int(sys.argv[2]) if len(sys.argv) > 2 else 256
raw = C.CDLL(os.environ["HANDS_HIP_LIB"])
model = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
inputs, meta = hands_amd.synthetic_inputs(bz, seed=0, device=torch.device("cuda"))
model.engine.overlap = mode != "serial"
model.engine.stream_k = False          # the persistent launches are not instrumented
with torch.no_grad():
    for _ in range(3):
        model(inputs, meta)["mano.v3d.cam.r"]
    torch.cuda.synchronize()
    ring = np.zeros(8192 * 4, dtype=np.uint64)
    n = C.c_uint(0)
    raw.hands_debug_ring(C.c_void_p(ring.ctypes.data), C.byref(n), 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        model(inputs, meta)["mano.v3d.cam.r"]
    e1.record()
    torch.cuda.synchronize()
    raw.hands_debug_ring(C.c_void_p(ring.ctypes.data), C.byref(n), 0)
r = ring.reshape(-1, 4)[:min(n.value, 8192)].astype(np.int64)
dur = (r[:, 2] - r[:, 0]) * 0.01
mhz = (r[:, 3] - r[:, 1]) / np.maximum(dur, 1e-3)
ok = dur > 20
print(f"{mode} bz={bz}: {e0.elapsed_time(e1) / 3:.2f} ms per forward, {n.value} stamped launches; workgroup-lifetime-weighted clock "
      f"{np.sum(mhz[ok] * dur[ok]) / np.sum(dur[ok]):.0f} MHz; percentiles 10/50/90: "
      + " / ".join(f"{np.percentile(mhz[ok], q):.0f}" for q in (10, 50, 90)))
