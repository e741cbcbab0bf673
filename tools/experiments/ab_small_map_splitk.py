#!/usr/bin/env python3
"""HandOccNet: the call-site split-K rules for small maps (handoccnet.py _conv_fns) on / off, alternating on ONE model instance
(the rules are read at every forward): hands/s pipelined at 32 and 256 samples, and the latency of one synchronous forward at 2.
usage (GPU box): python tools/ab_small_map_splitk.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SETTINGS = [("both rules", True), ("none", False), ("maps <= 8x8 only", "deep"), ("16x16 rule only", "16x16")]
m = hands_amd.apply_recipe(hands_amd.HandOccNet()).to("cuda").eval()
data = {bz: synthetic_inputs(bz, 0, device="cuda") for bz in (2, 32, 256)}
times = {(n, bz): [] for n, _ in SETTINGS for bz in data}
for r in range(rounds + 1):
    for name, val in SETTINGS:
        m.small_map_splitk = val
        for bz, (gi, gm) in data.items():
            m.async_forward = bz != 2
            reps = {2: 20, 32: 24, 256: 5}[bz]
            for _ in range(3):
                out = m(gi, gm)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                out = m(gi, gm)
                if bz == 2:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            if r:
                times[(name, bz)].append((time.perf_counter() - t) / reps)
            del out
for name, _ in SETTINGS:
    md = {bz: float(np.median(times[(name, bz)])) for bz in data}
    print(f"{name:20s} bz 32: {64 / md[32]:7.1f} hands/s   bz 256: {512 / md[256]:7.1f} hands/s   bz 2 synchronous: {1e3 * md[2]:6.3f} ms / forward")
