#!/usr/bin/env python3
"""HandOccNet: forwards in flight (pipeline_depth) x crop jobs per forward (chunks), alternating on ONE model instance.
usage (GPU box): python tools/ab_pipeline_depth.py [rounds]"""
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
m = hands_amd.apply_recipe(hands_amd.HandOccNet()).to("cuda").eval()
data = {bz: synthetic_inputs(bz, 0, device="cuda") for bz in (32, 256)}
SETTINGS = [(d, c) for d in (2, 3, 4, 6) for c in (1, 2)]
times = {(s, bz): [] for s in SETTINGS for bz in data}
for r in range(rounds + 1):
    for s in SETTINGS:
        m.pipeline_depth, m.chunks = s
        for bz, (gi, gm) in data.items():
            reps = 24 if bz == 32 else 6
            for _ in range(s[0] + 1):
                out = m(gi, gm)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                out = m(gi, gm)
            torch.cuda.synchronize()
            if r:
                times[(s, bz)].append((time.perf_counter() - t) / reps)
            del out
for s in SETTINGS:
    md = {bz: float(np.median(times[(s, bz)])) for bz in data}
    print(f"depth {s[0]} chunks {s[1]}:  bz 32: {64 / md[32]:7.1f} hands/s   bz 256: {512 / md[256]:7.1f} hands/s")
