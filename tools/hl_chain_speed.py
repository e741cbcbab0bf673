#!/usr/bin/env python3
"""hands_light: throughput cost of blocked summation settings, measured on ONE model instance (instances differ by +-2 % through
buffer placement): the engine reads the chain switches at every launch, so the settings alternate between timing rounds.
usage (GPU box): python tools/hl_chain_speed.py [bz] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs

bz = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
# (name, chain_limit, chain_min_k, chain_max_pix)
SETTINGS = [("single chains", 0, 0, 0), ("c64 maps<=49", 64, 0, 49), ("c64 K>=1024 maps<=49", 64, 1024, 49),
            ("c64 K>=1024 maps<=196", 64, 1024, 196), ("c64 K>=1024", 64, 1024, 0), ("c64 K>=512", 64, 512, 0), ("c64 all", 64, 0, 0),
            ("c128 K>=1024", 128, 1024, 0)]
m = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
gi, gm = synthetic_inputs(bz, 0, device="cuda")
times = {s[0]: [] for s in SETTINGS}
for r in range(rounds + 1):
    for name, lim, mk, mp in SETTINGS:
        e = m.engine
        e.chain_limit, e.chain_min_k, e.chain_max_pix, e.chain_in_kernel = lim, mk, mp, bool(lim)
        out = m(gi, gm)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(6):
            out = m(gi, gm)
        torch.cuda.synchronize()
        if r:                       # round 0 warms every setting up
            times[name].append((time.perf_counter() - t) / 6)
base = float(np.median(times[SETTINGS[0][0]]))
for name, ts in times.items():
    md = float(np.median(ts))
    print(f"{name:24s} median {2 * bz / md:8.1f} hands/s ({100 * (base / md - 1):+5.2f} %)  best {2 * bz / min(ts):8.1f}  worst {2 * bz / max(ts):8.1f}")
