#!/usr/bin/env python3
"""hands_light: which torch pool streams the forward's four side streams are matters (placement_variance.py: a later set of streams
costs ~3 %, the buffers do not).  Same instance, same buffers; the four named streams (side0, side1 = trunk jobs, tail, side_head)
are taken from a list of pool streams created up front, in several assignments; timing rounds alternate.
usage (GPU box): python tools/stream_mapping.py [bz] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs

bz = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gi, gm = synthetic_inputs(bz, 0, device="cuda")
m = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
for _ in range(3):
    m(gi, gm)
torch.cuda.synchronize()
first = m._ws
names = [k for k, v in first.items() if isinstance(v, torch.cuda.Stream)]
print("streams of the first forward, in creation order:", names, [hex(first[k].cuda_stream) for k in names])
pool = [torch.cuda.Stream() for _ in range(12)]
ASSIGN = {"first set": None, "pool 0 1 2 3": (0, 1, 2, 3), "pool 4 5 6 7": (4, 5, 6, 7), "pool 1 2 3 4": (1, 2, 3, 4),
          "pool 0 4 1 5": (0, 4, 1, 5), "pool 0 1 4 5": (0, 1, 4, 5), "pool 0 1 2 2": (0, 1, 2, 2), "pool 0 2 4 6": (0, 2, 4, 6),
          "pool 8 9 10 11": (8, 9, 10, 11)}
hi = [torch.cuda.Stream(priority=-1) for _ in range(4)]      # high-priority pool
ASSIGN.update({"trunks high prio": ("h0", "h1", 2, 3), "trunks high, tail+head on pool 4 5": ("h0", "h1", 4, 5),
               "tail+head high prio": (0, 1, "h2", "h3"), "all high prio": ("h0", "h1", "h2", "h3"),
               "tail = side0's stream, head = side1's": ("f0", "f1", "f0", "f1")})
pick = lambda i: (hi[int(i[1])] if i[0] == "h" else first[("side0", "side1")[int(i[1])]]) if isinstance(i, str) else pool[i]
sets = {}
for name, idx in ASSIGN.items():
    ws = dict(first)
    if idx is not None:
        for k, i in zip(("side0", "side1", "tail", "side_head"), idx):
            ws[k] = pick(i)
    sets[name] = ws
times = {n: [] for n in sets}
for r in range(rounds + 1):
    for n, ws in sets.items():
        m._ws = ws
        out = m(gi, gm)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(6):
            out = m(gi, gm)
        torch.cuda.synchronize()
        if r:
            times[n].append((time.perf_counter() - t) / 6)
base = float(np.median(times["first set"]))
for n in sets:
    md = float(np.median(times[n]))
    print(f"{n:40s} {2 * bz / md:8.1f} hands/s ({100 * (base / md - 1):+5.2f} %)  best {2 * bz / min(times[n]):8.1f}")
