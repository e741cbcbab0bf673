#!/usr/bin/env python3
"""hands_light: how much of the +-2 % spread between model instances is buffer placement?  (a) ONE instance whose workspaces
(`_ws`: the per-stream activation buffers) are dropped and re-allocated while the old blocks are held, so the allocator must hand
out other addresses; (b) fresh instances (other weight addresses as well).  Timing rounds alternate over all variants.
usage (GPU box): python tools/placement_variance.py [bz] [variants] [rounds]"""
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
nvar = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gi, gm = synthetic_inputs(bz, 0, device="cuda")


def timed(m, reps=6):
    out = m(gi, gm)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = m(gi, gm)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


m = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
for _ in range(3):
    m(gi, gm)
torch.cuda.synchronize()
# (a) the same instance (same weights) with several workspace placements: keep every set alive, swap the dict in
sets, held = [], []
for v in range(nvar):
    if v:
        held.append(torch.empty((37 + 61 * v) << 20, dtype=torch.uint8, device="cuda"))    # shift what the allocator hands out next
        m._ws = {}
        for _ in range(2):
            m(gi, gm)
        torch.cuda.synchronize()
    sets.append(m._ws)
# (a') new buffers but the FIRST set's streams and events: is it the memory or the streams (HIP maps streams onto 4 hardware queues)?
held.append(torch.empty(99 << 20, dtype=torch.uint8, device="cuda"))
m._ws = {k: v for k, v in sets[0].items() if not torch.is_tensor(v)}
for _ in range(2):
    m(gi, gm)
torch.cuda.synchronize()
sets.append(m._ws)
nvar_a = len(sets)
addr = lambda ws: sorted((k, v.data_ptr()) for k, v in ws.items() if torch.is_tensor(v) and v.numel() > (1 << 24))
ta = {v: [] for v in range(nvar_a)}
for r in range(rounds):
    for v in range(nvar_a):
        m._ws = sets[v]
        ta[v].append(timed(m))
base = float(np.median(ta[0]))
print("(a) one instance, workspace placements:")
for v in range(nvar_a):
    md = float(np.median(ta[v]))
    big = addr(sets[v])
    print(f"  placement {v}{' (new buffers, streams of placement 0)' if v == nvar_a - 1 else ''}: {2 * bz / md:8.1f} hands/s ({100 * (base / md - 1):+5.2f} %)  best {2 * bz / min(ta[v]):8.1f}; "
          f"{len(big)} buffers > 64 MB, first at {big[0][1]:#x} ({big[0][1] % (1 << 21):#x} mod 2 MB)")
m._ws = sets[0]
# (b) fresh instances
inst = [m] + [hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval() for _ in range(nvar - 1)]
for x in inst[1:]:
    for _ in range(3):
        x(gi, gm)
torch.cuda.synchronize()
tb = {i: [] for i in range(nvar)}
for r in range(rounds):
    for i, x in enumerate(inst):
        tb[i].append(timed(x))
base = float(np.median(tb[0]))
print("(b) fresh instances:")
for i in range(nvar):
    md = float(np.median(tb[i]))
    print(f"  instance {i}: {2 * bz / md:8.1f} hands/s ({100 * (base / md - 1):+5.2f} %)  best {2 * bz / min(tb[i]):8.1f}")
