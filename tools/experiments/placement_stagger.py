#!/usr/bin/env python3
"""hands_light: does the RELATIVE alignment of the activation workspaces matter?  One instance; every variant re-allocates the
workspaces (`_ws`) with buffer i shifted by i * S bytes inside a padded allocation (S = 0: plain late placement).  Placement 0 (the
process's first) is kept for reference; timing rounds alternate.  Companion of placement_variance.py (a later placement is ~2.7 %
slower than the first; a streaming kernel does not care: placement_stream.py).
usage (GPU box): python tools/placement_stagger.py [bz] [rounds]"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs

bz = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
STAGGERS = [0, 256, 4096 + 256, 65536 + 4096 + 256, (1 << 20) + 65536 + 4096 + 256, 0]
gi, gm = synthetic_inputs(bz, 0, device="cuda")
m = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
for _ in range(3):
    m(gi, gm)
torch.cuda.synchronize()
sets = [("placement 0 (first)", m._ws)]
state = {"S": 0, "i": 0}
plain_buf = type(m)._buf


def staggered_buf(self, name, numel, dev):
    t = self._ws.get(name)
    if t is None or t.numel() < numel or t.device != dev:
        S = state["S"]
        off = (state["i"] * S // 4) % (1 << 23)            # floats; at most 32 MB of shift
        state["i"] += 1
        base = torch.empty(numel + (1 << 23) + 64, dtype=torch.float32, device=dev)
        t = base[off:off + numel]
        self._ws[name] = t
        self._ws["_keep_" + name] = base
    return t


m._buf = types.MethodType(staggered_buf, m)
held = []
for k, S in enumerate(STAGGERS):
    held.append(torch.empty((37 + 61 * k) << 20, dtype=torch.uint8, device="cuda"))
    state["S"], state["i"] = S, 0
    m._ws = {}
    for _ in range(2):
        m(gi, gm)
    torch.cuda.synchronize()
    sets.append((f"late, stagger {S} B", m._ws))
times = {n: [] for n in range(len(sets))}
for r in range(rounds):
    for n, (_, ws) in enumerate(sets):
        m._ws = ws
        out = m(gi, gm)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(6):
            out = m(gi, gm)
        torch.cuda.synchronize()
        times[n].append((time.perf_counter() - t) / 6)
base = float(np.median(times[0]))
for n, (name, _) in enumerate(sets):
    md = float(np.median(times[n]))
    print(f"{name:32s} {2 * bz / md:8.1f} hands/s ({100 * (base / md - 1):+5.2f} %)  best {2 * bz / min(times[n]):8.1f}")
