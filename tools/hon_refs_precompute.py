#!/usr/bin/env python3
"""Precompute the fp32 (8 ATen threads) and fp64 oracle vertices of HandOccNet for many input seeds on CPU (dev container),
so that tools/hon_parity_ab.py --refs spends no GPU-box time on them.  Output: build_ab/hon_refs_<first>_<n>.npz (not committed).
usage: python tools/hon_refs_precompute.py [first seed] [n seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import hon_parity_ab as H

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
H._worker_init()
v32s, v64s = [], []
out = os.path.join(H.ROOT, "build_ab", f"hon_refs_{first}_{n}.npz")
for i, seed in enumerate(range(first, first + n)):
    _, v32, v64 = H._worker(seed)
    v32s.append(v32); v64s.append(v64)
    if (i + 1) % 50 == 0 or i + 1 == n:
        np.savez(out, first=first, v32=np.stack(v32s), v64=np.stack(v64s))
        print(i + 1, flush=True)
