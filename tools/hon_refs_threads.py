#!/usr/bin/env python3
"""fp32 oracle vertices of HandOccNet at ANOTHER ATen thread count for the seeds of build_ab/hon_refs_<first>_<n>.npz (dev
container, CPU): ATen's blocked sums move the reference's own vertices by 2-5e-7 m with the thread count, so
the many-seed A/B compares the HIP path with the reference run at 1, 8 and 16 threads.
Output: build_ab/hon_refs_<first>_<n>_t<threads>.npz (first, v32; not committed).
usage: python tools/hon_refs_threads.py <threads> [first] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import handoccnet_oracle as HO

threads = int(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
torch.set_num_threads(threads)
m = hands_amd.apply_recipe(hands_amd.HandOccNet())
sd = {k: v.clone() for k, v in m.state_dict().items()}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
out = os.path.join(ROOT, "build_ab", f"hon_refs_{first}_{n}_t{threads}.npz")
v32s = []
for i, seed in enumerate(range(first, first + n)):
    ci, cm = synthetic_inputs(2, seed)
    r = HO.handoccnet_forward(sd, ar, al, ci, cm)
    v32s.append(torch.stack([r[f"mano.vertices.{h}"] for h in "rl"]).numpy())
    if (i + 1) % 100 == 0 or i + 1 == n:
        np.savez(out, first=first, threads=threads, v32=np.stack(v32s))
        print(threads, i + 1, flush=True)
