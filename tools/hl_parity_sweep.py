#!/usr/bin/env python3
"""hands_light: max vertex error of the HIP path against the oracle (fp32 CPU port of the reference) over several input
seeds, per 3x3 route: direct kernel, Winograd F(2x2,3x3), F(4x4,3x3) in some / all ResNet stages (dev tool, GPU box).
usage: python tools/hl_parity_sweep.py [n seeds] [first seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import hands_oracle as O

torch.set_num_threads(min(8, os.cpu_count() or 1))          # the count tests/conftest.py pins
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
# arms: the direct 3x3 kernel, F(2x2,3x3) everywhere, F(4x4,3x3) in the given ResNet stages (F(2x2) elsewhere)
arms = {"direct": None, "f2x2": (), "f4x4:4": (4,), "f4x4:34": (3, 4), "f4x4:1234": (1, 2, 3, 4)}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
models = {}
for name, stages in arms.items():
    m = hands_amd.apply_recipe(hands_amd.HandsLight())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.winograd4_stages = stages or ()
    m = m.to("cuda").eval()
    m.engine.winograd = stages is not None
    m.async_tail = False
    models[name] = m
errs = {k: [] for k in arms}
for seed in range(first, first + n):
    ci, cm = synthetic_inputs(2, seed)
    ref = O.hands_light_forward(sd, ar, al, ci, cm)
    gi, gm = {k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()}
    for name, m in models.items():
        out = m(gi, gm)
        errs[name].append(max((out[f"mano.vertices.{h}"].cpu() - ref[f"mano.vertices.{h}"]).abs().max().item() for h in "rl"))
import numpy as np
for name, e in errs.items():
    e = np.array(e)
    print(f"{name:10s}: {len(e)} seeds: median {np.median(e):.2e}  p90 {np.percentile(e, 90):.2e}  max {e.max():.2e} m (bar 1e-6 m), above 7e-7: {(e > 7e-7).sum()}")
