#!/usr/bin/env python3
"""hands_light: max vertex error of the HIP path against the oracle (fp32 CPU port of the reference) over several input
seeds, with the Winograd route (default) and the direct 3x3 kernel (dev tool, GPU box).  usage: python tools/hl_parity_sweep.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import hands_oracle as O

torch.set_num_threads(16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = hands_amd.apply_recipe(hands_amd.HandsLight())
sd = {k: v.clone() for k, v in model.state_dict().items()}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
model = model.to("cuda").eval()
worst = {True: 0.0, False: 0.0}
for seed in range(n):
    ci, cm = synthetic_inputs(2, seed)
    ref = O.hands_light_forward(sd, ar, al, ci, cm)
    row = []
    for wino in (True, False):
        model.engine.winograd = wino
        out = model({k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()})
        e = max((out[f"mano.vertices.{h}"].cpu() - ref[f"mano.vertices.{h}"]).abs().max().item() for h in "rl")
        mp = max(O.mpjpe_ra_mm(out[f"mano.joints3d.{h}"].cpu(), ref[f"mano.joints3d.{h}"]) for h in "rl")
        worst[wino] = max(worst[wino], e)
        row.append(f"{'winograd' if wino else 'direct'} {e:.2e} m / {mp:.1e} mm")
    print(f"seed {seed}: " + " | ".join(row))
print(f"worst over {n} seeds: winograd {worst[True]:.2e} m, direct {worst[False]:.2e} m (bar 1e-6 m)")
