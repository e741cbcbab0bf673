#!/usr/bin/env python3
"""Distribution of the max vertex error of the SHIPPED default routes against the oracle over many input seeds (dev tool, GPU box).
usage: python tools/parity_many_seeds.py [n_seeds, default 64] [first seed, default 100]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch
import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import handoccnet_oracle as HO
from oracle import hands_oracle as O

torch.set_num_threads(16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
for name, ctor, fwd in (("handoccnet_light", hands_amd.HandOccNet, HO.handoccnet_forward), ("hands_light", hands_amd.HandsLight, O.hands_light_forward)):
    m = hands_amd.apply_recipe(ctor())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to("cuda").eval()
    errs = []
    for seed in range(s0, s0 + n):
        ci, cm = synthetic_inputs(2, seed)
        ref = fwd(sd, ar, al, ci, cm)
        out = m({k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()})
        torch.cuda.synchronize()
        errs.append(max((out[f"mano.vertices.{h}"].cpu() - ref[f"mano.vertices.{h}"]).abs().max().item() for h in "rl"))
    e = np.array(errs)
    print(f"{name}: {n} seeds from {s0}: max {e.max():.3e} (seed {s0 + int(e.argmax())}), p99 {np.percentile(e, 99):.3e}, p90 {np.percentile(e, 90):.3e}, "
          f"median {np.median(e):.3e}, min {e.min():.3e}; above 9e-7: {(e > 9e-7).sum()}, above 1e-6: {(e > 1e-6).sum()}")
