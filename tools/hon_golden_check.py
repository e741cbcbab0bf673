#!/usr/bin/env python3
"""HandOccNet golden fixtures (tests/golden/handoccnet_light_bz2_seed{0,1}.npz) against the HIP path with the direct 3x3
kernel, Winograd on every 3x3 / stride-1 layer, and Winograd on the trunk only (dev tool, GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.environ.setdefault("HANDS_SYNTHETIC_MANO","1")
import numpy as np, torch, hands_amd
from hands_amd.weights import synthetic_inputs
m=hands_amd.apply_recipe(hands_amd.HandOccNet()).eval().to("cuda"); m.async_forward=False
for scope,w in (("all",False),("all",True),("trunk",True),("backbone",True),("backbone+fit",True)):
    m.engine.winograd=w; m.winograd_scope=scope; m.invalidate_packed()
    for seed in (0,1):
        d=np.load(os.path.join(ROOT, "tests", "golden", f"handoccnet_light_bz2_seed{seed}.npz"))
        i,mt=synthetic_inputs(2,seed,device="cuda"); out=m(i,mt); torch.cuda.synchronize()
        e=[float(np.abs(out[f"mano.vertices.{h}"].cpu().numpy()-d[f"out/mano.vertices.{h}"]).max()) for h in "rl"]
        print("winograd",w,scope,"golden seed",seed,["%.2e"%x for x in e])
