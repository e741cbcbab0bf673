#!/usr/bin/env python3
"""One warm hands_light forward bracketed by marker kernels (dev tool for kernel-trace ordering questions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
import hands_amd
bz = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
model.overlap_trunks = False
inputs, meta = hands_amd.synthetic_inputs(bz, seed=0, device=torch.device("cuda"))
with torch.no_grad():
    for _ in range(2):
        model(inputs, meta)["mano.v3d.cam.r"]
    torch.cuda.synchronize()
    m = torch.zeros(7, device="cuda"); m.fill_(1.0); torch.cuda.synchronize()       # marker
    model(inputs, meta)["mano.v3d.cam.r"]
    torch.cuda.synchronize()
    m.fill_(2.0); torch.cuda.synchronize()
