#!/usr/bin/env python3
"""HandOccNet: distribution of the end-to-end vertex error of the HIP path against the oracle (fp32 CPU port of the
reference) and against an fp64 evaluation of the same network, with the direct 3x3 kernel and with Winograd F(2x2,3x3),
over several input seeds (dev tool, GPU box).  usage: python tools/hon_parity_sweep.py [n_seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import handoccnet_oracle as HO

torch.set_num_threads(16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = hands_amd.apply_recipe(hands_amd.HandOccNet())
sd = {k: v.clone() for k, v in model.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
model = model.to("cuda").eval()
model.async_forward = False
worst = {}
for seed in range(n):
    ci, cm = synthetic_inputs(2, seed)
    ref32 = HO.handoccnet_forward(sd, ar, al, ci, cm)
    c64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
    try:
        ref64 = HO.handoccnet_forward(sd64, ar, al, c64(ci), c64(cm))
    except Exception as e:      # the oracle may not be dtype-generic everywhere
        ref64 = None
    row = []
    for wino in (False, True, "backbone"):
        model.engine.winograd = bool(wino)
        scope = "backbone" if wino == "backbone" else "all"
        if model.winograd_scope != scope:
            model.winograd_scope = scope
            model.invalidate_packed()
        out = model({k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()})
        torch.cuda.synchronize()
        e32 = max((out[f"mano.vertices.{h}"].cpu() - ref32[f"mano.vertices.{h}"]).abs().max().item() for h in "rl")
        e64 = max((out[f"mano.vertices.{h}"].cpu().double() - ref64[f"mano.vertices.{h}"]).abs().max().item() for h in "rl") if ref64 else float("nan")
        row += [e32, e64]
        nm = {False: "direct", True: "wino", "backbone": "wino-backbone"}[wino]
        for key, v in ((nm + " vs ref fp32", e32), (nm + " vs fp64", e64)):
            worst[key] = max(worst.get(key, 0.0), v)
    r64 = max((ref32[f"mano.vertices.{h}"].double() - ref64[f"mano.vertices.{h}"]).abs().max().item() for h in "rl") if ref64 else float("nan")
    worst["ref fp32 vs fp64"] = max(worst.get("ref fp32 vs fp64", 0.0), r64)
    print(f"seed {seed}: direct vs ref32 {row[0]:.2e} vs fp64 {row[1]:.2e} | wino vs ref32 {row[2]:.2e} vs fp64 {row[3]:.2e} | "
          f"wino-backbone vs ref32 {row[4]:.2e} vs fp64 {row[5]:.2e} | ref32 vs fp64 {r64:.2e}")
print("worst over seeds:", {k: f"{v:.2e}" for k, v in worst.items()})
