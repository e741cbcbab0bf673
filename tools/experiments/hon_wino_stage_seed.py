#!/usr/bin/env python3
"""Which stage's Winograd F(2x2) launches carry a tail seed's error?  HandOccNet default numerics with Winograd restricted to
subsets of stages, max vertex error against fp64 and the live 8-thread fp32 oracle, for a few seeds (round 6 dev experiment).
usage: python tools/experiments/hon_wino_stage_seed.py 2,3,5"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch

import hands_amd
from hands_amd.handoccnet import STAGES
from hands_amd.weights import synthetic_inputs
from oracle import handoccnet_oracle as HO

torch.set_num_threads(8)
seeds = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "2,3,5").split(",")]
m = hands_amd.apply_recipe(hands_amd.HandOccNet())
sd = {k: v.clone() for k, v in m.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
m = m.to("cuda").eval()
m.async_forward = False
c64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
vs = lambda o: torch.stack([o[f"mano.vertices.{h}"] for h in "rl"]).double().cpu()
plans = [("all stages", None)] + [("without " + s, frozenset(STAGES) - {s}) for s in ("resnet", "fpn", "fit", "hourglass", "encoder")] + \
        [("none (direct)", frozenset())]
for seed in seeds:
    ci, cm = synthetic_inputs(2, seed)
    r32 = vs(HO.handoccnet_forward(sd, ar, al, ci, cm))
    r64 = vs(HO.handoccnet_forward(sd64, ar, al, c64(ci), c64(cm)))
    print(f"seed {seed}: reference fp32 vs fp64 {(r32 - r64).abs().max().item():.3e}")
    for name, ws in plans:
        m.wino_stages = ws
        m.invalidate_packed()
        out = vs(m({k: v.to('cuda') for k, v in ci.items()}, {k: v.to('cuda') for k, v in cm.items()}))
        print(f"   Winograd in {name:18s}: vs fp64 {(out - r64).abs().max().item():.3e}   vs the fp32 reference {(out - r32).abs().max().item():.3e}")
