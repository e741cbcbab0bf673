#!/usr/bin/env python3
"""HandOccNet: where does the end-to-end fp32 error come from?  Per stage, the max error (relative to the stage's value
range) of the HIP path and of the fp32 oracle (= the reference's arithmetic) against an fp64 evaluation of the same
network, for the direct / Winograd-backbone / Winograd-all routes (dev tool, GPU box).
"ideal" = the oracle with exact accumulation and fp32 storage (tools/experiments/hon_ideal.py): the floor of any fp32-activation path.
usage: python tools/hon_error_stages.py [n_seeds] [route,route,...]   route: direct | backbone | all, + "+f64all" or "+f:stage.stage"
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import handoccnet_oracle as HO
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments"))
import hon_ideal   # noqa: E402  (patches the oracle's conv / linear / matmul: a no-op unless asked for)

torch.set_num_threads(16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model = hands_amd.apply_recipe(hands_amd.HandOccNet())
sd = {k: v.clone() for k, v in model.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
model = model.to("cuda").eval()
model.async_forward = False
c64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}


def hip_stages(dbg, out, B2):
    nchw = lambda t, H, W, C: t.reshape(B2, H, W, -1)[..., :C].permute(0, 3, 1, 2)
    s = {}
    s["primary"] = nchw(dbg["primary"], 32, 32, 256)
    s["secondary"] = nchw(dbg["secondary"], 32, 32, 256)
    s["fit"] = nchw(dbg["fit"], 32, 32, 256)
    s["set"] = nchw(dbg["set"], 32, 32, 256)
    s["hourglass"] = nchw(dbg["hourglass"], 32, 32, 256)
    s["heatmaps"] = nchw(dbg["heat"], 32, 32, 21)
    s["mano_encoding"] = nchw(dbg["enc"], 2, 2, 256).reshape(B2, -1)
    s["pose6d"] = dbg["pred"][:, :96]
    s["verts"] = torch.cat([out["mano.vertices.r"], out["mano.vertices.l"]], 0)
    return {k: v.detach().cpu().double() for k, v in s.items()}


def ref_stages(out, inter):
    s = {k: inter[k] for k in ("primary", "secondary", "fit", "set", "hourglass", "heatmaps", "mano_encoding", "pose6d")}
    s["verts"] = torch.cat([out["mano.vertices.r"], out["mano.vertices.l"]], 0)
    return {k: v.double() for k, v in s.items()}


from hands_amd.handoccnet import STAGES
ROUTES = (sys.argv[2].split(",") if len(sys.argv) > 2 else ["direct", "backbone", "all", "direct+f64all"])
names = ("primary", "secondary", "fit", "set", "hourglass", "heatmaps", "mano_encoding", "pose6d", "verts")
worst = {}
for seed in range(n):
    ci, cm = synthetic_inputs(2, seed)
    r32 = ref_stages(*HO.handoccnet_forward(sd, ar, al, ci, cm, return_intermediates=True))
    r64 = ref_stages(*HO.handoccnet_forward(sd64, ar, al, c64(ci), c64(cm), return_intermediates=True))
    rows = {"ref32": r32, "ideal": ref_stages(*hon_ideal.forward(sd, ar, al, ci, cm, return_intermediates=True))}
    for route in ROUTES:
        model.engine.winograd = not route.startswith("direct")
        model.winograd_scope = "all" if route.startswith("all") else "backbone"
        model.acc64_stages = frozenset(STAGES) if route.endswith("f64all") else (frozenset(route.split("+f:")[1].split(".")) if "+f:" in route else frozenset())
        model.invalidate_packed()
        model.__dict__["_debug"] = dbg = {}
        out = model({k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()})
        torch.cuda.synchronize()
        model.__dict__.pop("_debug")
        rows[route] = hip_stages(dbg, out, 4)
    print(f"seed {seed}: rms(x - fp64) / rms(fp64)   (verts: max abs, metres)")
    for k in names:
        sc = 1.0 if k == "verts" else r64[k].pow(2).mean().sqrt().item()
        line = f"  {k:14s}"
        for nm, st in rows.items():
            e = ((st[k] - r64[k]).abs().max().item() if k == "verts" else (st[k] - r64[k]).pow(2).mean().sqrt().item()) / sc
            worst[(k, nm)] = max(worst.get((k, nm), 0.0), e)
            line += f" {nm} {e:.2e}"
        print(line)
print("worst over seeds:")
for k in names:
    print(f"  {k:14s}" + "".join(f" {nm} {worst[(k, nm)]:.2e}" for nm in ["ref32", "ideal"] + list(ROUTES)))
