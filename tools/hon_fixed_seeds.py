#!/usr/bin/env python3
"""HandOccNet: the FIXED seeds the -m gpu tests use, per arm (dev tool, GPU box): the two golden fixtures (the reference's own output,
generated in the dev container), the 10-seed sweep (10-19) and the 48 guard seeds (200-247) against the LIVE oracle of this box
(8 ATen threads, as tests/conftest.py pins).  usage: python tools/hon_fixed_seeds.py [arm,arm,...]   (arm syntax: tools/hon_parity_ab.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs
from hon_parity_ab import parse_arm
from oracle import handoccnet_oracle as HO

torch.set_num_threads(min(8, os.cpu_count() or 1))
arms = (sys.argv[1] if len(sys.argv) > 1 else "backbone,all,all+c256k512,all+c128i,all+c64i").split(",")
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
base = hands_amd.apply_recipe(hands_amd.HandOccNet())
sd = {k: v.clone() for k, v in base.state_dict().items()}
models = {}
for name in arms:
    scope, limit, min_k, max_pix, skip_tok, in_kernel = parse_arm(name)
    m = hands_amd.apply_recipe(hands_amd.HandOccNet()).to("cuda").eval()
    m.engine.winograd = scope != "direct"
    m.winograd_scope = scope if scope != "direct" else "backbone"
    m.engine.chain_limit, m.engine.chain_min_k, m.engine.chain_max_pix = limit, min_k, max_pix
    m.engine.chain_skip_tokens, m.engine.chain_in_kernel = skip_tok, in_kernel
    m.invalidate_packed()
    m.async_forward = False
    models[name] = m


def verts(out):
    return torch.stack([out[f"mano.vertices.{h}"] for h in "rl"]).cpu().numpy()


res = {n: {"golden": [], "sweep": [], "guard": []} for n in arms}
for seed in (0, 1):
    d = np.load(os.path.join(ROOT, "tests", "golden", f"handoccnet_light_bz2_seed{seed}.npz"))
    ref = np.stack([d[f"out/mano.vertices.{h}"] for h in "rl"])
    i, mt = synthetic_inputs(2, seed, device="cuda")
    live = verts(HO.handoccnet_forward(sd, ar, al, *synthetic_inputs(2, seed)))
    print(f"golden seed {seed}: live oracle of this box vs the fixture {np.abs(live - ref).max():.3e}")
    for n, m in models.items():
        res[n]["golden"].append(float(np.abs(verts(m(i, mt)) - ref).max()))
for key, seeds in (("sweep", range(10, 20)), ("guard", range(200, 248))):
    for seed in seeds:
        ci, cm = synthetic_inputs(2, seed)
        ref = verts(HO.handoccnet_forward(sd, ar, al, ci, cm))
        gi, gm = {k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()}
        for n, m in models.items():
            res[n][key].append(float(np.abs(verts(m(gi, gm)) - ref).max()))
for n in arms:
    g, s, q = (np.array(res[n][k]) for k in ("golden", "sweep", "guard"))
    print(f"{n:16s} golden {g[0]:.2e} {g[1]:.2e} | sweep max {s.max():.2e} | guard: median {np.median(q):.2e} p90 {np.percentile(q, 90):.2e} "
          f"max {q.max():.2e}, > 1e-6: {(q > 1e-6).sum()} of {len(q)}")
