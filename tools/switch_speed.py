#!/usr/bin/env python3
"""Throughput of a few non-default HandsLight configurations at bz = 256 (dev tool, GPU box): python tools/switch_speed.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch, hands_amd
from hands_amd.weights import synthetic_inputs, synthetic_dense_inputs
dev = torch.device("cuda:0")
for name, over in (("default", {}), ("use_depth_loss", dict(use_depth_loss=True)), ("dense", dict(pos_enc="dense")), ("dense_latent", dict(pos_enc="dense_latent")), ("arctic(no_crops)", dict(pos_enc=None, no_crops=True, use_glb_feat_w_grasp=False))):
    args = type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS, **over))
    m = hands_amd.apply_recipe(hands_amd.HandsLight(args=args)).eval().to(dev)
    bz = 256
    inputs, meta = synthetic_inputs(bz, 0)
    if over.get("pos_enc") in ("dense", "dense_latent"):
        inputs.update(synthetic_dense_inputs(bz, 0, over["pos_enc"]))
    inputs = {k: v.to(dev) for k, v in inputs.items()}; meta = {k: v.to(dev) for k, v in meta.items()}
    for _ in range(3): dict(m(inputs, meta).items())
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): out = m(inputs, meta)
    dict(out.items()); torch.cuda.synchronize()
    dt = (time.time() - t) / 10
    print(f"{name:18s} bz=256: {dt*1e3:6.2f} ms per forward, {2*bz/dt:8.0f} hands/s", flush=True)
    del m, out; torch.cuda.empty_cache()
