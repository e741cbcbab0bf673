#!/usr/bin/env python3
"""Non-default HandsLight configurations on random inputs beyond their one reference fixture: HIP forward against the oracle
(which reproduces the reference's fixtures, tests/test_oracle_golden.py) for a few more seeds, flips included.
usage: python tools/switch_parity_seeds.py [seeds, default 3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
import hands_amd
from hands_amd.mano import synthetic_mano_asset
from hands_amd.weights import synthetic_dense_inputs, synthetic_inputs
from oracle import hands_oracle as O
from switch_cases import oracle_kwargs

torch.set_num_threads(8)
CONFIGS = {"dense": dict(pos_enc="dense"), "dense_latent": dict(pos_enc="dense_latent"), "cam_conv": dict(pos_enc="cam_conv"),
           "pcl": dict(pos_enc="pcl"), "persp": dict(pos_enc="perspective_correction"), "depth": dict(use_depth_loss=True),
           "arctic": dict(pos_enc=None, no_crops=True, use_glb_feat_w_grasp=False), "separate": dict(separate_hands=True, regress_center_corner=True),
           "noglb": dict(use_glb_feat=False, use_glb_feat_w_grasp=False), "center_corner": dict(pos_enc="center+corner")}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
worst = {}
for name, over in CONFIGS.items():
    args = type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS, **over))
    model = hands_amd.apply_recipe(hands_amd.HandsLight(args=args)).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    errs = []
    for seed in range(100, 100 + n):
        inputs, meta = synthetic_inputs(3, seed)
        if over.get("pos_enc") in ("dense", "dense_latent", "cam_conv", "pcl"):
            inputs.update(synthetic_dense_inputs(3, seed, over["pos_enc"]))
        meta["is_flipped"] = torch.tensor([0, seed % 2, 0])
        ref = O.hands_light_forward(sd, synthetic_mano_asset(True), synthetic_mano_asset(False), inputs, meta, **oracle_kwargs(over))
        out = model({k: v.to(dev) for k, v in inputs.items()}, {k: v.to(dev) for k, v in meta.items()})
        assert sorted(out.keys()) == sorted(ref.keys()), name
        e = max((out[f"mano.vertices.{h}"].cpu() - ref[f"mano.vertices.{h}"]).abs().max().item() for h in "rl")
        g = max(((out[k].cpu() - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1.0)).item() for k in ref if not k.startswith("mano."))  if any(not k.startswith("mano.") for k in ref) else 0.0
        errs.append((e, g))
    worst[name] = (max(e for e, _ in errs), max(g for _, g in errs))
    print(f"{name:14s} max vertex error {worst[name][0]:.3e} m, other outputs (relative to max(1, |ref|_max)) {worst[name][1]:.2e}", flush=True)
print("worst vertex error over all:", max(v[0] for v in worst.values()))
