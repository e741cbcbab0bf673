#!/usr/bin/env python3
"""Generate tests/golden/handoccnet_light_*.npz by running the REAL reference HandOccNet on CPU (dev
container only; stubs as in make_golden.py / _ref_shims.py).  Also writes the parameter manifest
(names + shapes only) hands_amd/manifests/handoccnet_light.json.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_handoccnet.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ref_shims import *  # noqa: F401,F403
from _ref_shims import Args, META, REPO, probe
import numpy as np
import torch
import torch.utils.model_zoo as model_zoo

from hands_amd.weights import apply_recipe, synthetic_inputs

model_zoo.load_url = lambda *a, **k: None                         # no download
import src.models.handoccnet_light.backbone as ref_backbone  # noqa: E402

ref_backbone.ResNet.load_state_dict = lambda self, *a, **k: None   # backbone.py:126 is strict

from src.models.handoccnet_light.model import HandOccNet  # noqa: E402  (the real reference model)
from src.parsers.configs.handoccnet_light import DEFAULT_ARGS_EGO  # noqa: E402


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    args = Args(DEFAULT_ARGS_EGO)
    args.update(focal_length=1000.0, use_render_seg_loss=False)
    model = HandOccNet(1000.0, 224, args)
    apply_recipe(model)
    model.eval()
    for seed in (0, 1):
        inputs, meta_info = synthetic_inputs(2, seed)
        cap = {}

        def keep(name, pick=lambda o: o):
            def hook(m, i, o):
                cap.setdefault(name, pick(o))
            return hook

        hooks = [model.backbone.layer4.register_forward_hook(keep("c5")),
                 model.backbone.smooth3.register_forward_hook(keep("p2_smooth")),
                 model.backbone.register_forward_hook(keep("primary", lambda o: o[0])),
                 model.backbone.register_forward_hook(keep("secondary", lambda o: o[1])),
                 model.FIT.layers[0].register_forward_hook(keep("fit_block0")),
                 model.FIT.register_forward_hook(keep("fit")),
                 model.SET.register_forward_hook(keep("set")),
                 model.regressor.hand_regHead.hg[0].register_forward_hook(keep("hourglass")),
                 model.regressor.hand_regHead.register_forward_hook(keep("heatmaps", lambda o: o[0][-1])),
                 model.regressor.hand_Encoder.register_forward_hook(keep("mano_encoding"))]
        with torch.no_grad():
            out = model(inputs, meta_info)
        for h in hooks:
            h.remove()
        assert len(out) == 22
        rec = {"out/" + k: v.numpy() for k, v in out.items()}
        for name in ("c5", "p2_smooth", "primary", "secondary", "fit_block0", "fit", "set", "hourglass", "heatmaps"):
            for k, v in probe(cap[name]).items():
                rec[f"probe/{name}/{k}"] = v
        rec["mano_encoding"] = cap["mano_encoding"].numpy()
        rec["meta"] = np.array(json.dumps(dict(META, seed=seed, bz=2, model="handoccnet_light")))
        np.savez_compressed(os.path.join(out_dir, f"handoccnet_light_bz2_seed{seed}.npz"), **rec)
        print("seed", seed, "ok; beta.r", out["mano.beta.r"][0, :3].tolist(), "cam", out["mano.cam_t.wp.r"][0].tolist(),
              {n: round(float(cap[n].abs().max()), 3) for n in ("c5", "primary", "fit", "set", "hourglass", "heatmaps", "mano_encoding")})
    params = {n for n, _ in model.named_parameters()}
    manifest = {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
                    "kind": "param" if k in params else "buffer"}
                for k, v in model.state_dict().items() if ".mano." not in k}
    os.makedirs(os.path.join(REPO, "hands_amd", "manifests"), exist_ok=True)
    with open(os.path.join(REPO, "hands_amd", "manifests", "handoccnet_light.json"), "w") as fh:
        json.dump(manifest, fh, indent=0, sort_keys=True)
    print("state_dict tensors:", len(manifest), "params", sum(p.numel() for p in model.parameters()) / 1e6, "M")


if __name__ == "__main__":
    main()
