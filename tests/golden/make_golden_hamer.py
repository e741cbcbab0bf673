#!/usr/bin/env python3
"""Generate tests/golden/hamer_light_*.npz by running the REAL reference HAMER on CPU (dev container
only; see make_golden.py / _ref_shims.py for the import stubs and which of them carry arithmetic).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hamer.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ref_shims import *  # noqa: F401,F403
from _ref_shims import Args, META, TMP_DIR, probe
import numpy as np
import torch

from hands_amd.weights import apply_recipe, synthetic_inputs, synthetic_mano_mean_params

# the reference reads $DATA_DIR/hamer/_DATA/data/mano_mean_params.npz in the constructor
_d = os.path.join(os.environ["DATA_DIR"], "hamer", "_DATA", "data")
os.makedirs(_d, exist_ok=True)
np.savez(os.path.join(_d, "mano_mean_params.npz"), **synthetic_mano_mean_params())

from src.models.hamer_light.model import HAMER  # noqa: E402  (the real reference model)
from src.parsers.configs.hamer_light import DEFAULT_ARGS_EGO  # noqa: E402


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    args = Args(DEFAULT_ARGS_EGO)
    args.update(focal_length=1000.0, use_render_seg_loss=False, pretrained="none", method="hamer_light")
    model = HAMER(args, 1000.0, 224)
    apply_recipe(model)
    model.eval()
    for seed in (0, 1):
        inputs, meta_info = synthetic_inputs(2, seed)
        cap = {}

        def keep(name, pick=lambda o: o):
            def hook(m, i, o):          # must return None: a returned value would replace the output
                cap.setdefault(name, pick(o))
            return hook

        hooks = [model.backbone.register_forward_hook(keep("vit")),
                 model.backbone.patch_embed.register_forward_hook(keep("patch", lambda o: o[0])),
                 model.backbone.blocks[0].register_forward_hook(keep("block0")),
                 model.backbone.blocks[15].register_forward_hook(keep("block15")),
                 model.backbone.blocks[31].register_forward_hook(keep("block31")),
                 model.mano_head.transformer.register_forward_hook(keep("token_out")),
                 model.kpe.register_forward_hook(lambda m, i, o: cap.setdefault("kpe", []).append(o))]
        with torch.no_grad():
            out = model(inputs, meta_info)
        for h in hooks:
            h.remove()
        assert len(out) == 22
        rec = {"out/" + k: v.numpy() for k, v in out.items()}
        for name in ("patch", "block0", "block15", "block31"):
            for k, v in probe(cap[name].transpose(1, 2)).items():      # (B, N, C) -> channel dim 1
                rec[f"probe/{name}/{k}"] = v
        for k, v in probe(cap["vit"]).items():                        # (B, C, Hp, Wp)
            rec[f"probe/vit/{k}"] = v
        rec["token_out"] = cap["token_out"].squeeze(1).numpy()
        rec["kpe_r"] = cap["kpe"][0][:, 0].numpy()
        rec["kpe_l"] = cap["kpe"][1][:, 0].numpy()
        rec["meta"] = np.array(json.dumps(dict(META, seed=seed, bz=2, model="hamer_light",
                                               mano_mean_params="hands_amd.weights.synthetic_mano_mean_params")))
        np.savez_compressed(os.path.join(out_dir, f"hamer_light_bz2_seed{seed}.npz"), **rec)
        print("seed", seed, "ok; beta.r", out["mano.beta.r"][0, :3].tolist(), "cam", out["mano.cam_t.wp.r"][0].tolist(),
              "|tok|", float(cap["token_out"].abs().max()), "|vit|", float(cap["vit"].abs().max()))
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(out_dir, "hamer_state_dict_keys.json"), "w") as fh:
        json.dump(keys, fh, indent=0, sort_keys=True)
    print("state_dict tensors:", len(keys), "params", sum(p.numel() for p in model.parameters()) / 1e6, "M")


if __name__ == "__main__":
    main()
