#!/usr/bin/env python3
"""Generate tests/golden/eval_metrics.npz with the REAL reference metric functions
(src/utils/eval_modules.py:97-134,136-343,386-428; common/metrics.py:23-55).  Dev container only.

src/utils/eval_modules.py raises NameError at import (an undefined name in its module-level dict,
:711); the name is pre-seeded as a builtin placeholder so the module imports -- none of the functions
exercised here touch it.
"""
import builtins
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ref_shims import *  # noqa: F401,F403
from _ref_shims import META, _mod
import numpy as np
import torch

_mod("pytorch3d.ops", knn_points=None)
sys.modules["pytorch3d"].ops = sys.modules["pytorch3d.ops"]
builtins.eval_mpjpe_mano = None
import src.utils.eval_modules as em  # noqa: E402  (real reference code)


def main():
    g = torch.Generator().manual_seed(11)
    B = 24
    gt_r, gt_l = 0.1 * torch.randn(B, 21, 3, generator=g), 0.1 * torch.randn(B, 21, 3, generator=g)
    gt_r[..., 2] += 0.6
    gt_l[..., 2] += 0.7
    gt_l[..., 0] += 0.2
    # predictions: similarity-transformed + noisy copies, so Procrustes has something to undo
    def perturb(x, i):
        ang = torch.tensor(0.2 + 0.05 * i)
        Rz = torch.tensor([[torch.cos(ang), -torch.sin(ang), 0], [torch.sin(ang), torch.cos(ang), 0], [0, 0, 1.0]])
        return (1.1 * (x - x.mean(1, keepdim=True)) @ Rz.T) + x.mean(1, keepdim=True) + 0.01 * torch.randn(x.shape, generator=g)
    pr_r, pr_l = perturb(gt_r, 1), perturb(gt_l, 2)
    pr_r[3] = gt_r[3]                                   # exact match -> zero error
    j2d_gt_r, j2d_gt_l = 224 * torch.rand(B, 21, 2, generator=g), 224 * torch.rand(B, 21, 2, generator=g)
    j2d_pr_r, j2d_pr_l = j2d_gt_r + 3 * torch.randn(B, 21, 2, generator=g), j2d_gt_l + 3 * torch.randn(B, 21, 2, generator=g)
    is_valid = (torch.rand(B, generator=g) > 0.1).float()
    right_valid = (torch.rand(B, generator=g) > 0.2).float()
    left_valid = (torch.rand(B, generator=g) > 0.2).float()
    jv_r = (torch.rand(B, 21, generator=g) > 0.15).float()
    jv_l = (torch.rand(B, 21, generator=g) > 0.15).float()
    pred = {"mano.j3d.cam.r": pr_r, "mano.j3d.cam.l": pr_l, "mano.j2d.r": j2d_pr_r, "mano.j2d.l": j2d_pr_l}
    targets = {"mano.j3d.cam.r": gt_r, "mano.j3d.cam.l": gt_l, "mano.j2d.r": j2d_gt_r, "mano.j2d.l": j2d_gt_l,
               "is_valid": is_valid, "right_valid": right_valid, "left_valid": left_valid,
               "joints_valid_r": jv_r, "joints_valid_l": jv_l}
    meta_info = {"dataset": ["arctic"] * B}
    out = {}
    out.update(em.eval_mpjpe_ra(pred, targets, meta_info))
    out.update(em.eval_mpjpe_pa_ra(pred, targets, meta_info))
    out.update(em.eval_mrrpe_hand(pred, targets, meta_info))
    out.update(em.eval_pixel_error(pred, targets, meta_info))
    rec = {"in/" + k: v.numpy() for k, v in {**{"pred." + k: v for k, v in pred.items()},
                                             **{"targets." + k: v for k, v in targets.items()}}.items()}
    rec.update({"out/" + k: np.asarray(v) for k, v in out.items()})
    rec["meta"] = np.array(json.dumps(dict(META, what="eval_modules.py eval_mpjpe_ra / eval_mpjpe_pa_ra (21-joint branch) / "
                                           "eval_mrrpe_hand / eval_pixel_error")))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_metrics.npz"), **rec)
    for k, v in out.items():
        print(k, np.asarray(v).shape, np.asarray(v).reshape(-1)[:4])


if __name__ == "__main__":
    main()


def make_process_data():
    """tests/golden/process_data.npz: the REAL process_data_light (src/callbacks/process/process_arctic.py:4-75)
    with the MANO layers stubbed by oracle.mano_lbs on the synthetic asset (a9, flagged)."""
    import smplx  # the stub installed by _ref_shims
    from src.callbacks.process.process_arctic import process_data_light
    from _ref_shims import Args
    g = torch.Generator().manual_seed(5)
    B = 6
    targets = {}
    for h in "rl":
        targets[f"mano.pose.{h}"] = 0.4 * torch.randn(B, 48, generator=g)
        targets[f"mano.beta.{h}"] = torch.randn(B, 10, generator=g)
        targets[f"mano.j3d.full.{h}"] = 0.1 * torch.randn(B, 21, 3, generator=g) + torch.tensor([0.05, -0.02, 0.6])
    K = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]])[None].repeat(B, 1, 1)
    K[:, 0, 0] += 30 * torch.randn(B, generator=g)
    models = {"mano_r": smplx.MANO("", is_rhand=True), "mano_l": smplx.MANO("", is_rhand=False)}
    tin = {k: v.clone() for k, v in targets.items()}
    _, tout, _ = process_data_light(models, {}, targets, {"intrinsics": K}, "test", Args(img_res=224))
    rec = {"in/" + k: v.numpy() for k, v in tin.items()}
    rec["in/intrinsics"] = K.numpy()
    rec.update({"out/" + k: v.numpy() for k, v in tout.items() if k not in tin})
    rec["meta"] = np.array(json.dumps(dict(META, what="process_arctic.py:4-75 process_data_light")))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "process_data.npz"), **rec)
    print("process_data keys:", sorted(k for k in rec if k.startswith("out/")))


if __name__ == "__main__":
    make_process_data()
