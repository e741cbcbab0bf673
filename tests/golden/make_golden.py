#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on CPU.  Dev-container only.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference (read-only); never runs on the GPU box and nothing in tests/ imports it.

The reference cannot be imported as shipped (pytorch3d, smplx, cv2, loguru, trimesh, torchvision,
easydict are not installed), so it is imported under stubs:

* INERT stubs (no arithmetic): cv2, loguru, trimesh, pytorch3d.{structures,renderer}, torchvision,
  and a patched ``load_state_dict_from_url`` (no download).
* ARITHMETIC stubs, flagged in every fixture's ``meta`` entry:
    - ``pytorch3d.transforms.rotation_conversions.rotation_6d_to_matrix`` / ``matrix_to_rotation_6d``
      -> oracle restatement (a5, parity unpinned);  ``matrix_to_axis_angle`` / ``axis_angle_to_matrix``
      -> the reference's own vendored copies in common/rot.py (real reference code);
    - ``smplx.MANO`` -> oracle ``mano_lbs`` on the synthetic asset (a9, parity unpinned).
  Everything else that runs -- ResNet trunks, KPE, feature_conv, HandHMR/HMRLayer, grasp MLP,
  MANOHead's camera/projection, matrix_to_axis_angle -- is the reference's own code.

Fixtures hold data only: input seeds, probed intermediate values and the 22 output tensors.
"""
import json
import os
import sys
import tempfile
import types

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ref_shims import *  # noqa: F401,F403  (sets up sys.modules stubs + sys.path)
from _ref_shims import O, ref_rot, _ref_axis_angle_to_matrix, Args, probe, META
import numpy as np
import torch
from hands_amd.weights import apply_recipe, synthetic_inputs

import src.nets.backbone.resnet as ref_resnet

ref_resnet.load_state_dict_from_url = lambda *a, **k: {}

from src.models.hands_light.model import HandsLight  # noqa: E402  (the real reference model)
from src.parsers.configs.hands_light import DEFAULT_ARGS_EGO  # noqa: E402


def build_reference():
    args = Args(DEFAULT_ARGS_EGO)
    args.update(focal_length=1000.0, use_render_seg_loss=False)
    model = HandsLight("resnet50", 1000.0, 224, args)
    apply_recipe(model)
    model.eval()
    return model


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    model = build_reference()
    meta = dict(META)
    only_bz1 = "--bz1-only" in sys.argv      # add the BASELINE configs[0] fixture without touching the others

    # ---- full forward: bz=2, seeds 0..2 (seed 2 has one flipped sample); bz=1, seed 0 = BASELINE
    #      configs[0] "single 224x224 crop, bs=1"  -----------------------------
    for bz, seed in ((1, 0),) if only_bz1 else ((2, 0), (2, 1), (2, 2), (1, 0)):
        inputs, meta_info = synthetic_inputs(bz, seed)
        if seed == 2:
            meta_info["is_flipped"] = torch.tensor([0, 1])
        cap = {}
        hooks = [
            model.backbone.register_forward_hook(lambda m, i, o: cap.setdefault("features", o)),
            model.hand_backbone.register_forward_hook(lambda m, i, o: cap.setdefault("hand_feat", []).append(o)),
            model.feature_conv.register_forward_hook(lambda m, i, o: cap.setdefault("fc", []).append(o)),
            model.backbone.layer1.register_forward_hook(lambda m, i, o: cap.setdefault("layer1", o)),
            model.backbone.layer2.register_forward_hook(lambda m, i, o: cap.setdefault("layer2", o)),
            model.backbone.layer3.register_forward_hook(lambda m, i, o: cap.setdefault("layer3", o)),
            model.head_r.register_forward_hook(lambda m, i, o: cap.setdefault("hmr_r", o)),
            model.head_l.register_forward_hook(lambda m, i, o: cap.setdefault("hmr_l", o)),
        ]
        with torch.no_grad():
            out = model(inputs, meta_info)
        for h in hooks:
            h.remove()
        assert len(out) == 22, sorted(out.keys())
        rec = {"out/" + k: v.numpy() for k, v in out.items()}
        rec["is_flipped"] = meta_info["is_flipped"].numpy()
        for name in ("layer1", "layer2", "layer3", "features"):
            for k, v in probe(cap[name]).items():
                rec[f"probe/{name}/{k}"] = v
        for i, hn in enumerate(("r", "l")):
            for k, v in probe(cap["hand_feat"][i]).items():
                rec[f"probe/hand_feat_{hn}/{k}"] = v
            rec[f"feature_conv_{hn}"] = cap["fc"][i].numpy()
        for hn in ("r", "l"):
            h = cap["hmr_" + hn]
            rec[f"hmr_{hn}/pose_6d"] = h["pose_6d"].numpy()
            rec[f"hmr_{hn}/shape"] = h["shape"].numpy()
            rec[f"hmr_{hn}/cam_t.wp"] = h["cam_t.wp"].numpy()
        rec["meta"] = np.array(json.dumps(dict(meta, seed=seed, bz=bz)))
        np.savez_compressed(os.path.join(out_dir, f"hands_light_bz{bz}_seed{seed}.npz"), **rec)
        print("seed", seed, "ok;  |verts.r| max", float(out["mano.vertices.r"].abs().max()),
              " beta.r", out["mano.beta.r"][0, :3].tolist(), " cam", out["mano.cam_t.wp.r"][0].tolist())
    if only_bz1:
        return

    # ---- rotation conversions: reference common/rot.py on random + adversarial rotations ----
    g = torch.Generator().manual_seed(7)
    R = O.rotation_6d_to_matrix(torch.randn(500, 6, generator=g))
    axes = torch.nn.functional.normalize(torch.randn(12, 3, generator=g), dim=-1)
    thetas = torch.tensor([0.0, 1e-7, 1e-4, math_pi() - 1e-4, math_pi(), 3.0, 0.5e-6, 2e-6, 1.0, 2.0, 3.1, 1e-3])
    Radv = _ref_axis_angle_to_matrix(axes * thetas[:, None])
    Rall = torch.cat([R, Radv, torch.eye(3)[None]], 0)
    np.savez_compressed(os.path.join(out_dir, "rot_conversions.npz"),
                        R=Rall.numpy(), aa=ref_rot.matrix_to_axis_angle(Rall).numpy(),
                        quat=ref_rot.matrix_to_quaternion(Rall).numpy(),
                        aa_in=(axes * thetas[:, None]).numpy(), R_from_aa=Radv.numpy(),
                        meta=np.array(json.dumps(dict(meta, what="common/rot.py matrix_to_axis_angle, "
                                                      "matrix_to_quaternion, axis_angle_to_matrix"))))

    # ---- 6D twins: in-repo hamer geometry.rot6d_to_rotmat must equal the TRANSPOSE of a5 ----
    from src.models.hamer_light.geometry import rot6d_to_rotmat
    d6 = torch.randn(64, 6, generator=g)
    np.savez_compressed(os.path.join(out_dir, "rot6d_twin.npz"), d6=d6.numpy(),
                        hamer_rotmat=rot6d_to_rotmat(d6).numpy(),
                        meta=np.array(json.dumps(dict(meta, what="src/models/hamer_light/geometry.py:47-62; "
                                                      "a1=x[:3], a2=x[3:], b1,b2,b3 stacked as COLUMNS"))))

    # ---- camera / projection: reference common/camera.py, transforms.py, data_utils.py ------
    import common.camera as ref_cam
    import common.data_utils as ref_du
    import common.transforms as ref_tf
    s = torch.tensor([-1.0, 0.05, 0.1, 1.0, 5.0, 0.7, 1.3, 2.0])
    cam = torch.stack([s, 0.1 * torch.randn(8, generator=g), 0.1 * torch.randn(8, generator=g)], -1)
    Kc = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]])[None].repeat(8, 1, 1)
    Kc[:, 0, 0] += 50 * torch.randn(8, generator=g)
    Kc[:, 1, 1] += 50 * torch.randn(8, generator=g)
    Kc[:, 0, 2] += 5 * torch.randn(8, generator=g)
    f = (Kc[:, 0, 0] + Kc[:, 1, 1]) / 2
    cam_t = ref_cam.weak_perspective_to_perspective_torch(cam, focal_length=f, img_res=224, min_s=0.1)
    pts = 0.1 * torch.randn(8, 21, 3, generator=g) + cam_t[:, None]
    j2d = ref_du.normalize_kp2d(ref_tf.project2d_batch(Kc, pts), 224)
    np.savez_compressed(os.path.join(out_dir, "camera_projection.npz"), cam=cam.numpy(), K=Kc.numpy(),
                        cam_t=cam_t.numpy(), pts=pts.numpy(), j2d_norm=j2d.numpy(),
                        meta=np.array(json.dumps(dict(meta, what="camera.py:456-474, transforms.py:316-329, data_utils.py:361-365"))))

    # ---- KPE: reference HandsLight.compute_center_pos_enc / compute_corner_pos_enc ----------
    ang2 = 0.5 * torch.randn(5, 2, generator=g)
    ang8 = 0.5 * torch.randn(5, 8, generator=g)
    np.savez_compressed(os.path.join(out_dir, "kpe.npz"), center_angle=ang2.numpy(), corner_angle=ang8.numpy(),
                        center_enc=model.compute_center_pos_enc(ang2).numpy(),
                        corner_enc=model.compute_corner_pos_enc(ang8).numpy(),
                        meta=np.array(json.dumps(dict(meta, what="model.py:444-460"))))

    # ---- state_dict key inventory (names + shapes only) -------------------------------------
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(out_dir, "state_dict_keys.json"), "w") as fh:
        json.dump(keys, fh, indent=0, sort_keys=True)
    print("state_dict tensors:", len(keys))


def math_pi():
    import math
    return math.pi


if __name__ == "__main__":
    main()
