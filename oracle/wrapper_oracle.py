"""CPU oracle for the wrapper's GT preprocessing.  TEST INFRASTRUCTURE ONLY.
Restates process_data_light (src/callbacks/process/process_arctic.py:4-75) and the 2-D de-normalisation
of GenericWrapper.forward (src/models/generic/wrapper.py:118-134); pinned by tests/golden/process_data.npz,
produced by the reference function itself (MANO layer = oracle LBS, a9 unpinned)."""
import torch

from . import hands_oracle as O


def perspective_to_weak_perspective(cam_t, focal, img_res):
    """common/camera.py:10-29."""
    return torch.stack([2 * focal / (img_res * cam_t[:, 2] + 1e-9), cam_t[:, 0], cam_t[:, 1]], dim=-1)


def process_data_light(targets, K, asset_r, asset_l, img_res=224):
    out = {}
    f = (K[:, 0, 0] + K[:, 1, 1]) / 2.0
    for h, asset in (("r", asset_r), ("l", asset_l)):
        pose, beta, full = targets[f"mano.pose.{h}"], targets[f"mano.beta.{h}"], targets[f"mano.j3d.full.{h}"]
        verts, joints = O.mano_lbs(beta, pose[:, :3], pose[:, 3:], asset)
        out[f"mano.joints3d.{h}"], out[f"mano.vertices.{h}"] = joints, verts
        tr0 = (full - joints).mean(dim=1)
        cam_t = full[:, 0] - joints[:, 0]
        out[f"mano.cam_t.{h}"] = cam_t
        out[f"mano.cam_t.wp.{h}"] = perspective_to_weak_perspective(cam_t, f, img_res)
        out[f"mano.v3d.cam.{h}"] = verts + tr0[:, None, :]
        out[f"mano.j3d.cam.{h}"] = full
    return out


def unnormalize_kp2d(x, img_res):
    """common/data_utils.py:368-373."""
    return 0.5 * img_res * (x + 1)
