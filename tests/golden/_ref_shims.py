"""Import stubs that let the REAL reference be imported in the dev container (no pytorch3d, smplx,
cv2, loguru, trimesh, torchvision, timm, easydict here).  Dev-container only; never shipped to or
imported on the GPU box; nothing under tests/test_*.py imports it.  See make_golden.py for which
stubs are inert and which carry (flagged) arithmetic."""
import json
import os
import sys
import tempfile
import types

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import torch

from hands_amd.mano import synthetic_mano_asset
from hands_amd.weights import apply_recipe, synthetic_inputs
from oracle import hands_oracle as O

torch.set_num_threads(8)
_tmp = tempfile.mkdtemp()
os.environ.setdefault("MANO_DIR", _tmp)
os.environ.setdefault("DATA_DIR", _tmp)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    def __getattr__(self, k):
        return lambda *a, **kw: None


# ---- inert stubs ---------------------------------------------------------------------------------
_mod("cv2")
_mod("loguru", logger=_Anything())
_mod("trimesh", Trimesh=object)
_mod("torchvision", ops=types.ModuleType("ops"))
_mod("pytorch3d")
_mod("pytorch3d.structures", Meshes=object)
_mod("pytorch3d.renderer", **{n: object for n in (
    "look_at_view_transform", "FoVPerspectiveCameras", "PerspectiveCameras", "RasterizationSettings",
    "MeshRenderer", "MeshRasterizer", "SoftSilhouetteShader", "BlendParams", "TexturesVertex",
    "PointLights", "SoftPhongShader", "HardPhongShader")})

import common.rot as ref_rot  # real reference code (needs the cv2 stub above)


def _ref_axis_angle_to_matrix(aa):
    # pytorch3d defines this as the composition of two functions the reference vendors verbatim
    # (common/rot.py:754-782 and :86-115); the composition itself is the only thing added here.
    return ref_rot.quaternion_to_matrix(ref_rot.axis_angle_to_quaternion(aa))


# ---- arithmetic stubs (flagged) ------------------------------------------------------------------
_p3d_t = _mod("pytorch3d.transforms")
_rc = _mod("pytorch3d.transforms.rotation_conversions",
           rotation_6d_to_matrix=O.rotation_6d_to_matrix,
           matrix_to_rotation_6d=O.matrix_to_rotation_6d,
           axis_angle_to_matrix=_ref_axis_angle_to_matrix,
           matrix_to_axis_angle=ref_rot.matrix_to_axis_angle,
           euler_angles_to_matrix=lambda e, convention: O.euler_angles_to_matrix_xyz(e) if convention == "XYZ" else None)
_p3d_t.rotation_conversions = _rc
sys.modules["pytorch3d"].transforms = _p3d_t


class _ManoOut:
    def __init__(self, v, j):
        self.vertices, self.joints = v, j


class _StubMANO(torch.nn.Module):
    def __init__(self, model_path, create_transl=False, use_pca=False, flat_hand_mean=False,
                 is_rhand=True, **kw):
        super().__init__()
        assert not use_pca and not flat_hand_mean
        self.asset = synthetic_mano_asset(is_rhand)
        self.faces = self.asset.faces

    def forward(self, betas, hand_pose, global_orient, **kw):
        v, j = O.mano_lbs(betas, global_orient, hand_pose, self.asset)
        return _ManoOut(v, j)


_mod("smplx", MANO=_StubMANO)


# timm helpers used by hamer_light/vit.py:10 (inert: tuple helper, identity drop_path in eval, torch init)
_mod("timm")
_mod("timm.models")
_mod("timm.models.layers", to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
     drop_path=lambda x, p=0.0, training=False: x, trunc_normal_=torch.nn.init.trunc_normal_)


class Args(dict):
    __getattr__ = dict.get


META = {"reference": "ap229997/hands @ 2024-10-22", "torch": torch.__version__,
        "arithmetic_stubs": ["pytorch3d rotation_6d_to_matrix/matrix_to_rotation_6d/euler_angles_to_matrix -> oracle (a5 unpinned)",
                             "smplx.MANO -> oracle.mano_lbs on synthetic asset (a9 unpinned)"],
        "weights": "hands_amd.weights.apply_recipe", "inputs": "hands_amd.weights.synthetic_inputs"}
TMP_DIR = _tmp


def probe(t, n=64, seed=0):
    """per-channel mean / abs-max + n probed elements of a (B,C,...) tensor."""
    t = t.detach().float()
    flat = t.reshape(-1)
    idx = torch.from_numpy(np.random.RandomState(seed).randint(0, flat.numel(), n))
    red = tuple(i for i in range(t.ndim) if i != 1)
    return {"mean_c": t.mean(dim=red).numpy(), "absmax_c": t.abs().amax(dim=red).numpy(),
            "idx": idx.numpy(), "val": flat[idx].numpy(), "shape": np.array(t.shape)}
