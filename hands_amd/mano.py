"""MANO asset container for the hot path.

The reference builds ``smplx.MANO(MANO_DIR, use_pca=False, flat_hand_mean=False, is_rhand=...)``
(reference: common/body_models.py:92-99) and calls it at src/nets/hand_heads/mano_head.py:34-38.
``smplx`` (third party, unpinned) and the licensed ``MANO_{RIGHT,LEFT}.pkl`` files are absent
from /root/reference, so this module provides

* :class:`ManoAsset` -- the arrays the linear-blend-skinning kernels need, in the shapes smplx
  registers them as buffers;
* :func:`synthetic_mano_asset` -- the deterministic stand-in asset (SURVEY.md section 8c recipe) used
  by tests, ``smoke()`` and ``bench.py``;
* :func:`load_mano_pkl` -- a chumpy-free reader for the real files when a user supplies them
  (``$MANO_DIR/MANO_RIGHT.pkl``);
* :class:`ManoLayer` -- an ``nn.Module`` buffer holder so ``state_dict`` carries
  ``mano_{r,l}.mano.<buffer>`` keys like the reference's.

No arithmetic of the hot path lives here; LBS runs in csrc/mano_lbs.hip.
"""
from __future__ import annotations

import os
import pickle
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

NUM_VERTS = 778
NUM_FACES = 1538
NUM_JOINTS = 16
NUM_BETAS = 10
NUM_POSE_FEAT = 135
# kinematic tree of MANO (wrist, then 5 fingers x 3), smplx ``parents`` buffer
PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)
# fingertip vertices appended as joints 16..20 (thumb, index, middle, ring, pinky); the reference
# needs 21 joints (src/nets/hand_heads/mano_head.py:55, hands_light_dataset.py:494)
TIP_IDS = (744, 320, 443, 554, 671)


@dataclass
class ManoAsset:
    v_template: np.ndarray   # (778, 3) f32
    shapedirs: np.ndarray    # (778, 3, 10) f32
    posedirs: np.ndarray     # (135, 2334) f32  == asset (778,3,135).reshape(-1,135).T
    J_regressor: np.ndarray  # (16, 778) f32 dense
    lbs_weights: np.ndarray  # (778, 16) f32
    hands_mean: np.ndarray   # (45,) f32 (added to hand_pose because flat_hand_mean=False)
    faces: np.ndarray        # (1538, 3) int64
    is_rhand: bool = True

    def validate(self):
        assert self.v_template.shape == (NUM_VERTS, 3)
        assert self.shapedirs.shape == (NUM_VERTS, 3, NUM_BETAS)
        assert self.posedirs.shape == (NUM_POSE_FEAT, NUM_VERTS * 3)
        assert self.J_regressor.shape == (NUM_JOINTS, NUM_VERTS)
        assert self.lbs_weights.shape == (NUM_VERTS, NUM_JOINTS)
        assert self.hands_mean.shape == (45,)
        assert self.faces.shape == (NUM_FACES, 3)
        return self


def synthetic_mano_asset(is_rhand: bool = True) -> ManoAsset:
    """Deterministic synthetic asset: seed 0 = right, seed 1 = left."""
    rng = np.random.RandomState(0 if is_rhand else 1)
    v_template = (0.08 * rng.randn(NUM_VERTS, 3)).astype(np.float32)
    shapedirs = (0.005 * rng.randn(NUM_VERTS, 3, NUM_BETAS)).astype(np.float32)
    posedirs = (5e-4 * rng.randn(NUM_POSE_FEAT, NUM_VERTS * 3)).astype(np.float32)

    def sparse_softmax(rows, cols, nnz):
        out = np.zeros((rows, cols), np.float64)
        for r in range(rows):
            idx = rng.choice(cols, nnz, replace=False)
            logits = rng.randn(nnz)
            e = np.exp(logits - logits.max())
            out[r, idx] = e / e.sum()
        return out.astype(np.float32)

    J_regressor = sparse_softmax(NUM_JOINTS, NUM_VERTS, 20)
    lbs_weights = sparse_softmax(NUM_VERTS, NUM_JOINTS, 4)
    hands_mean = (0.3 * rng.randn(45)).astype(np.float32)
    faces = rng.randint(0, NUM_VERTS, (NUM_FACES, 3)).astype(np.int64)
    return ManoAsset(v_template, shapedirs, posedirs, J_regressor, lbs_weights, hands_mean, faces,
                     is_rhand).validate()


class _Stub:
    """Stand-in for any class a MANO pickle references (chumpy.Ch, scipy sparse...)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state


class _ForgivingUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("numpy") or module.startswith("scipy"):
            try:
                return super().find_class(module, name)
            except Exception:
                return _Stub
        if module in ("builtins", "collections", "copyreg", "_codecs"):
            return super().find_class(module, name)
        return _Stub  # chumpy & friends


def _as_array(x):
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "toarray"):
        return np.asarray(x.toarray())
    st = getattr(x, "_state", None)
    if isinstance(st, dict):
        for key in ("x", "r", "a"):
            if key in st:
                return _as_array(st[key])
    raise TypeError(f"cannot convert {type(x)} from MANO pickle")


def load_mano_pkl(path: str, is_rhand: bool) -> ManoAsset:
    """Read a real ``MANO_RIGHT.pkl`` / ``MANO_LEFT.pkl`` without chumpy installed."""
    with open(path, "rb") as f:
        d = _ForgivingUnpickler(f, encoding="latin1").load()
    v_template = _as_array(d["v_template"]).astype(np.float32)
    shapedirs = _as_array(d["shapedirs"]).astype(np.float32)[:, :, :NUM_BETAS]
    pd = _as_array(d["posedirs"]).astype(np.float32)  # (778,3,135)
    posedirs = np.ascontiguousarray(pd.reshape(-1, pd.shape[-1]).T)
    J_regressor = _as_array(d["J_regressor"]).astype(np.float32)
    lbs_weights = _as_array(d["weights"]).astype(np.float32)
    hands_mean = _as_array(d["hands_mean"]).astype(np.float32).reshape(-1)
    faces = _as_array(d["f"]).astype(np.int64)
    return ManoAsset(v_template, shapedirs, posedirs, J_regressor, lbs_weights, hands_mean, faces,
                     is_rhand).validate()


def synthetic_allowed(allow_synthetic=None) -> bool:
    """The synthetic stand-ins (MANO asset, hamer mean parameters) are an explicit opt-in: an argument,
    or HANDS_SYNTHETIC_MANO=1 (set by tests/conftest.py, bench.py, smoke() and the tools)."""
    if allow_synthetic is not None:
        return bool(allow_synthetic)
    return os.environ.get("HANDS_SYNTHETIC_MANO", "") == "1"


def build_mano_asset(is_rhand: bool, allow_synthetic=None) -> ManoAsset:
    """The real asset from ``$MANO_DIR/MANO_{RIGHT,LEFT}.pkl`` (common/body_models.py:90-99).  Without
    it the reference hard-fails (``os.environ['MANO_DIR']``); so does this, unless the synthetic
    stand-in was asked for explicitly -- a model that silently serves a random hand template would
    produce well-shaped, meaningless meshes."""
    mano_dir = os.environ.get("MANO_DIR", "")
    fn = os.path.join(mano_dir, "MANO_RIGHT.pkl" if is_rhand else "MANO_LEFT.pkl")
    if mano_dir and os.path.isfile(fn):
        return load_mano_pkl(fn, is_rhand)
    if synthetic_allowed(allow_synthetic):
        return synthetic_mano_asset(is_rhand)
    raise FileNotFoundError(
        f"hands_amd: MANO asset not found ({fn!r}; $MANO_DIR={mano_dir!r}).  Point $MANO_DIR at the licensed "
        "MANO_RIGHT.pkl / MANO_LEFT.pkl, pass mano_assets=(right, left) ManoAsset objects, or opt in to the "
        "synthetic stand-in with HANDS_SYNTHETIC_MANO=1 (tests / benchmarks only).")


# buffers smplx.MANO(use_pca=False, flat_hand_mean=False) registers (smplx/body_models.py, SMPL.__init__ +
# MANO.__init__; third party, absent here -> names restated from the published source, "parity unpinned").
# Its nn.Parameters (betas, global_orient, hand_pose) and the vertex_joint_selector index buffer are not
# used by the forward path (the reference passes betas / poses explicitly, mano_head.py:34-38) and are
# reported as unexpected keys by load_state_dict(strict=False), exactly like any other extra key.
SMPLX_MANO_BUFFERS = ("faces_tensor", "v_template", "shapedirs", "J_regressor", "posedirs", "parents",
                      "lbs_weights", "hand_mean", "pose_mean")


class ManoLayer(nn.Module):
    """Buffer holder named like smplx.MANO so reference checkpoints load (strict=False)."""

    def __init__(self, asset: ManoAsset):
        super().__init__()
        self.is_rhand = asset.is_rhand
        self.register_buffer("faces_tensor", torch.from_numpy(asset.faces.copy()))
        self.register_buffer("v_template", torch.from_numpy(asset.v_template.copy()))
        self.register_buffer("shapedirs", torch.from_numpy(asset.shapedirs.copy()))
        self.register_buffer("J_regressor", torch.from_numpy(asset.J_regressor.copy()))
        self.register_buffer("posedirs", torch.from_numpy(asset.posedirs.copy()))
        self.register_buffer("parents", torch.tensor(PARENTS, dtype=torch.long))
        self.register_buffer("lbs_weights", torch.from_numpy(asset.lbs_weights.copy()))
        self.register_buffer("hand_mean", torch.from_numpy(asset.hands_mean.copy()))
        pose_mean = np.concatenate([np.zeros(3, np.float32), asset.hands_mean])
        self.register_buffer("pose_mean", torch.from_numpy(pose_mean))

    @property
    def faces(self):
        return self.faces_tensor.cpu().numpy()

    def asset(self) -> ManoAsset:
        t = lambda b: b.detach().cpu().numpy()
        return ManoAsset(t(self.v_template), t(self.shapedirs), t(self.posedirs),
                         t(self.J_regressor), t(self.lbs_weights), t(self.pose_mean)[3:],
                         t(self.faces_tensor), self.is_rhand)
