"""Test helper: writes pickles shaped like the licensed MANO_{RIGHT,LEFT}.pkl (chumpy objects, scipy-sparse
J_regressor, the keys smplx ignores) and an asset that looks like a real one rather than like the synthetic
seed assets (common/body_models.py:90-99)."""
import numpy as np

import hands_amd
from hands_amd.mano import ManoAsset, NUM_BETAS, NUM_FACES, NUM_JOINTS, NUM_POSE_FEAT, NUM_VERTS


def realistic_mano_asset(is_rhand, seed=7):
    """An asset of the real file's character: a hand-sized template (~0.2 m), a sparse J_regressor with
    different supports per joint, 2-5 skinning weights per vertex with exact zeros elsewhere, shapedirs of
    decreasing scale per component, a non-trivial mean pose.  NOT the recipe of hands_amd.synthetic_mano_asset."""
    rng = np.random.RandomState(seed + (0 if is_rhand else 100))
    v = rng.rand(NUM_VERTS, 3).astype(np.float64) * np.array([0.19, 0.09, 0.03]) - np.array([0.095, 0.0, 0.015])
    if not is_rhand:
        v[:, 0] *= -1
    sd = rng.randn(NUM_VERTS, 3, NUM_BETAS) * (0.004 / (1 + np.arange(NUM_BETAS)))
    pd = rng.randn(NUM_VERTS, 3, NUM_POSE_FEAT) * 3e-4
    J = np.zeros((NUM_JOINTS, NUM_VERTS))
    for j in range(NUM_JOINTS):
        idx = rng.choice(NUM_VERTS, rng.randint(8, 40), replace=False)
        w = rng.rand(len(idx)) ** 2
        J[j, idx] = w / w.sum()
    W = np.zeros((NUM_VERTS, NUM_JOINTS))
    for i in range(NUM_VERTS):
        idx = rng.choice(NUM_JOINTS, rng.randint(2, 6), replace=False)
        w = rng.rand(len(idx))
        W[i, idx] = w / w.sum()
    mean = rng.randn(45) * 0.25
    faces = rng.randint(0, NUM_VERTS, (NUM_FACES, 3))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return ManoAsset(f32(v), f32(sd), f32(pd.reshape(-1, NUM_POSE_FEAT).T), f32(J), f32(W), f32(mean),
                     faces.astype(np.int64), is_rhand).validate()


def write_fake_mano_pkl(path, asset, protocol=2):
    """A pickle shaped like the licensed MANO_{RIGHT,LEFT}.pkl: a dict whose `shapedirs` (and here also
    `v_template`, `posedirs`) are chumpy.ch.Ch objects -- a class NOT importable at load time --, whose
    `J_regressor` is a scipy.sparse.csc_matrix, plus the keys smplx ignores."""
    import pickle
    import sys
    import types
    import scipy.sparse as sp

    ch_mod = types.ModuleType("chumpy.ch")

    class Ch(object):                          # chumpy.ch.Ch stores its value in the attribute `x`
        def __init__(self, x):
            self.x = np.asarray(x)
            self._dirty_vars = set()
            self._itr = None

    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    ch_mod.Ch = Ch
    pkg = types.ModuleType("chumpy")
    pkg.ch = ch_mod
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = pkg, ch_mod
    try:
        d = {
            "v_template": Ch(asset.v_template.astype(np.float64)),
            "shapedirs": Ch(asset.shapedirs.astype(np.float64)),
            "posedirs": Ch(asset.posedirs.T.reshape(778, 3, 135).astype(np.float64)),
            "J_regressor": sp.csc_matrix(asset.J_regressor.astype(np.float64)),
            "weights": asset.lbs_weights.astype(np.float64),
            "hands_mean": asset.hands_mean.astype(np.float64),
            "f": asset.faces.astype(np.uint32),
            "kintree_table": np.stack([np.array([2 ** 32 - 1] + list(hands_amd.mano.PARENTS[1:]), dtype=np.int64),
                                       np.arange(16)]),
            "hands_components": np.eye(45), "hands_coeffs": np.zeros((10, 45)), "J": np.zeros((16, 3)),
            "bs_style": "lbs", "bs_type": "lrotmin",
        }
        with open(path, "wb") as fh:
            pickle.dump(d, fh, protocol=protocol)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
