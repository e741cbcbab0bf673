"""f3 on the device: assets and checkpoints in the reference's ON-DISK formats go
through the loaders and then through the HIP path, and are compared with the oracle on the SAME loaded data.

  * common/body_models.py:90-99 -- ``smplx.MANO($MANO_DIR, use_pca=False, flat_hand_mean=False, is_rhand=...)``
    reads the chumpy pickles ``MANO_RIGHT.pkl`` / ``MANO_LEFT.pkl``  ->  ``hands_amd.mano.load_mano_pkl``;
  * scripts_method/train.py:34-37 -- ``load_state_dict(torch.load(ckpt)['state_dict'], strict=False)`` on the
    Lightning wrapper (keys prefixed ``model.``)                   ->  ``hands_amd.checkpoint.load_reference_checkpoint``.

The licensed files are absent, so the pickles are written here in the real files' format (tests/fake_mano.py)
from an asset that is NOT the synthetic seed asset the other tests use; HANDS_SYNTHETIC_MANO is removed for the
duration, so any fall-back to the stand-in fails loudly instead of passing.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import hands_amd
from fake_mano import realistic_mano_asset, write_fake_mano_pkl
from hands_amd import _lib
from hands_amd._lib import check, ptr
from hands_amd.checkpoint import load_reference_checkpoint
from hands_amd.mano import build_mano_asset, load_mano_pkl
from hands_amd.packing import pack_mano
from hands_amd.weights import synthetic_inputs
from oracle import hands_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture()
def mano_dir(tmp_path, monkeypatch):
    written = {}
    for is_rhand in (True, False):
        a = realistic_mano_asset(is_rhand)
        write_fake_mano_pkl(str(tmp_path / ("MANO_RIGHT.pkl" if is_rhand else "MANO_LEFT.pkl")), a)
        written[is_rhand] = a
    monkeypatch.setenv("MANO_DIR", str(tmp_path))
    monkeypatch.delenv("HANDS_SYNTHETIC_MANO", raising=False)
    return tmp_path, written


def _mano_inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    rot = O.rotation_6d_to_matrix(torch.randn(B * 16, 6, generator=g)).view(B, 16, 3, 3)
    betas = torch.randn(B, 10, generator=g)
    cam = torch.tensor([1.0, 0, 0]) + 0.1 * torch.randn(B, 3, generator=g)
    K = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]])[None].repeat(B, 1, 1)
    return rot, betas, cam, K


def test_pkl_loaded_asset_through_the_mano_kernel(mano_dir):
    """MANO_{RIGHT,LEFT}.pkl -> load_mano_pkl -> hands_pack_mano_f32 -> hands_mano_heads_f32 (both hands, one
    launch) vs the oracle's MANOHead on the same loaded asset: <= 1e-6 m, and vs its fp64 run."""
    tmp, written = mano_dir
    assets = [load_mano_pkl(str(tmp / "MANO_RIGHT.pkl"), True), load_mano_pkl(str(tmp / "MANO_LEFT.pkl"), False)]
    for a, is_rhand in zip(assets, (True, False)):
        assert np.array_equal(a.J_regressor, written[is_rhand].J_regressor) and a.faces.dtype == np.int64
        assert np.array_equal(a.faces, written[is_rhand].faces)                      # bit-exact face indices
    B = 21
    rot, betas, cam, K = _mano_inputs(2 * B, 3)
    K = K[:B]
    L = _lib.lib()
    mps = [pack_mano(a, DEV) for a in assets]
    keep = [t.to(DEV).contiguous() for t in (rot, betas, cam, K)]
    sides = (_lib.ManoSide * 2)()
    outs = []
    names = ("vertices", "joints3d", "v3d.cam", "j3d.cam", "j2d.norm", "cam_t")
    shapes = ((778, 3), (21, 3), (778, 3), (21, 3), (21, 2), (3,))
    for s_, mp in enumerate(mps):
        consts = _lib.ManoConsts(ptr(mp["pose_mean"]), ptr(mp["J_template"]), ptr(mp["J_shapedirs"]),
                                 ptr(mp["lbs_weights"]), ptr(mp["tip_ids"]))
        o = {n: torch.full((B,) + sh, float("nan"), device=DEV) for n, sh in zip(names, shapes)}
        mo = _lib.ManoOut(*[ptr(o[n]) for n in names])
        sides[s_] = _lib.ManoSide(consts, ptr(mp["blend"].w), ptr(mp["blend"].bias), ptr(keep[0], s_ * B * 144),
                                  ptr(keep[1], s_ * B * 10), ptr(keep[2], s_ * B * 3), mo)
        outs.append(o)
    check(L.hands_mano_heads_f32(sides, 2, ptr(keep[3]), 10, 224.0, 0.1, B, 0, torch.cuda.current_stream().cuda_stream),
          "mano_heads")
    torch.cuda.synchronize()
    for s_, asset in enumerate(assets):
        sl = slice(s_ * B, (s_ + 1) * B)
        ref = O.mano_head(rot[sl], betas[sl], cam[sl], K, asset, 224, "")
        aa = O.matrix_to_axis_angle(rot[sl].double().view(-1, 3, 3)).view(-1, 48)
        v64, j64 = O.mano_lbs(betas[sl], aa[:, :3], aa[:, 3:], asset, dtype=torch.float64)
        g = {k: v.cpu() for k, v in outs[s_].items()}
        assert all(torch.isfinite(v).all() for v in g.values())
        assert (g["vertices"] - ref["vertices"]).abs().max().item() < 1e-6
        assert (g["joints3d"] - ref["joints3d"]).abs().max().item() < 1e-6
        assert (g["vertices"].double() - v64).abs().max().item() < 1e-6
        assert (g["joints3d"].double() - j64).abs().max().item() < 1e-6
        assert torch.allclose(g["v3d.cam"], ref["v3d.cam"], rtol=2e-6, atol=2e-6)
        assert (g["j2d.norm"] - ref["j2d.norm"]).abs().max().item() < 1e-5
    # the two sides really used their own file
    assert (outs[0]["vertices"][0] - outs[1]["vertices"][0]).abs().max().item() > 1e-3


def test_reference_checkpoint_and_pkl_assets_through_the_forward(mano_dir, tmp_path):
    """The INTEGRATION.md swap end to end: ``HandsLight()`` reads $MANO_DIR like the reference's constructor,
    a Lightning-format checkpoint (``model.``-prefixed, plus the wrapper's own ``mano_{r,l}.*`` buffers, plus
    optimizer junk) is loaded from disk, the model goes to the GPU and its forward equals the oracle's on the
    same loaded state_dict and the same loaded assets (<= 1e-6 m vertices, <= 1e-3 mm MPJPE)."""
    tmp, written = mano_dir
    donor = hands_amd.apply_recipe(hands_amd.HandsLight())
    assert np.array_equal(donor.mano_r.mano.asset().v_template, written[True].v_template)     # came from the pickle
    sd = {"model." + k: v.detach().clone() for k, v in donor.state_dict().items()}
    for side in ("mano_r", "mano_l"):                                  # generic/wrapper.py:36-39 wrapper-level layers
        for k, v in getattr(donor, side).state_dict().items():
            sd[f"{side}.{k}"] = v.detach().clone()
    ckpt = {"state_dict": sd, "epoch": 12, "global_step": 3456, "optimizer_states": [{"state": {}}],
            "pytorch-lightning_version": "1.5.10"}
    path = str(tmp_path / "last.ckpt")
    torch.save(ckpt, path)

    model = hands_amd.HandsLight()                                      # fresh, initial weights, assets from $MANO_DIR
    rep = load_reference_checkpoint(model, path)
    assert rep.missing_keys == [] and rep.unexpected_keys == []
    for k, v in donor.state_dict().items():
        assert torch.equal(model.state_dict()[k], v), k
    model = model.to(DEV).eval()
    inputs, meta = synthetic_inputs(3, 5)
    meta["is_flipped"] = torch.tensor([0, 1, 0])
    own_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref = O.hands_light_forward(own_sd, build_mano_asset(True), build_mano_asset(False), inputs, meta)
    out = model({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in meta.items()})
    torch.cuda.synchronize()
    assert len(out) == 22
    for hn in "rl":
        verr = (out[f"mano.vertices.{hn}"].cpu() - ref[f"mano.vertices.{hn}"]).abs().max().item()
        mp = O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"].cpu(), ref[f"mano.joints3d.{hn}"])
        assert verr < 1e-6 and mp < 1e-3, (hn, verr, mp)
        assert torch.allclose(out[f"mano.v3d.cam.{hn}"].cpu(), ref[f"mano.v3d.cam.{hn}"], rtol=2e-5, atol=2e-5)
        assert torch.allclose(out[f"grasp.{hn}"].cpu(), ref[f"grasp.{hn}"], rtol=1e-4, atol=1e-4)
    # and the mesh really is the pickle's: the same weights on the synthetic asset give another mesh
    syn = O.hands_light_forward(own_sd, hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False), inputs, meta)
    assert (syn["mano.vertices.r"] - ref["mano.vertices.r"]).abs().max().item() > 1e-3
    assert np.array_equal(model.mano_r.faces, written[True].faces) and np.array_equal(model.mano_l.faces, written[False].faces)
