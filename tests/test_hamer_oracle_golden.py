"""Pins oracle/hamer_oracle.py against fixtures produced by the IMPORTED REFERENCE HAMER
(tests/golden/make_golden_hamer.py)."""
import json
import os

import numpy as np
import pytest
import torch

import hands_amd
from hands_amd.mano import synthetic_mano_asset
from hands_amd.weights import synthetic_inputs
from oracle import hamer_oracle as H
from oracle import hands_oracle as O

torch.set_num_threads(min(8, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def hamer_model():
    m = hands_amd.apply_recipe(hands_amd.HAMER())
    m.eval()
    return m


@pytest.fixture(scope="module")
def hamer_sd(hamer_model):
    return {k: v.detach().cpu() for k, v in hamer_model.state_dict().items()}


def test_hamer_state_dict_keys_match_reference(golden_dir, hamer_model):
    ref = json.load(open(os.path.join(golden_dir, "hamer_state_dict_keys.json")))
    mine = {k: list(v.shape) for k, v in hamer_model.state_dict().items() if ".mano." not in k}
    assert mine == ref and len(ref) == 515


def _probe_check(t_bnc, d, prefix, tol):
    t = t_bnc.detach().float().transpose(1, 2)
    flat = t.reshape(-1)
    np.testing.assert_allclose(flat[torch.from_numpy(d[prefix + "/idx"])].numpy(), d[prefix + "/val"], rtol=tol, atol=tol)
    np.testing.assert_allclose(t.abs().amax(dim=(0, 2)).numpy(), d[prefix + "/absmax_c"], rtol=tol, atol=tol)


@pytest.mark.parametrize("seed", [0, 1])
def test_hamer_forward_matches_reference(golden_dir, hamer_sd, seed):
    d = np.load(os.path.join(golden_dir, f"hamer_light_bz2_seed{seed}.npz"))
    inputs, meta_info = synthetic_inputs(2, seed)
    out, inter = H.hamer_forward(hamer_sd, synthetic_mano_asset(True), synthetic_mano_asset(False), inputs,
                                 meta_info, return_intermediates=True)
    np.testing.assert_allclose(inter["kpe"][:2].numpy(), d["kpe_r"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(inter["kpe"][2:].numpy(), d["kpe_l"], rtol=1e-5, atol=1e-5)
    _probe_check(inter["blocks"][0], d, "probe/block0", 2e-5)
    _probe_check(inter["blocks"][15], d, "probe/block15", 1e-4)
    _probe_check(inter["blocks"][31], d, "probe/block31", 2e-4)
    np.testing.assert_allclose(inter["token_out"].numpy(), d["token_out"], rtol=2e-4, atol=2e-4)
    keys = sorted(k[4:] for k in d.files if k.startswith("out/"))
    assert sorted(out.keys()) == keys and len(keys) == 22
    for k in keys:
        ref, got = d["out/" + k], out[k].numpy()
        assert got.shape == ref.shape, k
        tol = 1e-3 if k.startswith("grasp") else 5e-5
        np.testing.assert_allclose(got, ref, rtol=tol, atol=tol, err_msg=k)
    for hn in "rl":
        assert np.abs(out[f"mano.vertices.{hn}"].numpy() - d[f"out/mano.vertices.{hn}"]).max() < 1e-5
        assert O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"], torch.from_numpy(d[f"out/mano.joints3d.{hn}"])) < 1e-2


def test_rot6d_columns_is_transpose_of_rows():
    g = torch.Generator().manual_seed(0)
    d6 = torch.randn(32, 6, generator=g)
    assert torch.allclose(H.rot6d_to_rotmat_columns(d6), O.rotation_6d_to_matrix(d6).transpose(1, 2), atol=1e-7)


@pytest.mark.parametrize("name", ["hamer_light", "handoccnet_light"])
def test_no_kpe_no_grasp_switches_match_reference(golden_dir, name):
    """pos_enc=None (no KPE term anywhere) and use_grasp_loss=False for HAMER (hamer_light/model.py:53-72,91-104,136-143) and
    HandOccNet (handoccnet_light/model.py:74-89,113-120): the host mirror drops the `kpe.*` / `grasp_classifier.*` tensors like
    the reference and the oracle reproduces the reference's 20 outputs (tests/golden/make_golden_switches_other.py)."""
    import hands_amd
    from oracle import handoccnet_oracle as HO
    d = np.load(os.path.join(golden_dir, f"{name}_switch_nokpe.npz"))
    meta = json.loads(str(d["meta"]))
    base = hands_amd.HAMER_DEFAULT_ARGS if name == "hamer_light" else hands_amd.HANDOCC_DEFAULT_ARGS
    args = type(base)(dict(base, **meta["config"]))
    model = hands_amd.apply_recipe(hands_amd.HAMER(args) if name == "hamer_light" else hands_amd.HandOccNet(args=args)).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert sum(1 for k in sd if ".mano." not in k) == meta["n_state_dict"]
    assert not any(k.startswith(("kpe.", "grasp_classifier.")) for k in sd)
    inputs, meta_info = synthetic_inputs(meta["bz"], meta["seed"])
    fwd = H.hamer_forward if name == "hamer_light" else HO.handoccnet_forward
    out = fwd(sd, synthetic_mano_asset(True), synthetic_mano_asset(False), inputs, meta_info, pos_enc=None, use_grasp_loss=False)
    keys = sorted(k[4:] for k in d.files if k.startswith("out/"))
    assert sorted(out.keys()) == keys and len(keys) == 20
    for k in keys:
        np.testing.assert_allclose(out[k].numpy(), d["out/" + k], rtol=5e-5, atol=5e-5, err_msg=k)
    for hn in "rl":
        assert np.abs(out[f"mano.vertices.{hn}"].numpy() - d[f"out/mano.vertices.{hn}"]).max() < 1e-5
