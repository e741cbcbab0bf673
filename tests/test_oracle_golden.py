"""Pins oracle/ against fixtures produced by the IMPORTED REFERENCE (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from hands_amd.mano import synthetic_mano_asset
from hands_amd.weights import synthetic_inputs
from oracle import hands_oracle as O
from switch_cases import SWITCH_CASES

torch.set_num_threads(min(8, os.cpu_count() or 1))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _probe_check(t, d, prefix, rtol=2e-5, atol=2e-5):
    t = t.detach().float()
    red = tuple(i for i in range(t.ndim) if i != 1)
    np.testing.assert_allclose(t.mean(dim=red).numpy(), d[prefix + "/mean_c"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(t.abs().amax(dim=red).numpy(), d[prefix + "/absmax_c"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(t.reshape(-1)[torch.from_numpy(d[prefix + "/idx"])].numpy(), d[prefix + "/val"],
                               rtol=rtol, atol=atol)
    assert list(t.shape) == list(d[prefix + "/shape"])


@pytest.mark.parametrize("bz,seed", [(2, 0), (2, 1), (2, 2), (1, 0)])
def test_forward_matches_reference(golden_dir, recipe_sd, bz, seed):
    """bz=2 fixtures + the bz=1 one: BASELINE configs[0] (single crop, bs=1, CPU forward -- plumbing)."""
    d = _load(golden_dir, f"hands_light_bz{bz}_seed{seed}.npz")
    meta = json.loads(str(d["meta"]))
    assert meta["bz"] == bz and meta["seed"] == seed
    inputs, meta_info = synthetic_inputs(bz, seed)
    meta_info["is_flipped"] = torch.from_numpy(d["is_flipped"])
    ar, al = synthetic_mano_asset(True), synthetic_mano_asset(False)
    out, inter = O.hands_light_forward(recipe_sd, ar, al, inputs, meta_info, return_intermediates=True)
    keys = sorted(k[4:] for k in d.files if k.startswith("out/"))
    assert sorted(out.keys()) == keys and len(keys) == 22
    # intermediates (real reference code end to end)
    _probe_check(inter["features"], d, "probe/features")
    _probe_check(inter["r_feat"], d, "probe/hand_feat_r")
    _probe_check(inter["l_feat"], d, "probe/hand_feat_l")
    np.testing.assert_allclose(inter["r_vec"].numpy(), d["feature_conv_r"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(inter["l_vec"].numpy(), d["feature_conv_l"], rtol=2e-5, atol=2e-5)
    for hn in "rl":
        h = inter["hmr_" + hn]
        np.testing.assert_allclose(h["pose_6d"].numpy(), d[f"hmr_{hn}/pose_6d"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(h["shape"].numpy(), d[f"hmr_{hn}/shape"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(h["cam_t.wp"].numpy(), d[f"hmr_{hn}/cam_t.wp"], rtol=1e-5, atol=1e-5)
    for k in keys:
        ref = d["out/" + k]
        got = out[k].numpy()
        assert got.shape == ref.shape, k
        if k.startswith("grasp"):
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4, err_msg=k)
        elif ".cam." in k or k.startswith("mano.cam_t."):
            # camera-space values sit at metres-from-camera: compare relatively (fp32 ulp at 9 m ~ 1e-6)
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5, err_msg=k)
        else:
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5, err_msg=k)
    # the headline tolerance: 1e-3 mm (1e-6 m) on canonical vertices / joints
    for hn in "rl":
        assert out[f"mano.vertices.{hn}"].shape == (bz, 778, 3) and out[f"mano.joints3d.{hn}"].shape == (bz, 21, 3)
        assert np.abs(out[f"mano.vertices.{hn}"].numpy() - d[f"out/mano.vertices.{hn}"]).max() < 1e-6
        assert O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"], torch.from_numpy(d[f"out/mano.joints3d.{hn}"])) < 1e-3


def test_trunk_stage_probes(golden_dir, recipe_sd):
    d = _load(golden_dir, "hands_light_bz2_seed0.npz")
    inputs, _ = synthetic_inputs(2, 0)
    with torch.no_grad():
        _, stages = O.resnet50_trunk(inputs["img"], recipe_sd, "backbone", return_stages=True)
    for li in (1, 2, 3):
        _probe_check(stages[li], d, f"probe/layer{li}")


def test_rot_conversions_match_reference(golden_dir):
    d = _load(golden_dir, "rot_conversions.npz")
    R = torch.from_numpy(d["R"])
    np.testing.assert_allclose(O.matrix_to_quaternion(R).numpy(), d["quat"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(O.matrix_to_axis_angle(R).numpy(), d["aa"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(O.axis_angle_to_matrix(torch.from_numpy(d["aa_in"])).numpy(), d["R_from_aa"],
                               rtol=0, atol=1e-6)


def test_rot6d_is_transpose_of_in_repo_twin(golden_dir):
    """a5 cross-check: hamer's rot6d_to_rotmat (geometry.py:47-62) takes a1 = x[:3], a2 = x[3:] like
    pytorch3d but stacks b1,b2,b3 as COLUMNS, so it must equal the transpose of the rows convention."""
    d = _load(golden_dir, "rot6d_twin.npz")
    mine = O.rotation_6d_to_matrix(torch.from_numpy(d["d6"]))
    np.testing.assert_allclose(mine.transpose(1, 2).numpy(), d["hamer_rotmat"], rtol=0, atol=1e-6)


def test_camera_projection_match_reference(golden_dir):
    d = _load(golden_dir, "camera_projection.npz")
    cam, K = torch.from_numpy(d["cam"]), torch.from_numpy(d["K"])
    f = (K[:, 0, 0] + K[:, 1, 1]) / 2.0
    cam_t = O.weak_perspective_to_perspective(cam, f, 224, 0.1)
    np.testing.assert_array_equal(cam_t.numpy(), d["cam_t"])
    j2d = O.normalize_kp2d(O.project2d_batch(K, torch.from_numpy(d["pts"])), 224)
    np.testing.assert_allclose(j2d.numpy(), d["j2d_norm"], rtol=0, atol=1e-6)


def test_kpe_matches_reference(golden_dir):
    d = _load(golden_dir, "kpe.npz")
    np.testing.assert_array_equal(O.pos_enc(torch.from_numpy(d["center_angle"])).numpy(), d["center_enc"])
    np.testing.assert_array_equal(O.pos_enc(torch.from_numpy(d["corner_angle"])).numpy(), d["corner_enc"])


def test_state_dict_keys_match_reference(golden_dir, recipe_model):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    mine = {k: list(v.shape) for k, v in recipe_model.state_dict().items() if ".mano." not in k}
    assert mine == ref and len(ref) == 681


@pytest.mark.parametrize("name", SWITCH_CASES)
def test_switch_configurations_match_reference(golden_dir, name):
    """Non-default HandsLight switches (model.py:40-47,60-86,127,199-232,316-318,401-411): the host mirror builds the
    reference's parameter tree for the configuration (names + shapes from the reference's own state_dict) and the oracle
    reproduces the reference's outputs on the seeded inputs."""
    import hands_amd
    from switch_cases import load_case, oracle_kwargs
    d, cfg, args, inputs, meta_info = load_case(golden_dir, name)
    model = hands_amd.apply_recipe(hands_amd.HandsLight(args=args)).eval()
    base = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    diff = json.load(open(os.path.join(golden_dir, "switch_state_dict_keys.json")))[name]
    want = {k: v for k, v in base.items() if k not in diff["absent"]}
    want.update(diff["changed"])
    mine = {k: list(v.shape) for k, v in model.state_dict().items() if ".mano." not in k}
    assert mine == want and len(mine) == diff["n_keys"]
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = O.hands_light_forward(sd, synthetic_mano_asset(True), synthetic_mano_asset(False), inputs, meta_info,
                                **oracle_kwargs(cfg))
    keys = sorted(k[4:] for k in d.files if k.startswith("out/"))
    assert sorted(out.keys()) == keys and len(keys) == ((22 if cfg.get("use_grasp_loss", True) else 20) + (4 if cfg.get("regress_center_corner") else 0)
                                                     + (2 if cfg.get("use_depth_loss") else 0))
    for k in keys:
        mlp = k.startswith(("grasp", "center.", "corner."))
        tol = 1e-4 if mlp else 2e-5
        np.testing.assert_allclose(out[k].numpy(), d["out/" + k], rtol=tol if (".cam." in k or "cam_t" in k or mlp) else 0,
                                   atol=tol, err_msg=k)
    for hn in "rl":
        assert np.abs(out[f"mano.vertices.{hn}"].numpy() - d[f"out/mano.vertices.{hn}"]).max() < 1e-6
        assert O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"], torch.from_numpy(d[f"out/mano.joints3d.{hn}"])) < 1e-3
