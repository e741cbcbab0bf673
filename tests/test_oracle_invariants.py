"""Known-answer / invariant tests that pin the UNPINNED parts of the oracle (a5: pytorch3d 6D, a9:
smplx MANO LBS) -- SURVEY.md section 8c (i)-(v) -- plus independent scipy cross-checks."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation as SciRot

from hands_amd.mano import PARENTS, TIP_IDS, synthetic_mano_asset
from oracle import hands_oracle as O


@pytest.fixture()
def asset():
    return synthetic_mano_asset(True)


def _t(a):
    return torch.from_numpy(np.asarray(a)).double()


def test_zero_pose_gives_shaped_template(asset):
    asset.hands_mean[:] = 0
    beta = torch.randn(3, 10, dtype=torch.float64)
    v, j = O.mano_lbs(beta, torch.zeros(3, 3).double(), torch.zeros(3, 45).double(), asset, dtype=torch.float64)
    v_shaped = _t(asset.v_template) + torch.einsum("bl,mkl->bmk", beta, _t(asset.shapedirs))
    # skinning weights are fp32 and sum to 1 only to ~6e-8, so T_v = (sum_j w_vj) I is not exactly I
    assert (v - v_shaped).abs().max() < 5e-8                                   # (i)
    assert (j[:, :16] - torch.einsum("ji,bik->bjk", _t(asset.J_regressor), v_shaped)).abs().max() < 1e-9


def test_global_rotation_is_rigid_about_the_wrist(asset):
    asset.hands_mean[:] = 0
    beta = torch.randn(2, 10, dtype=torch.float64)
    rv = torch.tensor([[0.3, -1.1, 0.7], [2.0, 0.1, -0.4]], dtype=torch.float64)
    v, j = O.mano_lbs(beta, rv, torch.zeros(2, 45).double(), asset, dtype=torch.float64)
    R = _t(SciRot.from_rotvec(rv.numpy()).as_matrix())
    v_shaped = _t(asset.v_template) + torch.einsum("bl,mkl->bmk", beta, _t(asset.shapedirs))
    J0 = torch.einsum("i,bik->bk", _t(asset.J_regressor[0]), v_shaped)
    exp = torch.einsum("bij,bvj->bvi", R, v_shaped - J0[:, None]) + J0[:, None]
    assert (v - exp).abs().max() < 1e-7                                        # (ii) (1e-8 inside the norm)


def test_one_hot_skinning_follows_each_joint(asset):
    asset.hands_mean[:] = 0
    asset.posedirs[:] = 0
    w = np.zeros_like(asset.lbs_weights)
    owner = np.arange(778) % 16
    w[np.arange(778), owner] = 1
    asset.lbs_weights[:] = w
    g = torch.Generator().manual_seed(0)
    aa = 0.5 * torch.randn(1, 48, generator=g, dtype=torch.float64)
    beta = torch.zeros(1, 10, dtype=torch.float64)
    v, j = O.mano_lbs(beta, aa[:, :3], aa[:, 3:], asset, dtype=torch.float64)
    # forward kinematics by hand
    R = _t(SciRot.from_rotvec(aa.view(16, 3).numpy()).as_matrix())
    J = _t(asset.J_regressor) @ _t(asset.v_template)
    G = [None] * 16
    for i in range(16):
        T = torch.eye(4, dtype=torch.float64)
        T[:3, :3] = R[i]
        T[:3, 3] = J[i] - (J[PARENTS[i]] if i else 0)
        G[i] = T if i == 0 else G[PARENTS[i]] @ T
    for vid in (0, 5, 100, 777):
        o = owner[vid]
        exp = G[o][:3, :3] @ (_t(asset.v_template[vid]) - J[o]) + G[o][:3, 3]
        assert (v[0, vid] - exp).abs().max() < 1e-7                            # (iii)
    assert (j[0, :16] - torch.stack([g_[:3, 3] for g_ in G])).abs().max() < 1e-7


def test_fingertips_are_the_listed_vertices(asset):
    g = torch.Generator().manual_seed(1)
    v, j = O.mano_lbs(torch.randn(2, 10, generator=g), torch.randn(2, 3, generator=g),
                      0.3 * torch.randn(2, 45, generator=g), asset)
    assert j.shape == (2, 21, 3) and torch.equal(j[:, 16:], v[:, list(TIP_IDS)])   # (iv)


def test_fp32_lbs_close_to_fp64(asset):
    g = torch.Generator().manual_seed(2)
    b, go, hp = torch.randn(8, 10, generator=g), torch.randn(8, 3, generator=g), 0.5 * torch.randn(8, 45, generator=g)
    v32, j32 = O.mano_lbs(b, go, hp, asset)
    v64, j64 = O.mano_lbs(b, go, hp, asset, dtype=torch.float64)
    assert (v32.double() - v64).abs().max() < 5e-7 and (j32.double() - j64).abs().max() < 5e-7   # (v)


def test_rodrigues_matches_scipy():
    g = torch.Generator().manual_seed(3)
    rv = torch.randn(64, 3, generator=g, dtype=torch.float64)
    assert np.abs(O.batch_rodrigues(rv).numpy() - SciRot.from_rotvec(rv.numpy()).as_matrix()).max() < 1e-7


def test_matrix_to_axis_angle_matches_scipy():
    g = torch.Generator().manual_seed(4)
    R = O.rotation_6d_to_matrix(torch.randn(256, 6, generator=g, dtype=torch.float64))
    # the reference's conversion may return an angle in (pi, 2pi) (negative real quaternion part);
    # compare as rotations, not as vectors
    aa = O.matrix_to_axis_angle(R).numpy()
    assert np.abs(SciRot.from_rotvec(aa).as_matrix() - R.numpy()).max() < 1e-9
    canon = np.linalg.norm(aa, axis=1) <= np.pi
    assert canon.any() and np.abs(aa[canon] - SciRot.from_matrix(R.numpy()).as_rotvec()[canon]).max() < 1e-9
    rv = torch.randn(32, 3, generator=g, dtype=torch.float64)
    assert np.abs(O.axis_angle_to_matrix(rv).numpy() - SciRot.from_rotvec(rv.numpy()).as_matrix()).max() < 1e-12


def test_rotation_6d_properties():
    g = torch.Generator().manual_seed(5)
    R = O.rotation_6d_to_matrix(torch.randn(128, 6, generator=g, dtype=torch.float64))
    eye = torch.eye(3, dtype=torch.float64)
    assert (R @ R.transpose(1, 2) - eye).abs().max() < 1e-12 and (torch.linalg.det(R) - 1).abs().max() < 1e-12
    ident = O.rotation_6d_to_matrix(torch.tensor([[1.0, 0, 0, 0, 1.0, 0]]))
    assert torch.equal(ident[0], torch.eye(3))
    assert torch.equal(O.matrix_to_rotation_6d(torch.eye(3)[None]), torch.tensor([[1.0, 0, 0, 0, 1.0, 0]]))
    assert (O.rotation_6d_to_matrix(O.matrix_to_rotation_6d(R)) - R).abs().max() < 1e-12


def test_mpjpe_metric():
    a = torch.randn(4, 21, 3)
    assert O.mpjpe_ra_mm(a, a + torch.tensor([1.0, 2.0, 3.0])) < 1e-3        # root-aligned: translation-free
    b = a.clone()
    b[:, 1:] += torch.tensor([0.001, 0.0, 0.0])
    assert abs(O.mpjpe_ra_mm(a, b) - 1.0 * 20 / 21) < 1e-3


def test_rodrigues_matches_reference_in_repo_twin(golden_dir):
    """a9's batch_rodrigues is smplx's (absent); the reference vendors the same map in quaternion form
    (common/rot.py:316-327).  The restatement must agree with that real code to fp32 rounding."""
    import os
    d = np.load(os.path.join(golden_dir, "rodrigues_twin.npz"))
    rv = torch.from_numpy(d["rotvec"])
    got64 = O.batch_rodrigues(rv.double()).numpy()
    assert np.abs(got64 - d["R64"]).max() < 1e-7          # 1e-8 guard: the twin normalises (r+1e-8), we ||r+1e-8||
    got32 = O.batch_rodrigues(rv).numpy()
    assert np.abs(got32 - d["R32"]).max() < 2e-6
