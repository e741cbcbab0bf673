"""Kernels behind the non-default HandsLight encodings, each against the oracle's statement of the reference code
(oracle/hands_oracle.py: dense_pos_enc / cam_conv_pos_enc / depth_head / euler_angles_to_matrix_xyz) or against ATen's own
F.interpolate, on geometries the end-to-end fixtures do not reach (img_res_ds != img_res: both interpolations active)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hands_amd import _lib
from hands_amd._lib import check, ptr
from hands_amd.weights import synthetic_dense_inputs
from oracle import hands_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("kind,R,Ho", [("dense", 224, 224), ("dense", 112, 112), ("dense_latent", 224, 7), ("dense_latent", 96, 7),
                                        ("cam_conv", 224, 7), ("cam_conv", 160, 5)])
def test_dense_posenc_vs_oracle(kind, R, Ho):
    L = _lib.lib()
    bz, nf = 3, 4
    d = synthetic_dense_inputs(bz, 5, "cam_conv" if kind == "cam_conv" else "dense")
    ang, msk = d["r_dense_angle"], d["r_dense_mask"]
    ref = O.cam_conv_pos_enc(ang, msk, R) if kind == "cam_conv" else O.dense_pos_enc(ang, msk, nf, R)
    if Ho != R:
        ref = F.interpolate(ref, size=(Ho, Ho), mode="bilinear", align_corners=True)
    Ce, Ca = ref.shape[1], ang.shape[1]
    img = torch.randn(bz, 3, Ho, Ho) if kind == "dense" else None
    ld, c_off = ((3 + Ce + 15) // 16 * 16, 3) if img is not None else (Ce + 8, 4)
    out = torch.full((bz, Ho, Ho, ld), -7.0, device=DEV)
    ang_d, msk_d, img_d = ang.to(DEV), msk.to(DEV), (img.to(DEV) if img is not None else None)     # (kept alive across the launch)
    check(L.hands_dense_posenc_f32(ptr(ang_d), ptr(msk_d), ptr(img_d) if img is not None else None, ptr(out),
                                   bz, Ca, ang.shape[2], ang.shape[3], 0 if kind == "cam_conv" else nf, R, Ho, Ho, ld, c_off,
                                   _stream()), "dense_posenc")
    got = out.cpu()
    enc = got[..., c_off:c_off + Ce].permute(0, 3, 1, 2)
    scale = max(1.0, float(ref.abs().max()))
    assert (enc - ref).abs().max().item() < 2e-6 * scale        # sinf / cosf and the interpolation weights: a few ulp
    if img is not None:
        assert torch.equal(got[..., :3].permute(0, 3, 1, 2), img) and torch.all(got[..., 3 + Ce:] == 0)
    else:
        assert torch.all(got[..., :c_off] == -7.0) and torch.all(got[..., c_off + Ce:] == -7.0)      # untouched outside its channels


def test_concat_nhwc_vs_torch_cat():
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    B, Bg, HW, Ca, Cb = 6, 3, 49, 40, 6
    a, add, ex = torch.randn(B, HW, Ca + 8, generator=g), torch.randn(Bg, HW, Ca, generator=g), torch.randn(B, HW, Cb, generator=g)
    ld = 64
    out = torch.empty(B, HW, ld, device=DEV)
    a_d, add_d, ex_d = a.to(DEV), add.to(DEV), ex.to(DEV)
    check(L.hands_concat_nhwc_f32(ptr(a_d), Ca + 8, Ca, ptr(add_d), Ca, ptr(ex_d), HW * Cb, Cb, ptr(out), ld,
                                  B, Bg, HW, _stream()), "concat")
    ref = torch.cat([a[..., :Ca] + add.repeat(2, 1, 1), ex, torch.zeros(B, HW, ld - Ca - Cb)], -1)
    assert torch.equal(out.cpu(), ref)
    grid = torch.randn(HW, 2, generator=g)                       # one map for every sample (the depth head's grid), no addend
    grid_d = grid.to(DEV)
    check(L.hands_concat_nhwc_f32(ptr(a_d), Ca + 8, Ca, None, 0, ptr(grid_d), 0, 2, ptr(out), ld, B, B, HW, _stream()), "concat")
    ref = torch.cat([a[..., :Ca], grid[None].repeat(B, 1, 1), torch.zeros(B, HW, ld - Ca - 2)], -1)
    assert torch.equal(out.cpu(), ref)
    assert L.hands_concat_nhwc_f32(ptr(a_d), Ca, Ca, None, 0, None, 0, 2, ptr(out), ld, B, B, HW, _stream()) != 0   # Cb without a map


@pytest.mark.parametrize("h,k,C", [(7, 4, 256), (28, 4, 128), (12, 2, 32), (5, 3, 8)])
def test_upsample_bilinear_align_corners_vs_aten(h, k, C):
    L = _lib.lib()
    x = torch.randn(3, C, h, h + 1, generator=torch.Generator().manual_seed(h))
    ref = F.interpolate(x, scale_factor=k, mode="bilinear", align_corners=True)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.empty(3, h * k, (h + 1) * k, C, device=DEV)
    check(L.hands_upsample_bilinear_ac_f32(ptr(xn), ptr(out), 3, h, h + 1, h * k, (h + 1) * k, C, _stream()), "upsample_ac")
    assert (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-6
    assert L.hands_upsample_bilinear_ac_f32(ptr(xn), ptr(out), 3, h, h + 1, h * k, (h + 1) * k, C + 1, _stream()) != 0


def test_rotation_corrections_vs_oracle():
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    bz = 5
    rot = O.axis_angle_to_matrix(torch.randn(2 * bz, 16, 3, generator=g))            # (2 bz, 16, 3, 3)
    fix = synthetic_dense_inputs(2 * bz, 1, "pcl")["r_rot"]
    got, fix_d = rot.clone().to(DEV), fix.to(DEV)
    check(L.hands_rot_leftmul_f32(ptr(got), ptr(fix_d), 2 * bz, _stream()), "rot_leftmul")
    ref = rot.clone()
    ref[:, 0] = torch.bmm(fix, rot[:, 0])
    assert (got.cpu() - ref).abs().max().item() < 1e-6 and torch.equal(got.cpu()[:, 1:], rot[:, 1:])
    center = 0.4 * torch.randn(2 * bz, 2, generator=g)
    e = O.euler_angles_to_matrix_xyz(torch.cat([-center, torch.zeros(2 * bz, 1)], -1))
    want = rot.clone()
    want[:, 0] = torch.matmul(e, rot[:, 0])
    for flips in ([0] * bz, [0, 0, 1, 0, 0]):
        sw, un, c_d, f_d = rot.clone().to(DEV), rot.clone().to(DEV), center.to(DEV), torch.tensor(flips, device=DEV)
        check(L.hands_perspective_correction_f32(ptr(sw), ptr(un), ptr(c_d), ptr(f_d), bz, _stream()), "persp")
        assert (sw.cpu() - want).abs().max().item() < 1e-6
        # in place in the reference: the grasp head's copy follows only when no sample of the batch is flipped (model.py:341, 370-376)
        assert torch.equal(un.cpu(), sw.cpu() if not any(flips) else rot)
    # euler_angles_to_matrix('XYZ') itself against scipy's extrinsic-free statement: R = Rx Ry Rz
    from scipy.spatial.transform import Rotation
    ang = torch.randn(7, 3, generator=g)
    assert np.abs(O.euler_angles_to_matrix_xyz(ang).numpy() - Rotation.from_euler("XYZ", ang.numpy()).as_matrix()).max() < 1e-6
