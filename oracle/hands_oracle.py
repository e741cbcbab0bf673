"""CPU oracle for the hands_light forward path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``hands_amd``) never does and fails loudly when its HIP library is missing.

It is a plain torch-CPU restatement of the reference's algorithm, function by function, each
citing the reference file:line it follows (paths relative to the reference checkout).  It works
from a flat ``state_dict`` with the reference's key names, so the same parameters drive the
imported reference (when the golden fixtures were generated), this oracle, and the HIP path.

Pinning status
--------------
* a1-a4, a6-a8, a10 (trunk, KPE, feature_conv, HMR, grasp, R->axis-angle, camera/projection):
  pinned -- ``tests/golden/*.npz`` were produced by the *imported reference* (shim harness in
  ``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py`` checks this file against
  them.
* a5 (``pytorch3d`` 6D<->matrix) and a9 (``smplx.MANO`` / ``smplx.lbs``): **parity unpinned** --
  both are third-party, un-vendored, version-unpinned dependencies that are absent from the
  reference checkout, and the reference holds no tests or vectors for them.  They are restated from
  their published definitions and pinned only by known-answer/invariant tests
  (``tests/test_oracle_invariants.py``) and, for 6D, by the in-repo transposed twins
  (src/models/hamer_light/geometry.py:47-62).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)
TIP_IDS = (744, 320, 443, 554, 671)
RESNET50_LAYERS = (3, 4, 6, 3)


# ------------------------------------------------------------------------------------------------
# a1  ResNet-50 v1.5 trunk (src/nets/backbone/resnet.py:264-280, Bottleneck :134-154)
# ------------------------------------------------------------------------------------------------
def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=False, eps=eps)


def bottleneck(x, sd, p, stride):
    """resnet.py:134-154 -- stride sits on the 3x3 (v1.5)."""
    identity = x
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1), sd, p + ".bn2"))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    if (p + ".downsample.0.weight") in sd:
        identity = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
    return F.relu(out + identity)


def resnet50_trunk(x, sd, prefix, return_stages=False):
    """resnet.py:264-280: conv7x7s2 -> BN -> ReLU -> maxpool3x3s2 -> layer1..4, no avgpool/fc."""
    x = F.conv2d(x, sd[prefix + ".conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(x, sd, prefix + ".bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    stages = [x]
    for li, nblocks in enumerate(RESNET50_LAYERS, start=1):
        for bi in range(nblocks):
            stride = 2 if (bi == 0 and li > 1) else 1
            x = bottleneck(x, sd, f"{prefix}.layer{li}.{bi}", stride)
        stages.append(x)
    return (x, stages) if return_stages else x


# ------------------------------------------------------------------------------------------------
# a2  KPE (src/models/hands_light/model.py:444-460) and feature assembly (:258-274)
# ------------------------------------------------------------------------------------------------
def pos_enc(angle, n_freq=4):
    """model.py:444-451 / :453-460: stack([sin(2^k a), cos(2^k a)], -1) laid out (bz, L, c, 2)."""
    bz, c = angle.shape
    freq = (2 ** torch.arange(n_freq)).reshape(1, n_freq, 1).to(angle.device)  # int64, as the reference
    a = angle.reshape(bz, 1, c)
    return torch.stack([torch.sin(freq * a), torch.cos(freq * a)], dim=-1).reshape(bz, -1).float()


def assemble_features(crop_feat, glb_feat, center_angle, corner_angle, n_freq=4):
    """model.py:258-271: cat([crop+glb, center_enc, corner_enc], C) with encodings repeated 7x7."""
    bz, _, h, w = crop_feat.shape
    ce = pos_enc(center_angle, n_freq).view(bz, -1, 1, 1).repeat(1, 1, h, w)
    co = pos_enc(corner_angle, n_freq).view(bz, -1, 1, 1).repeat(1, 1, h, w)
    return torch.cat([crop_feat + glb_feat, ce, co], dim=1)


# ------------------------------------------------------------------------------------------------
# a3  feature_conv (model.py:91-101)
# ------------------------------------------------------------------------------------------------
def feature_conv(x, sd, p="feature_conv"):
    x = F.relu(F.conv2d(x, sd[p + ".0.weight"]))
    x = F.relu(F.conv2d(x, sd[p + ".2.weight"]))  # 3x3, padding 0: 7 -> 5
    x = F.relu(F.conv2d(x, sd[p + ".4.weight"]))  # 3x3, padding 0: 5 -> 3
    x = x.flatten(1)                              # NCHW flatten: index c*9 + h*3 + w
    return F.relu(F.linear(x, sd[p + ".7.weight"], sd[p + ".7.bias"]))


# ------------------------------------------------------------------------------------------------
# a5  6D <-> rotation matrix (pytorch3d.transforms.rotation_conversions; call sites
#     src/nets/hand_heads/hand_hmr.py:50-51,85-87).  PARITY UNPINNED (third party, absent).
# ------------------------------------------------------------------------------------------------
def rotation_6d_to_matrix(d6):
    """Gram-Schmidt on a1=d6[:3], a2=d6[3:]; b1,b2,b3 stacked as ROWS (dim=-2)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def matrix_to_rotation_6d(m):
    return m[..., :2, :].clone().reshape(*m.shape[:-2], 6)


# ------------------------------------------------------------------------------------------------
# a8  matrix -> quaternion -> axis-angle (common/rot.py:118-193, :55-83, :44-52)
# ------------------------------------------------------------------------------------------------
def _sqrt_positive_part(x):
    """rot.py:44-52."""
    return torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0)), torch.zeros_like(x))


def matrix_to_quaternion(matrix):
    """rot.py:118-177: four candidates, pick the one with the largest denominator (argmax)."""
    batch = matrix.shape[:-2]
    m = matrix.reshape(batch + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, -1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1, dtype=q_abs.dtype)
    cand = cand / (2.0 * q_abs[..., None].max(flr))
    idx = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, idx[..., None, None].expand(batch + (1, 4))).squeeze(-2)


def quaternion_to_axis_angle(q):
    """rot.py:55-83: atan2 half-angle; series 0.5 - x^2/48 when abs(angle) < 1e-6."""
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(ang), ang)
    s = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / safe)
    return q[..., 1:] / s


def matrix_to_axis_angle(matrix):
    """rot.py:180-193."""
    return quaternion_to_axis_angle(matrix_to_quaternion(matrix))


def axis_angle_to_matrix(aa):
    """rot.py:754-782 + :86-115 (via quaternion); used only by the is_flipped branch (model.py:345-353)."""
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = ang * 0.5
    small = ang.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(ang), ang)
    s = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / safe)
    q = torch.cat([torch.cos(half), aa * s], dim=-1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


# ------------------------------------------------------------------------------------------------
# a4  HandHMR / HMRLayer (src/nets/hand_heads/hand_hmr.py:46-92, src/nets/hmr_layer.py:67-86)
# ------------------------------------------------------------------------------------------------
def hand_hmr(feat, sd, p, n_iter=3):
    """feat is 2-D (use_pool=False, model.py:320-321) so cam_init takes it directly (hand_hmr.py:64)."""
    bz = feat.shape[0]
    lin = lambda x, q: F.linear(x, sd[q + ".weight"], sd[q + ".bias"])
    cam = lin(F.relu(lin(F.relu(lin(feat, p + ".cam_init.0")), p + ".cam_init.2")), p + ".cam_init.4")
    init_cam = cam.clone()
    ident6 = torch.tensor([1.0, 0, 0, 0, 1.0, 0], dtype=feat.dtype)   # hand_hmr.py:49-55
    pose6d = ident6.repeat(16).reshape(1, 96).repeat(bz, 1)
    shape = torch.zeros(bz, 10, dtype=feat.dtype)
    per_iter = []
    for _ in range(n_iter):
        # hmr_layer.py:80: cat([feat] + vectors) in dict order pose_6d, shape, cam_t/wp (hand_hmr.py:66-69)
        xc = torch.cat([feat, pose6d, shape, cam], dim=1)
        xc = F.relu(lin(xc, p + ".hmr_layer.refine.0"))   # Dropout is identity in eval
        xc = F.relu(lin(xc, p + ".hmr_layer.refine.3"))
        # hmr_layer.py:82-83: decoders in ModuleDict order pose_6d, cam_t/wp, shape; all read the same xc
        pose6d = lin(xc, p + ".hmr_layer.decoders.pose_6d") + pose6d
        cam = lin(xc, p + ".hmr_layer.decoders.cam_t/wp") + cam
        shape = lin(xc, p + ".hmr_layer.decoders.shape") + shape
        per_iter.append((pose6d.clone(), shape.clone(), cam.clone()))
    rotmat = rotation_6d_to_matrix(pose6d.reshape(-1, 6)).view(bz, 16, 3, 3)   # hand_hmr.py:85-87
    return {"pose_6d": pose6d, "shape": shape, "cam_t.wp": cam, "pose": rotmat,
            "cam_t.wp.init": init_cam, "_per_iter": per_iter}


# ------------------------------------------------------------------------------------------------
# a6  grasp classifier (model.py:117-125, 401-404)
# ------------------------------------------------------------------------------------------------
def grasp_classifier(shape, rotmat, feat_vec, sd, p="grasp_classifier"):
    bz = shape.shape[0]
    parts = [shape, rotmat.reshape(bz, -1)] + ([feat_vec] if feat_vec is not None else [])     # model.py:402-407
    x = torch.cat(parts, dim=1)
    for i in (0, 2, 4):
        x = F.relu(F.linear(x, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"]))
    return F.linear(x, sd[f"{p}.6.weight"], sd[f"{p}.6.bias"])


# ------------------------------------------------------------------------------------------------
# a9  MANO linear blend skinning (smplx.MANO.forward + smplx.lbs.*).  PARITY UNPINNED.
#     Restated from the published smplx algorithm; constructed by the
#     reference at common/body_models.py:92-99 with use_pca=False, flat_hand_mean=False.
# ------------------------------------------------------------------------------------------------
def batch_rodrigues(rot_vecs):
    """smplx.lbs.batch_rodrigues: theta = ||r + 1e-8||, R = I + sin*K + (1-cos)*K@K."""
    n = rot_vecs.shape[0]
    dt = rot_vecs.dtype
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle)[:, None]
    sin = torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=dt)
    Kmat = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(n, 3, 3)
    ident = torch.eye(3, dtype=dt)[None]
    return ident + sin * Kmat + (1 - cos) * torch.bmm(Kmat, Kmat)


def mano_lbs(betas, global_orient, hand_pose, asset, dtype=None):
    """smplx.MANO.forward: full_pose = cat(global, hand) + pose_mean; lbs(); + 5 fingertip joints.

    ``asset`` has v_template (778,3), shapedirs (778,3,10), posedirs (135,2334), J_regressor
    (16,778), lbs_weights (778,16), hands_mean (45,).  Returns vertices (B,778,3), joints (B,21,3).
    """
    dt = dtype or betas.dtype
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dt)
    v_template, shapedirs, posedirs = t(asset.v_template), t(asset.shapedirs), t(asset.posedirs)
    J_regressor, lbs_weights = t(asset.J_regressor), t(asset.lbs_weights)
    pose_mean = torch.cat([torch.zeros(3, dtype=dt), t(asset.hands_mean)])
    betas, global_orient, hand_pose = betas.to(dt), global_orient.to(dt), hand_pose.to(dt)
    B = betas.shape[0]

    full_pose = torch.cat([global_orient, hand_pose], dim=1) + pose_mean
    v_shaped = v_template + torch.einsum("bl,mkl->bmk", betas, shapedirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)
    rot_mats = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, 16, 3, 3)
    pose_feature = (rot_mats[:, 1:] - torch.eye(3, dtype=dt)).reshape(B, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, posedirs).view(B, -1, 3)

    # batch_rigid_transform
    rel_J = J.clone()
    parents = list(PARENTS)
    rel_J[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    T = torch.zeros(B, 16, 4, 4, dtype=dt)
    T[:, :, :3, :3] = rot_mats
    T[:, :, :3, 3] = rel_J
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, 16):
        chain.append(torch.matmul(chain[parents[i]], T[:, i]))
    G = torch.stack(chain, dim=1)
    posed_joints = G[:, :, :3, 3]
    J_h = torch.cat([J, torch.zeros(B, 16, 1, dtype=dt)], dim=2)[..., None]
    A = G.clone()
    A[:, :, :, 3] = G[:, :, :, 3] - torch.matmul(G, J_h)[..., 0]

    Tv = torch.matmul(lbs_weights[None].expand(B, -1, -1), A.view(B, 16, 16)).view(B, -1, 4, 4)
    v_h = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dt)], dim=2)
    verts = torch.matmul(Tv, v_h[..., None])[:, :, :3, 0]
    joints = torch.cat([posed_joints, verts[:, list(TIP_IDS)]], dim=1)
    return verts, joints


# ------------------------------------------------------------------------------------------------
# a10 camera + projection (common/camera.py:456-474, common/transforms.py:316-329,69-77,
#     common/data_utils.py:361-365)
# ------------------------------------------------------------------------------------------------
def weak_perspective_to_perspective(cam, focal_length, img_res, min_s=0.1):
    s = torch.clamp(cam[:, 0], min_s)
    return torch.stack([cam[:, 1], cam[:, 2], 2 * focal_length / (img_res * s + 1e-9)], dim=-1)


def project2d_batch(K, pts_cam):
    homo = torch.bmm(K, pts_cam.permute(0, 2, 1)).permute(0, 2, 1)
    return homo[:, :, :2] / homo[:, :, 2:3]


def normalize_kp2d(kp2d, img_res):
    out = kp2d.clone()
    out[:, :, :2] = 2.0 * kp2d[:, :, :2] / img_res - 1.0
    return out


# ------------------------------------------------------------------------------------------------
# a7  MANOHead.forward (src/nets/hand_heads/mano_head.py:21-65)
# ------------------------------------------------------------------------------------------------
def mano_head(rotmat, shape, cam, K, asset, img_res, postfix):
    rotmat_original = rotmat.clone()
    aa = matrix_to_axis_angle(rotmat.reshape(-1, 3, 3)).reshape(-1, 48)
    verts, joints = mano_lbs(shape, aa[:, :3], aa[:, 3:], asset)
    avg_f = (K[:, 0, 0] + K[:, 1, 1]) / 2.0
    cam_t = weak_perspective_to_perspective(cam, avg_f, img_res, 0.1)
    j3d_cam = joints + cam_t[:, None, :]
    v3d_cam = verts + cam_t[:, None, :]
    j2d = normalize_kp2d(project2d_batch(K, j3d_cam), img_res)
    out = {"cam_t.wp": cam, "cam_t": cam_t, "joints3d": joints, "vertices": verts,
           "j3d.cam": j3d_cam, "v3d.cam": v3d_cam, "j2d.norm": j2d, "beta": shape,
           "pose": rotmat_original}
    return {k + postfix: v for k, v in out.items()}


# ------------------------------------------------------------------------------------------------
# HandsLight.forward, default config (src/models/hands_light/model.py:187-437)
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def image_level_input(img, center_angle, corner_angle, mode, n_freq=4):
    """model.py:203-218: pos_enc 'center' / 'corner' / 'center+corner' -- the encoding repeated over the pixels as extra
    input channels of the hand trunk's (widened) conv1."""
    bz, _, w, h = img.shape
    parts = [img]
    if mode in ("center", "center+corner"):
        parts.append(pos_enc(center_angle, n_freq).view(bz, -1, 1, 1).repeat(1, 1, w, h))
    if mode in ("corner", "center+corner"):
        parts.append(pos_enc(corner_angle, n_freq).view(bz, -1, 1, 1).repeat(1, 1, w, h))
    return torch.cat(parts, dim=1)


def dense_pos_enc(angle, mask, n_freq=4, img_res_ds=224):
    """model.py:462-472: per-pixel encodings of the (bz, c, w, h) angle maps -- cat([sin, cos], dim=3) of the
    (bz, L, c, w, h) products reshaped to (bz, 2 L c, w, h), i.e. channel (l c + ci) 2 + {sin, cos} -- times the crop mask,
    then F.interpolate(bilinear, align_corners=True) to the trunk's input size."""
    bz, c, w, h = angle.shape
    freq = (2 ** torch.arange(n_freq)).reshape(1, n_freq, 1, 1, 1).to(angle.device)
    a = angle.reshape(bz, 1, c, w, h)
    enc = torch.cat([torch.sin(freq * a), torch.cos(freq * a)], dim=3).reshape(bz, -1, w, h).float()
    enc = enc * mask.unsqueeze(1).repeat(1, 2 * n_freq * c, 1, 1)
    return F.interpolate(enc, size=(img_res_ds, img_res_ds), mode="bilinear", align_corners=True)


def cam_conv_pos_enc(angle, mask, img_res_ds=224):
    """model.py:474-481: the masked (bz, 6, w, h) maps (angles, centred pixel offsets, normalised coordinates) resized as above."""
    angle = angle * mask.unsqueeze(1).repeat(1, angle.shape[1], 1, 1)
    return F.interpolate(angle, size=(img_res_ds, img_res_ds), mode="bilinear", align_corners=True)


def euler_angles_to_matrix_xyz(e):
    """pytorch3d.transforms.euler_angles_to_matrix(e, 'XYZ') (call site model.py:370-376; third party, absent: PARITY
    UNPINNED, published definition): Rx(e0) @ Ry(e1) @ Rz(e2) with the right-handed elementary rotations."""
    c, s = torch.cos(e), torch.sin(e)
    one, zero = torch.ones_like(c[..., 0]), torch.zeros_like(c[..., 0])
    rx = torch.stack([one, zero, zero, zero, c[..., 0], -s[..., 0], zero, s[..., 0], c[..., 0]], -1).reshape(e.shape[:-1] + (3, 3))
    ry = torch.stack([c[..., 1], zero, s[..., 1], zero, one, zero, -s[..., 1], zero, c[..., 1]], -1).reshape(e.shape[:-1] + (3, 3))
    rz = torch.stack([c[..., 2], -s[..., 2], zero, s[..., 2], c[..., 2], zero, zero, zero, one], -1).reshape(e.shape[:-1] + (3, 3))
    return torch.matmul(torch.matmul(rx, ry), rz)


def depth_head(x, sd, p="depth_mlp"):
    """model.py:134-155, 177-185, 438-441: the 7x7 (x, y) grid of torch.meshgrid(linspace(-1,1,7), linspace(-1,1,7)) ('ij')
    appended as two channels, then eight 3x3 / pad 1 convolutions with three align_corners bilinear upsamplings (x4, x4, x2)."""
    bz = x.shape[0]
    lin = torch.linspace(-1, 1, 7)
    xg, yg = torch.meshgrid(lin, lin, indexing="ij")
    x = torch.cat((x, xg.expand(bz, 1, -1, -1), yg.expand(bz, 1, -1, -1)), dim=1)
    conv = lambda t, i: F.conv2d(t, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"], padding=1)
    up = lambda t, k: F.interpolate(t, scale_factor=k, mode="bilinear", align_corners=True)
    x = F.relu(conv(x, 0)); x = F.relu(conv(x, 2)); x = up(x, 4)
    x = F.relu(conv(x, 5)); x = F.relu(conv(x, 7)); x = up(x, 4)
    x = F.relu(conv(x, 10)); x = F.relu(conv(x, 12)); x = up(x, 2)
    x = F.relu(conv(x, 15))
    return conv(x, 17)


def hands_light_forward(sd, asset_r, asset_l, inputs, meta_info, img_res=224, n_freq=4,
                        return_intermediates=False, pos_enc_mode="center+corner_latent", no_crops=False,
                        use_grasp_loss=True, use_glb_feat_w_grasp=True, separate_hands=False, regress_center_corner=False,
                        use_glb_feat=True, use_depth_loss=False, img_res_ds=224):
    """model.py:187-437 with tf_decoder=False.  ``pos_enc_mode``:
    'center+corner_latent' | 'sinusoidal_cc' (same code path, model.py:258-271 / 288-304), 'center' | 'corner' |
    'center+corner' | 'dense' (image level, model.py:203-224), 'dense_latent' | 'cam_conv' (per-pixel maps resized to the 7x7
    feature map, :244-256 / :276-288), 'pcl' | 'perspective_correction' (no encoding; the global rotation is corrected after
    the heads, :330-334 / :370-376) or None."""
    K = meta_info["intrinsics"]
    bz = inputs["img"].shape[0]
    features = feat_vec = None
    if use_glb_feat:                                                              # model.py:191-196
        features = resnet50_trunk(inputs["img"], sd, "backbone")                  # model.py:193
        feat_vec = features.view(bz, features.shape[1], -1).sum(dim=2)            # model.py:196 (SUM)
    r_feat = l_feat = None
    if no_crops:                                                                  # model.py:199-201, 316-318
        # HandHMR.forward(features, use_pool=True): nn.AdaptiveAvgPool2d(1) (hand_hmr.py:73-78)
        r_vec = l_vec = F.adaptive_avg_pool2d(features, 1).view(bz, -1)
    else:
        if pos_enc_mode in ("center", "corner", "center+corner"):
            r_in = image_level_input(inputs["r_img"], inputs["r_center_angle"], inputs["r_corner_angle"], pos_enc_mode, n_freq)
            l_in = image_level_input(inputs["l_img"], inputs["l_center_angle"], inputs["l_corner_angle"], pos_enc_mode, n_freq)
        elif pos_enc_mode == "dense":                                             # model.py:220-224
            r_in = torch.cat([inputs["r_img"], dense_pos_enc(inputs["r_dense_angle"], inputs["r_dense_mask"], n_freq, img_res_ds)], 1)
            l_in = torch.cat([inputs["l_img"], dense_pos_enc(inputs["l_dense_angle"], inputs["l_dense_mask"], n_freq, img_res_ds)], 1)
        else:
            r_in, l_in = inputs["r_img"], inputs["l_img"]
        r_feat = resnet50_trunk(r_in, sd, "hand_backbone_r" if separate_hands else "hand_backbone")   # model.py:226-239
        l_feat = resnet50_trunk(l_in, sd, "hand_backbone_l" if separate_hands else "hand_backbone")
        if pos_enc_mode in ("center+corner_latent", "sinusoidal_cc"):
            glb = features if use_glb_feat else torch.zeros_like(r_feat)          # model.py:263-267 (x + 0 == x exactly)
            r_cat = assemble_features(r_feat, glb, inputs["r_center_angle"], inputs["r_corner_angle"], n_freq)
            l_cat = assemble_features(l_feat, glb, inputs["l_center_angle"], inputs["l_corner_angle"], n_freq)
        elif pos_enc_mode in ("dense_latent", "cam_conv"):                        # model.py:244-256 / 276-288
            cats = []
            for sd_, ft in (("r", r_feat), ("l", l_feat)):
                ang, msk = inputs[f"{sd_}_dense_angle"], inputs[f"{sd_}_dense_mask"]
                e = (dense_pos_enc(ang, msk, n_freq, img_res_ds) if pos_enc_mode == "dense_latent"
                     else cam_conv_pos_enc(ang, msk, img_res_ds))
                e = F.interpolate(e, size=tuple(ft.shape[2:]), mode="bilinear", align_corners=True)
                cats.append(torch.cat([ft + features if use_glb_feat else ft, e], dim=1))
            r_cat, l_cat = cats
        else:
            r_cat, l_cat = r_feat, l_feat            # the global features are NOT added on these routes (model.py:241-304)
        if use_depth_loss:                                                        # model.py:308-310
            depth_r, depth_l = depth_head(r_cat, sd), depth_head(l_cat, sd)
        r_vec = feature_conv(r_cat, sd)                                           # model.py:313
        l_vec = feature_conv(l_cat, sd)                                           # model.py:314
    hmr_r = hand_hmr(r_vec, sd, "head_r")                                         # model.py:320
    hmr_l = hand_hmr(l_vec, sd, "head_l")                                         # model.py:321

    root_r, root_l = hmr_r["cam_t.wp"], hmr_l["cam_t.wp"]
    root_r_init, root_l_init = hmr_r["cam_t.wp.init"], hmr_l["cam_t.wp.init"]
    if pos_enc_mode == "pcl":         # model.py:330-334: IN PLACE on the head outputs -- the flip branch and the grasp head see it
        hmr_r["pose"][:, 0] = torch.bmm(inputs["r_rot"], hmr_r["pose"][:, 0])
        hmr_l["pose"][:, 0] = torch.bmm(inputs["l_rot"], hmr_l["pose"][:, 0])
    pose_r, shape_r, pose_l, shape_l = hmr_r["pose"], hmr_r["shape"], hmr_l["pose"], hmr_l["shape"]

    flipped = meta_info["is_flipped"].bool()
    if int(flipped.sum()) > 0:                                                    # model.py:341-368
        sgn = torch.tensor([[1.0, -1.0, 1.0]])

        def mirror(pose):
            aa = matrix_to_axis_angle(pose).view(bz, -1).clone()
            aa[:, 1::3] *= -1
            aa[:, 2::3] *= -1
            return axis_angle_to_matrix(aa.view(bz, 16, 3)).view(bz, 16, 3, 3)

        f1 = flipped[:, None]
        f3 = flipped[:, None, None, None]
        root_r, root_l = (torch.where(f1, hmr_l["cam_t.wp"] * sgn, root_r),
                          torch.where(f1, hmr_r["cam_t.wp"] * sgn, root_l))
        pose_r, pose_l = (torch.where(f3, mirror(hmr_l["pose"]), pose_r),
                          torch.where(f3, mirror(hmr_r["pose"]), pose_l))
        shape_r, shape_l = (torch.where(f1, hmr_l["shape"], shape_r),
                            torch.where(f1, hmr_r["shape"], shape_l))
        root_r_init, root_l_init = (torch.where(f1, hmr_l["cam_t.wp.init"] * sgn, root_r_init),
                                    torch.where(f1, hmr_r["cam_t.wp.init"] * sgn, root_l_init))

    if pos_enc_mode == "perspective_correction":
        # model.py:370-376, after the flip swap and IN PLACE: without a flipped sample in the batch pose_r IS the head's output
        # tensor, so the grasp head below reads the corrected rotation; with one, torch.where made a copy and it does not
        for pose, ang in ((pose_r, inputs["r_center_angle"]), (pose_l, inputs["l_center_angle"])):
            rot = euler_angles_to_matrix_xyz(torch.cat([-ang, torch.zeros(bz, 1)], dim=-1))
            pose[:, 0] = torch.matmul(rot, pose[:, 0])

    out = {}
    mr = mano_head(pose_r, shape_r, root_r, K, asset_r, img_res, ".r")            # model.py:378-383
    ml = mano_head(pose_l, shape_l, root_l, K, asset_l, img_res, ".l")            # model.py:385-390
    mr["cam_t.wp.init.r"] = root_r_init
    ml["cam_t.wp.init.l"] = root_l_init
    out.update({"mano." + k: v for k, v in mr.items()})
    out.update({"mano." + k: v for k, v in ml.items()})
    # model.py:401-411: the grasp head reads the UN-flipped HMR outputs
    if use_grasp_loss:
        gf = feat_vec if use_glb_feat_w_grasp else None
        out["grasp.r"] = grasp_classifier(hmr_r["shape"], hmr_r["pose"], gf, sd)
        out["grasp.l"] = grasp_classifier(hmr_l["shape"], hmr_l["pose"], gf, sd)
    if use_depth_loss:                                                            # model.py:422-424
        out["depth.r"], out["depth.l"] = depth_r.squeeze(1), depth_l.squeeze(1)
    if regress_center_corner:                                                     # model.py:426-433
        def head3(x, p):
            for i in (0, 2):
                x = F.relu(F.linear(x, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"]))
            return F.linear(x, sd[f"{p}.4.weight"], sd[f"{p}.4.bias"])
        out["center.r"], out["center.l"] = head3(r_vec, "center_head"), head3(l_vec, "center_head")
        out["corner.r"], out["corner.l"] = head3(r_vec, "corner_head"), head3(l_vec, "corner_head")
    if return_intermediates:
        inter = {"features": features, "feat_vec": feat_vec, "r_feat": r_feat, "l_feat": l_feat,
                 "r_vec": r_vec, "l_vec": l_vec, "hmr_r": hmr_r, "hmr_l": hmr_l}
        return out, inter
    return out


# ------------------------------------------------------------------------------------------------
# Metric the bench reports next to throughput ("MPJPE vs ref"), root-aligned as
# src/utils/eval_modules.py:97-134 -> common/metrics.py:23-32 (x1000 -> mm).
# ------------------------------------------------------------------------------------------------
def mpjpe_ra_mm(j_pred, j_ref):
    p = j_pred - j_pred[:, :1]
    g = j_ref - j_ref[:, :1]
    return float((p - g).norm(dim=-1).mean() * 1000.0)
