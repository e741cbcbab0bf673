"""(f2) crop / KPE front-end: oracle vs the reference-generated fixture and vs OpenCV's documented
properties (CPU), HIP kernels vs the oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import frontend_oracle as F

GOLD = os.path.join(os.path.dirname(__file__), "golden", "frontend.npz")


# ---------------------------------------------------------------- CPU: oracle pinned / properties
def test_oracle_crop_window_matches_reference():
    g = np.load(GOLD)
    for box, scale, nb, tr in zip(g["box_xywh"], g["scale"], g["new_bbox"], g["trans"]):
        b = None if box[0] < 0 else box.astype(np.int16)
        patch, new_bbox = F.crop_window(b, 224, float(scale))
        assert np.array_equal(np.asarray(new_bbox).astype(np.int64), nb)
        t = F.gen_trans_from_patch(patch[0], patch[1], patch[2], patch[3], 224, 224)
        assert t.dtype == np.float32 and np.array_equal(t, tr)


def test_oracle_gen_trans_matches_reference():
    g = np.load(GOLD)
    for a, t in zip(g["gen_trans_in"], g["gen_trans_out"]):
        assert np.array_equal(F.gen_trans_from_patch(a[0], a[1], a[2], a[3], a[4], a[5]), t)


def test_oracle_cubic_weights():
    tab = F.cubic_table()
    assert tab.shape == (32, 4) and np.array_equal(tab[0], np.array([0, 1, 0, 0], np.float32))
    np.testing.assert_allclose(tab.sum(1), 1.0, atol=1e-7)
    np.testing.assert_allclose(tab[16], [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-7)   # A=-0.75 at x=0.5


def test_oracle_warp_identity_and_shift_are_exact():
    rng = np.random.default_rng(0)
    img = rng.random((40, 52, 3), dtype=np.float32)
    eye = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    assert np.array_equal(F.warp_affine_cubic(img, eye, 40, 52), img)
    sh = np.array([[1, 0, 3], [0, 1, -2]], np.float32)          # dst(x, y) = src(x - 3, y + 2)
    out = F.warp_affine_cubic(img, sh, 40, 52)
    assert np.array_equal(out[:38, 3:], img[2:, :49])
    assert np.all(out[38:] == 0) and np.all(out[:, :3] == 0)    # constant-0 border


def test_oracle_warp_constant_and_ramp():
    yy, xx = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(64, dtype=np.float32), indexing="ij")
    img = np.stack([xx, yy, 0.5 * xx + 0.25 * yy], -1) / 64
    t = F.gen_trans_from_patch(32, 30, 20, 20, 48, 48)           # 2.4x zoom
    out = F.warp_affine_cubic(img, t, 48, 48)
    M = F.invert_affine(t)
    u = np.arange(48)
    sx = M[0] * u + M[2]
    # A=-0.75 cubic is not linear-exact (only A=-0.5 is): error of a unit ramp <= 3/64 px, plus the
    # 1/32-px quantisation of the fixed-point coordinates
    np.testing.assert_allclose(out[10, :, 0] * 64, sx, atol=3.0 / 64 + 1.0 / 32)
    const = F.warp_affine_cubic(np.full((64, 64, 3), 0.37, np.float32), t, 48, 48)
    np.testing.assert_allclose(const, 0.37, atol=1e-7)                             # weights sum to 1


def test_oracle_bbox_and_angles():
    j = np.zeros((21, 3), np.float32)
    j[:, 0] = np.linspace(-0.5, 0.25, 21)
    j[:, 1] = np.linspace(0.1, 0.6, 21)
    box, og = F.bbox_from_joints2d(j, 224)
    assert box.dtype == np.int16 and list(box) == [55, 122, 83, 55] and np.array_equal(box, og)
    none, og = F.bbox_from_joints2d(np.full((21, 2), -3.0, np.float32), 224)      # off-image hand
    assert none is None and list(og) == [0, 0, 223, 223]
    K = np.array([[1000, 0, 112], [0, 1000, 112], [0, 0, 1]], np.float32)
    c, q = F.kpe_angles(np.array([10, 20, 110, 220], np.int16), K)
    assert c.dtype == np.float32 and q.shape == (8,)
    np.testing.assert_allclose(c, np.arctan2([60 - 112, 120 - 112], 1000).astype(np.float32))
    # corners: int16 - float32 K stays float32 in numpy, so these angles are float32 arithmetic
    np.testing.assert_allclose(q[[0, 1, 6, 7]], np.arctan2(np.float32([10 - 112, 20 - 112, 110 - 112, 220 - 112]), np.float32(1000)))
    assert q[0] == q[2] and q[1] == q[5] and q[4] == q[6] and q[3] == q[7]


# ---------------------------------------------------------------- GPU: HIP vs oracle
def _batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, 224, 224, generator=g)
    img = (img + 0.3 * torch.randn(B, 3, 224, 224, generator=g)).clamp(-0.2, 1.2)       # exercise the clip
    ctr = 0.6 * (torch.rand(B, 2, 1, 2, generator=g) * 2 - 1)
    ext = 0.05 + 0.5 * torch.rand(B, 2, 1, 2, generator=g)
    j = ctr + ext * (torch.rand(B, 2, 21, 2, generator=g) * 2 - 1)
    j = torch.cat([j, torch.ones(B, 2, 21, 1)], -1)
    j[1, 0] = -4.0                    # right hand of sample 1 off-image -> no box -> whole image
    j[2, 1, :, 0] = 0.123             # left hand of sample 2 zero width
    j[3, 0] = j[3, 0] * 3             # partly off-image -> clipped box, crop window leaves the image
    K = torch.eye(3).repeat(B, 1, 1)
    K[:, 0, 0] = 900 + 200 * torch.rand(B, generator=g)
    K[:, 1, 1] = 900 + 200 * torch.rand(B, generator=g)
    K[:, 0, 2] = 112 + 10 * torch.randn(B, generator=g)
    K[:, 1, 2] = 112 + 10 * torch.randn(B, generator=g)
    return img, j[:, 0].contiguous(), j[:, 1].contiguous(), K


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1.5, 2.5])
def test_gpu_frontend_matches_oracle(scale):
    from hands_amd import HandsFrontEnd
    B = 6
    img, jr, jl, K = _batch(B, 3)
    fe = HandsFrontEnd({"bbox_scale": scale})
    dev = torch.device("cuda:0")
    out = fe(img.to(dev), jr.to(dev), jl.to(dev), K.to(dev))
    geo = fe.boxes(jr.to(dev), jl.to(dev), K.to(dev))
    torch.cuda.synchronize()
    assert out["r_bbox"].dtype == torch.int16 and out["r_img"].shape == (B, 3, 224, 224)
    worst = 0.0
    for b in range(B):
        ref = F.frontend_sample(img[b].numpy(), jr[b].numpy(), jl[b].numpy(), K[b].numpy(), bbox_scale=scale)
        for h in "rl":
            assert np.array_equal(out[f"{h}_bbox"][b].cpu().numpy(), ref[f"{h}_bbox"].astype(np.int16)), (b, h)
            assert np.array_equal(out[f"{h}_bbox_og"][b].cpu().numpy().astype(np.int64), np.asarray(ref[f"{h}_bbox_og"]).astype(np.int64))
            # center: float64 atan2 rounded to float32 (<= 1 ulp); corner: float32 atan2f (<= 4 ulp)
            for k, ulps in (("center_angle", 1), ("corner_angle", 4)):
                a, r = out[f"{h}_{k}"][b].cpu().numpy(), ref[f"{h}_{k}"]
                assert np.all(np.abs(a - r) <= ulps * np.spacing(np.abs(r)).astype(np.float32)), (b, h, k)
            t = geo[f"{h}_trans"][b].cpu().numpy()
            tr = ref[f"{h}_trans"].reshape(6)
            assert np.all(np.abs(t - tr) <= np.maximum(np.spacing(np.abs(tr)), 1e-12)), (b, h, t, tr)
            # pure warp parity: the oracle warps with the device's own affine
            patch = F.warp_affine_cubic(img[b].numpy().transpose(1, 2, 0), t.reshape(2, 3), 224, 224)
            want = F.normalize_img(np.clip(patch, 0, 1).transpose(2, 0, 1), fe.mean, fe.std)
            err = np.abs(out[f"{h}_img"][b].cpu().numpy() - want).max()
            worst = max(worst, err)
            assert err <= 2e-6, (b, h, err)
        assert np.abs(out["img"][b].cpu().numpy() - ref["img"]).max() <= 1e-6
    # whole-image fallback is an exact copy before Normalize
    want = F.normalize_img(np.clip(img[1].numpy(), 0, 1), fe.mean, fe.std)
    assert np.array_equal(out["r_img"][1].cpu().numpy(), want)
    print("front-end max abs crop err", worst)


@pytest.mark.gpu
def test_gpu_frontend_feeds_forward_batch_independent():
    """bz=64 front-end: first 2 samples identical to a bz=2 run (per-sample independence, bit-exact)."""
    from hands_amd import HandsFrontEnd
    img, jr, jl, K = _batch(64, 9)
    dev = torch.device("cuda:0")
    fe = HandsFrontEnd()
    big = fe(img.to(dev), jr.to(dev), jl.to(dev), K.to(dev))
    small = fe(img[:2].to(dev), jr[:2].to(dev), jl[:2].to(dev), K[:2].to(dev))
    for k, v in small.items():
        assert torch.equal(big[k][:2], v), k


@pytest.mark.gpu
def test_gpu_frontend_into_hands_light():
    """The front-end's dict is what HandsLight.forward consumes (§8b input contract)."""
    import hands_amd
    from hands_amd import HandsFrontEnd
    dev = torch.device("cuda:0")
    img, jr, jl, K = _batch(4, 21)
    inputs = HandsFrontEnd()(img.to(dev), jr.to(dev), jl.to(dev), K.to(dev))
    model = hands_amd.apply_recipe(hands_amd.HandsLight()).to(dev).eval()
    meta = {"intrinsics": K.to(dev), "is_flipped": torch.zeros(4, dtype=torch.int64, device=dev)}
    out = model(inputs, meta)
    assert out["mano.vertices.r"].shape == (4, 778, 3) and all(torch.isfinite(v).all() for v in out.values())


def test_oracle_dense_maps_layout():
    """hands_light_dataset.py:281-333 restated (numpy lines of `__getitem__`, not callable on their own): window pixels in the
    top-left corner, first index x, zero elsewhere, mask on the window; cam_conv = angles + centred offsets + normalised coords."""
    K = np.array([[1000.0, 0, 110.5], [0, 950.0, 120.25], [0, 0, 1]], np.float32)
    ang, msk = F.dense_maps([30, 50, 129, 199], K, 224, cam_conv=True)
    assert ang.shape == (6, 224, 224) and msk.shape == (224, 224) and ang.dtype == np.float32
    assert msk[:100, :150].all() and msk.sum() == 100 * 150 and not ang[:, 100:].any() and not ang[:, :, 150:].any()
    assert ang[0, 7, 3] == np.float32(np.arctan2(37 - np.float64(K[0, 2]), np.float64(K[0, 0])))       # x along the FIRST index
    assert ang[1, 7, 3] == np.float32(np.arctan2(53 - np.float64(K[1, 2]), np.float64(K[1, 1])))
    assert ang[2, 7, 3] == np.float32(37 - 110.5) and ang[3, 7, 3] == np.float32(53 - 120.25)
    assert ang[4, 7, 3] == np.float32(2 * 37 / 224 - 1) and ang[5, 7, 3] == np.float32(2 * 53 / 224 - 1)
    a2, m2 = F.dense_maps([30, 50, 129, 199], K, 224)
    assert np.array_equal(a2, ang[:2]) and np.array_equal(m2, msk)


@pytest.mark.gpu
@pytest.mark.parametrize("pos_enc", ["dense_latent", "cam_conv"])
def test_gpu_frontend_dense_maps_match_oracle(pos_enc):
    from hands_amd import HandsFrontEnd
    B = 6
    img, jr, jl, K = _batch(B, 4)
    fe = HandsFrontEnd({"pos_enc": pos_enc})
    dev = torch.device("cuda:0")
    out = fe(img.to(dev), jr.to(dev), jl.to(dev), K.to(dev))
    torch.cuda.synchronize()
    n = 6 if pos_enc == "cam_conv" else 2
    for b in range(B):
        for h in "rl":
            ang, msk = F.dense_maps(out[f"{h}_bbox"][b].cpu().numpy(), K[b].numpy(), 224, cam_conv=pos_enc == "cam_conv")
            got = out[f"{h}_dense_angle"][b].cpu().numpy()
            assert got.shape == (n, 224, 224) and np.array_equal(out[f"{h}_dense_mask"][b].cpu().numpy(), msk)
            assert np.all(np.abs(got[:2] - ang[:2]) <= np.spacing(np.abs(ang[:2])))       # double atan2 rounded to float32: <= 1 ulp
            assert np.array_equal(got[2:], ang[2:])


@pytest.mark.gpu
def test_gpu_frontend_dense_into_hands_light():
    """Front-end maps -> HandsLight(pos_enc='dense_latent') on the device == the oracle's forward on the same tensors."""
    import hands_amd
    from hands_amd.mano import synthetic_mano_asset
    from oracle import hands_oracle as O
    B = 4
    img, jr, jl, K = _batch(B, 5)
    args = type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS, pos_enc="dense_latent"))
    dev = torch.device("cuda:0")
    inputs = hands_amd.HandsFrontEnd(args)(img.to(dev), jr.to(dev), jl.to(dev), K.to(dev))
    meta = {"intrinsics": K.to(dev), "is_flipped": torch.zeros(B, dtype=torch.long, device=dev)}
    model = hands_amd.apply_recipe(hands_amd.HandsLight(args=args)).eval().to(dev)
    out = model(inputs, meta)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = O.hands_light_forward(sd, synthetic_mano_asset(True), synthetic_mano_asset(False),
                                {k: v.cpu().float() if v.is_floating_point() else v.cpu() for k, v in inputs.items()},
                                {k: v.cpu() for k, v in meta.items()}, pos_enc_mode="dense_latent")
    for hn in "rl":
        assert (out[f"mano.vertices.{hn}"].cpu() - ref[f"mano.vertices.{hn}"]).abs().max().item() < 1e-6


@pytest.mark.gpu
def test_gpu_frontend_no_intrx_matches_oracle():
    """args.no_intrx (hands_light_dataset.py:247-253): the encodings come from the float64 stand-in intrinsics f = c = img_res / 2;
    boxes, crops and the intrinsics handed to the model do not change."""
    from hands_amd import HandsFrontEnd
    B = 6
    img, jr, jl, K = _batch(B, 6)
    dev = torch.device("cuda:0")
    base = HandsFrontEnd({"pos_enc": "cam_conv"})(img.to(dev), jr.to(dev), jl.to(dev), K.to(dev))
    out = HandsFrontEnd({"pos_enc": "cam_conv", "no_intrx": True})(img.to(dev), jr.to(dev), jl.to(dev), K.to(dev))
    torch.cuda.synchronize()
    Kn = F.no_intrx_matrix(224)
    for b in range(B):
        for h in "rl":
            bbox = out[f"{h}_bbox"][b].cpu().numpy()
            assert np.array_equal(bbox, base[f"{h}_bbox"][b].cpu().numpy()) and torch.equal(out[f"{h}_img"][b], base[f"{h}_img"][b])
            ce, co = F.kpe_angles(bbox, Kn)
            for got, ref in ((out[f"{h}_center_angle"][b], ce), (out[f"{h}_corner_angle"][b], co)):
                assert np.all(np.abs(got.cpu().numpy() - ref) <= np.spacing(np.abs(ref)).astype(np.float32))     # double atan2: <= 1 ulp
            ang, msk = F.dense_maps(bbox, Kn, 224, cam_conv=True)
            got = out[f"{h}_dense_angle"][b].cpu().numpy()
            assert np.all(np.abs(got[:2] - ang[:2]) <= np.spacing(np.abs(ang[:2]))) and np.array_equal(got[2:], ang[2:])
    assert not torch.equal(out["r_corner_angle"], base["r_corner_angle"])
