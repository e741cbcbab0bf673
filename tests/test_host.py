"""CPU tests of the host side: strict dict, packing identities, C-ABI library exports, sharding."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import hands_amd
from hands_amd import _lib
from hands_amd.dist import pack_predictions, shard_batch, shard_range, unpack_predictions
from hands_amd.packing import fold_bn, hmr_state_columns, pack_conv, pack_linear, pack_mano
from hands_amd.xdict import prefix_dict, xdict

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


# ---- xdict (reference: common/xdict.py) -----------------------------------------------------------
def test_xdict_strict_assignment_and_merge():
    d = xdict({"a": 1})
    with pytest.raises(AssertionError):
        d["a"] = 2                       # xdict.py:50-55
    d.overwrite("a", 3)
    assert d["a"] == 3
    with pytest.raises(AssertionError):
        d.merge({"a": 0})                # xdict.py:89-104
    d.merge({"b": 2})
    assert d.prefix("p.").sorted_keys() == ["p.a", "p.b"]
    assert d.postfix(".r").sorted_keys() == ["a.r", "b.r"]
    assert xdict({"cam_t/wp": 1}).replace_keys("/", ".").sorted_keys() == ["cam_t.wp"]
    assert d.search("a").sorted_keys() == ["a"] and d.rm("a").sorted_keys() == ["b"]
    assert d.subset(["b"]) == {"b": 2}
    assert prefix_dict({"x": 1}, "mano.") == {"mano.x": 1}


def test_xdict_detach_and_invalid(capsys):
    d = xdict({"t": torch.ones(2, requires_grad=True) * 2, "l": [torch.zeros(1)], "s": "name"})
    out = d.detach()
    assert not out["t"].requires_grad and out["l"][0].device.type == "cpu" and out["s"] == "name"
    assert not d.has_invalid()
    assert xdict({"x": torch.tensor([float("nan")])}).has_invalid()
    assert "nan" in capsys.readouterr().out
    m = xdict({"v": torch.ones(2), "w": [torch.ones(1)]}).mul(2)
    assert m["v"].tolist() == [2, 2] and m["w"][0].item() == 2


def test_stream_xdict_joins_at_first_use(monkeypatch):
    """The asynchronous-tail result container: empty at the C level until ANY access makes the current stream
    wait for the producer; dict(d), {**d}, merge and the xdict helpers all go through the join."""
    from hands_amd.xdict import stream_xdict
    waits = []

    class _Stream:
        def wait_event(self, ev):
            waits.append(ev)

    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: _Stream())
    mk = lambda: stream_xdict({"a": torch.ones(2), "b": torch.zeros(3)}, object(), "cpu")
    d = mk()
    assert d.is_pending and dict.__len__(d) == 0 and not waits
    assert sorted(dict(d)) == ["a", "b"] and len(waits) == 1 and not d.is_pending
    assert d["a"].sum() == 2 and len(waits) == 1                       # joined once
    for use in (lambda x: {**x}, lambda x: x.prefix("p."), lambda x: list(x), lambda x: len(x), lambda x: "a" in x,
                lambda x: x.items(), lambda x: x.detach(), lambda x: x.get("a"), lambda x: x == {}, lambda x: repr(x)):
        n, x = len(waits), mk()
        use(x)
        assert len(waits) == n + 1 and not x.is_pending, use
    o, x = xdict({"c": 1}), mk()
    o.merge(x)
    assert sorted(o) == ["a", "b", "c"] and isinstance(x, xdict)
    x = mk()
    with pytest.raises(AssertionError):
        x["a"] = 1                                                      # strict assignment still holds


# ---- packing --------------------------------------------------------------------------------------
def _packed_conv_reference(pc, x_nhwc):
    """Emulate the kernel's GEMM view on CPU with the PACKED weights (torch fp64)."""
    B, H, W, C = x_nhwc.shape
    x = x_nhwc.double().permute(0, 3, 1, 2)
    w = pc.w.double()[:, : pc.KH * pc.KW * pc.Cin].view(-1, pc.KH, pc.KW, pc.Cin).permute(0, 3, 1, 2)
    y = F.conv2d(x, w, pc.bias.double(), stride=pc.stride, padding=pc.pad)
    return y[:, : pc.Cout]


def test_pack_conv_bn_fold_matches_conv_then_bn():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 9, 9, generator=g)
    w = torch.randn(24, 16, 3, 3, generator=g)
    bn = [torch.rand(24, generator=g) + 0.5, torch.randn(24, generator=g), torch.randn(24, generator=g),
          torch.rand(24, generator=g) + 0.5]
    ref = F.batch_norm(F.conv2d(x.double(), w.double(), stride=2, padding=1), bn[2].double(), bn[3].double(),
                       bn[0].double(), bn[1].double(), False, 0.0, 1e-5)
    pc = pack_conv(*fold_bn(w, *bn), 2, 1, "cpu")
    assert pc.w.shape == (128, 144) and pc.Kpad == 144 and pc.Cout == 24 and pc.macs_per_pixel == 24 * 16 * 9
    got = _packed_conv_reference(pc, x.permute(0, 2, 3, 1))
    assert (got - ref).abs().max() < 1e-5
    assert torch.all(pc.w[24:] == 0) and torch.all(pc.bias[24:] == 0)


def test_pack_stem_pads_rgb_to_4():
    w = torch.randn(64, 3, 7, 7)
    pc = pack_conv(w, None, 2, 3, "cpu", cin_pad_to=4)
    assert pc.Cin == 4 and pc.Kpad == 208 and pc.macs_per_pixel == 64 * 147
    wk = pc.w[:64, :196].view(64, 7, 7, 4)
    assert torch.equal(wk[..., :3], w.permute(0, 2, 3, 1)) and torch.all(wk[..., 3] == 0)


def test_hmr_state_and_grasp_column_permutations():
    F_ = 32
    cols = hmr_state_columns(F_)
    assert len(cols) == F_ + 109 and len(set(cols)) == len(cols) and max(cols) == F_ + 110
    g = torch.Generator().manual_seed(1)
    w = torch.randn(8, F_ + 109, generator=g)
    b = torch.randn(8, generator=g)
    pc = pack_linear(w, b, "cpu", col_index=cols, k_total=F_ + 112)
    feat, pose, shape, cam = (torch.randn(3, n, generator=g) for n in (F_, 96, 10, 3))
    row = torch.zeros(3, F_ + 112)
    row[:, :F_], row[:, F_:F_ + 96], row[:, F_ + 96:F_ + 106], row[:, F_ + 108:F_ + 111] = feat, pose, shape, cam
    ref = F.linear(torch.cat([feat, pose, shape, cam], 1), w, b)
    got = F.linear(row, pc.w[:8, : F_ + 112], pc.bias[:8])
    assert (got - ref).abs().max() < 1e-5
    # decoders stacked in state order with residual columns
    rows = list(range(96)) + [96 + i for i in range(10)] + [108 + i for i in range(3)]
    wd = torch.randn(109, 16, generator=g)
    pd = pack_linear(wd, None, "cpu", row_index=rows, n_total=112)
    assert pd.Cout == 112 and torch.equal(pd.w[108:111, :16], wd[106:109]) and torch.all(pd.w[106:108] == 0)


def test_pack_mano_blend_matrix_reproduces_blendshapes():
    a = hands_amd.synthetic_mano_asset(True)
    mp = pack_mano(a, "cpu")
    g = torch.Generator().manual_seed(2)
    beta, pf = torch.randn(4, 10, generator=g).double(), torch.randn(4, 135, generator=g).double()
    row = torch.zeros(4, 160, dtype=torch.float64)
    row[:, :10], row[:, 10:145] = beta, pf
    got = row @ mp["blend"].w.double()[:2336].T + mp["blend"].bias.double()[:2336]
    ref = (torch.from_numpy(a.v_template).double()[None] +
           torch.einsum("bl,mkl->bmk", beta, torch.from_numpy(a.shapedirs).double())).reshape(4, -1) + \
        pf @ torch.from_numpy(a.posedirs).double()
    assert (got[:, :2334] - ref).abs().max() < 1e-7 and torch.all(got[:, 2334:] == 0)
    J = torch.from_numpy(a.J_regressor).double() @ (ref - pf @ torch.from_numpy(a.posedirs).double()).view(4, 778, 3)
    J2 = mp["J_template"].double()[None] + (mp["J_shapedirs"].double() @ beta.T).T.reshape(4, 16, 3)
    assert (J - J2).abs().max() < 1e-6


# ---- C-ABI library -------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hands_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(hands_[a-z0-9_]+)\s*\(", header, flags=re.M))
    declared |= set(re.findall(r"^long long\s+(hands_[a-z0-9_]+)\s*\(", header, flags=re.M))
    declared -= {"hands_conv_desc", "hands_mano_consts", "hands_mano_out"}
    assert declared == set(_lib.SIGNATURES) | set(_lib.EXTRA_SYMBOLS), declared ^ (set(_lib.SIGNATURES) | set(_lib.EXTRA_SYMBOLS))
    L = _lib.lib()                                   # loads without a GPU
    for name in declared:
        assert hasattr(L, name), name
    assert L.hands_abi_version() == _lib.ABI_VERSION == 5
    assert L.hands_error_string(0) == b"ok" and b"invalid" in L.hands_error_string(10001)
    assert ctypes.sizeof(_lib.ConvDesc) == 16 * 4


def test_c_packers_equal_the_python_restatement_bit_for_bit():
    """`hands_pack_*` live behind the C ABI (csrc/pack.cpp, host code); hands_amd.packing is a
    ctypes wrapper.  tests/ref_packing.py is an independent pure-torch restatement of the layouts: every
    packed weight / bias must be equal bit for bit."""
    import ref_packing as R
    from hands_amd.packing import pack_conv1x1_dual
    g = torch.Generator().manual_seed(4)
    eq = lambda a, b: (torch.equal(a.w, b.w) and torch.equal(a.bias, b.bias) and
                       (a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad, a.Kpad, a.macs_per_pixel) ==
                       (b.Cin, b.Cout, b.KH, b.KW, b.stride, b.pad, b.Kpad, b.macs_per_pixel))
    for (Cout, Cin, k, cin_pad) in ((24, 16, 3, None), (64, 3, 7, 4), (130, 32, 1, None), (256, 64, 1, None)):
        w = torch.randn(Cout, Cin, k, k, generator=g)
        bn = [torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g), torch.randn(Cout, generator=g),
              torch.rand(Cout, generator=g) + 0.5]
        wf, bf = fold_bn(w, *bn)
        wr, br = R.fold_bn(w, *bn)
        assert wf.dtype == torch.float64 and torch.equal(wf, wr) and torch.equal(bf, br)
        assert eq(pack_conv(wf, bf, 2, 1, "cpu", cin_pad_to=cin_pad), R.pack_conv(wr, br, 2, 1, "cpu", cin_pad_to=cin_pad))
        assert eq(pack_conv(w, None, 1, 0, "cpu", cin_pad_to=cin_pad), R.pack_conv(w, None, 1, 0, "cpu", cin_pad_to=cin_pad))
    # linear layers with column / row permutations (HMR state row, decoder stack)
    F_ = 48
    w, b = torch.randn(40, F_ + 109, generator=g), torch.randn(40, generator=g)
    cols = hmr_state_columns(F_)
    assert eq(pack_linear(w, b, "cpu", col_index=cols, k_total=F_ + 112), R.pack_linear(w, b, "cpu", col_index=cols, k_total=F_ + 112))
    rows = list(range(96)) + [96 + i for i in range(10)] + [108 + i for i in range(3)]
    wd = torch.randn(109, 32, generator=g)
    assert eq(pack_linear(wd, None, "cpu", row_index=rows, n_total=112), R.pack_linear(wd, None, "cpu", row_index=rows, n_total=112))
    assert eq(pack_linear(w, b, "cpu", n_total=44), R.pack_linear(w, b, "cpu", n_total=44))
    # two-source 1x1 (conv3 + downsample): [W0 | W1], bias summed in fp64
    w0, w1 = torch.randn(256, 64, 1, 1, generator=g).double(), torch.randn(256, 128, 1, 1, generator=g).double()
    b0, b1 = torch.randn(256, generator=g).double(), torch.randn(256, generator=g).double()
    assert eq(pack_conv1x1_dual(w0, b0, w1, b1, "cpu"), R.pack_conv(torch.cat([w0, w1], 1), b0 + b1, 1, 0, "cpu"))
    # MANO constants: blend matrix bit-exact; the J_regressor contractions are fp64 sums of 778 terms whose
    # order differs from numpy's dgemm -> equal up to one fp32 rounding
    a = hands_amd.synthetic_mano_asset(False)
    m, r = pack_mano(a, "cpu"), R.pack_mano(a, "cpu")
    assert eq(m["blend"], r["blend"])
    for k in ("pose_mean", "lbs_weights", "tip_ids", "faces"):
        assert torch.equal(m[k], r[k]), k
    for k in ("J_template", "J_shapedirs"):
        assert m[k].shape == r[k].shape and (m[k] - r[k]).abs().max() <= 2 ** -23 * r[k].abs().max(), k


def test_workspace_query_matches_the_split_policy():
    L = _lib.lib()
    d = _lib.ConvDesc(256, 1, 1, 2304, 1, 1, 2048, 1, 1, 1, 0, 2304, 2048, 0, 2304, 1)    # feature_conv Linear at bz=128
    S = L.hands_conv2d_splitk_factor(ctypes.byref(d))
    assert S == 8 and L.hands_conv2d_workspace_floats(ctypes.byref(d), 0) == 8 * 256 * 2048
    assert L.hands_conv2d_workspace_floats(ctypes.byref(d), 3) == 3 * 256 * 2048
    assert L.hands_conv2d_workspace_floats(ctypes.byref(d), 1) == 0
    d2 = _lib.ConvDesc(4, 56, 56, 64, 56, 56, 64, 3, 3, 1, 1, 64, 64, 0, 576, 1)           # a trunk conv: never split by policy
    assert L.hands_conv2d_workspace_floats(ctypes.byref(d2), 0) == 0
    assert L.hands_conv2d_workspace_floats(None, 2) == -1


def test_bad_descriptors_are_rejected_without_a_gpu():
    L = _lib.lib()
    d = _lib.ConvDesc(1, 7, 7, 30, 7, 7, 64, 1, 1, 1, 0, 30, 64, 0, 32, 0)      # Cin % 4 != 0
    assert L.hands_conv2d_nhwc_f32(ctypes.byref(d), 16, 16, 16, None, 16, None) == 10001
    assert L.hands_conv2d_nhwc_f32(None, 16, 16, 16, None, 16, None) == 10001
    assert L.hands_maxpool3x3s2_nhwc_f32(16, 16, 1, 8, 8, 6, None) == 10001
    # output map larger than the (H, W, stride, pad) geometry allows -> would read outside the input
    d = _lib.ConvDesc(1, 7, 7, 16, 9, 9, 64, 1, 1, 1, 0, 16, 64, 0, 16, 0)
    assert L.hands_conv2d_nhwc_f32(ctypes.byref(d), 16, 16, 16, None, 16, None) == 10001
    d = _lib.ConvDesc(1, 7, 7, 16, 4, 4, 64, 1, 1, 2, 0, 16, 64, 0, 16, 0)       # stride 2: 4x4 is legal ...
    d5 = _lib.ConvDesc(1, 7, 7, 16, 5, 5, 64, 1, 1, 2, 0, 16, 64, 0, 16, 0)      # ... 5x5 is not
    assert L.hands_conv2d_nhwc_splitk_n_f32(ctypes.byref(d5), 16, 16, 16, None, 16, 2, 16, 1 << 20, None) == 10001
    d = _lib.ConvDesc(70000, 224, 224, 16, 224, 224, 64, 1, 1, 1, 0, 16, 64, 0, 16, 0)   # > 2^31 elements
    assert L.hands_conv2d_nhwc_f32(ctypes.byref(d), 16, 16, 16, None, 16, None) == 10001
    with pytest.raises(RuntimeError):
        _lib.check(10001, "x")


def test_product_refuses_cpu_tensors(recipe_model):
    inputs, meta = hands_amd.synthetic_inputs(1, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        recipe_model(inputs, meta)


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "hands_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle/", ""), f
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_unsupported_switches_fail_loudly():
    a = dict(hands_amd.DEFAULT_ARGS)
    a["tf_decoder"] = True
    with pytest.raises(NotImplementedError):
        hands_amd.HandsLight(args=type(hands_amd.DEFAULT_ARGS)(a))
    with pytest.raises(NotImplementedError):
        hands_amd.HandsLight(backbone="resnet18")
    for bad in (dict(pos_enc="no_such_encoding"), dict(use_render_seg_loss=True), dict(use_depth_loss=True, no_crops=True),
                dict(use_glb_feat=False), dict(regress_center_corner=True, no_crops=True)):
        with pytest.raises(NotImplementedError):
            hands_amd.HandsLight(args=type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS, **bad)))
    # built non-default switches construct (parity: tests/test_oracle_golden.py, tests/test_gpu_parity.py)
    m = hands_amd.HandsLight(args=type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS, pos_enc="center+corner")))
    assert m.hand_backbone.conv1.in_channels == 3 + 80 and m.feature_conv[0].in_channels == 2048
    m = hands_amd.HandsLight(args=type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS, pos_enc="cam_conv", use_depth_loss=True)))
    assert m.feature_conv[0].in_channels == 2048 + 6 and m.depth_mlp[0].in_channels == 2048 + 6 + 2 and m.depth_mlp[17].out_channels == 1


def test_state_dict_roundtrip_and_wrapper_prefix(recipe_model):
    m2 = hands_amd.HandsLight()
    ck = {"model." + k: v for k, v in recipe_model.state_dict().items()}         # Lightning-style checkpoint
    missing = m2.load_state_dict({k[len("model."):]: v for k, v in ck.items()}, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert torch.equal(m2.state_dict()["head_r.hmr_layer.decoders.cam_t/wp.weight"],
                       recipe_model.state_dict()["head_r.hmr_layer.decoders.cam_t/wp.weight"])
    assert m2._packed is None
    assert recipe_model.mano_r.faces.shape == (1538, 3) and recipe_model.mano_r.faces.dtype == np.int64
    assert np.array_equal(recipe_model.mano_l.faces, hands_amd.synthetic_mano_asset(False).faces)   # bit-exact pass-through


# ---- sharding helpers ----------------------------------------------------------------------------
def test_shard_ranges_cover_the_batch():
    for bz in (1, 7, 8, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(bz, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == bz
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_shard_and_pack_roundtrip():
    inputs, meta = hands_amd.synthetic_inputs(6, 0, img_res=8)
    meta["imgname"] = [f"{i}.jpg" for i in range(6)]
    parts = [shard_batch(inputs, meta, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p[0]["r_img"] for p in parts]), inputs["r_img"])
    assert sum((p[1]["imgname"] for p in parts), []) == meta["imgname"]
    out = {"a": torch.randn(3, 778, 3), "b": torch.randn(3, 16, 3, 3), "c": torch.randn(3, 9)}
    flat, layout = pack_predictions(out)
    back = unpack_predictions(flat, layout)
    assert flat.shape == (3, 778 * 3 + 144 + 9) and all(torch.equal(back[k], out[k]) for k in out)


def test_load_reference_style_checkpoint(tmp_path, recipe_model):
    from hands_amd.checkpoint import load_reference_checkpoint
    sd = {"model." + k: v.clone() for k, v in recipe_model.state_dict().items()}
    sd["mano_r.shapedirs"] = torch.zeros(778, 3, 10)            # wrapper-level buffers are ignored
    sd["model.mano_r.mano.unknown_smplx_buffer"] = torch.zeros(3)
    path = str(tmp_path / "last.ckpt")
    torch.save({"state_dict": sd, "epoch": 3}, path)
    m = hands_amd.HandsLight()
    rep = load_reference_checkpoint(m, path)
    assert rep.missing_keys == [] and rep.unexpected_keys == ["mano_r.mano.unknown_smplx_buffer"]
    for k, v in recipe_model.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    # a wrong-architecture checkpoint raises like the reference's load_state_dict (size mismatch), a
    # MANO buffer of another shape is skipped, and a checkpoint without a single matching key raises too
    wrong = dict(sd)
    wrong["model.feature_conv.0.weight"] = torch.zeros(1024, 2048, 1, 1)
    with pytest.raises(RuntimeError, match="size mismatch"):
        load_reference_checkpoint(hands_amd.HandsLight(), {"state_dict": wrong})
    pca = dict(sd)
    pca["model.mano_r.mano.shapedirs"] = torch.zeros(778, 3, 45)
    m3 = hands_amd.HandsLight()
    load_reference_checkpoint(m3, {"state_dict": pca})
    assert torch.equal(m3.mano_r.mano.shapedirs, recipe_model.mano_r.mano.shapedirs)
    with pytest.raises(RuntimeError, match="no key"):
        load_reference_checkpoint(hands_amd.HandsLight(), {"state_dict": {"model.foo": torch.zeros(1)}})


def test_graphed_forward_and_frontend_refuse_cpu():
    import torch
    import hands_amd
    with pytest.raises(RuntimeError):
        hands_amd.GraphedForward(None, {"img": torch.zeros(1, 3, 8, 8)}, {})
    with pytest.raises(RuntimeError):
        hands_amd.HandsFrontEnd().boxes(torch.zeros(1, 21, 2), torch.zeros(1, 21, 2), torch.eye(3)[None])


# ---- real-asset loader (f3): chumpy-free MANO pickle reader -----------------------------------------
from fake_mano import write_fake_mano_pkl as _write_fake_mano_pkl  # noqa: E402


@pytest.mark.parametrize("is_rhand", [True, False])
def test_load_mano_pkl_without_chumpy(tmp_path, monkeypatch, is_rhand):
    """hands_amd.mano.load_mano_pkl replaces smplx.MANO(model_path=$MANO_DIR, use_pca=False,
    flat_hand_mean=False) (common/body_models.py:90-99): reads the chumpy / scipy-sparse pickle with
    neither chumpy nor smplx installed and yields exactly the arrays the LBS kernels consume."""
    import sys
    from hands_amd.mano import build_mano_asset, load_mano_pkl, ManoLayer, SMPLX_MANO_BUFFERS
    assert "chumpy" not in sys.modules
    ref = hands_amd.synthetic_mano_asset(is_rhand)
    fn = tmp_path / ("MANO_RIGHT.pkl" if is_rhand else "MANO_LEFT.pkl")
    _write_fake_mano_pkl(str(fn), ref)
    assert b"chumpy" in open(fn, "rb").read()
    got = load_mano_pkl(str(fn), is_rhand)
    assert "chumpy" not in sys.modules
    for name in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "hands_mean"):
        a, b = getattr(got, name), getattr(ref, name)
        assert a.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b), name
    assert got.faces.dtype == np.int64 and np.array_equal(got.faces, ref.faces)          # bit-exact face indices
    assert got.is_rhand == is_rhand
    # $MANO_DIR route (what the constructors use) + the smplx buffer names on the module
    monkeypatch.setenv("MANO_DIR", str(tmp_path))
    monkeypatch.delenv("HANDS_SYNTHETIC_MANO", raising=False)
    via_env = build_mano_asset(is_rhand)
    assert np.array_equal(via_env.posedirs, ref.posedirs)
    layer = ManoLayer(via_env)
    assert tuple(layer.state_dict().keys()) == SMPLX_MANO_BUFFERS
    assert layer.state_dict()["posedirs"].shape == (135, 2334) and layer.state_dict()["faces_tensor"].dtype == torch.int64
    assert torch.equal(layer.state_dict()["pose_mean"][3:], layer.state_dict()["hand_mean"])
    back = layer.asset()
    assert np.array_equal(back.hands_mean, ref.hands_mean) and np.array_equal(back.faces, ref.faces)
    # the other side's file is missing and the synthetic stand-in was not asked for: fail like the reference
    with pytest.raises(FileNotFoundError, match="MANO"):
        build_mano_asset(not is_rhand)
    assert build_mano_asset(not is_rhand, allow_synthetic=True).v_template.shape == (778, 3)


def test_missing_assets_fail_loudly_without_opt_in(monkeypatch):
    monkeypatch.delenv("HANDS_SYNTHETIC_MANO", raising=False)
    monkeypatch.delenv("MANO_DIR", raising=False)
    monkeypatch.delenv("DATA_DIR", raising=False)
    with pytest.raises(FileNotFoundError):
        hands_amd.HandsLight()
    from hands_amd.hamer import load_mano_mean_params
    with pytest.raises(FileNotFoundError):
        load_mano_mean_params()
    assert load_mano_mean_params(allow_synthetic=True)["pose"].shape == (96,)


def test_smplx_named_mano_buffers_load_from_a_reference_checkpoint(recipe_model):
    """A reference checkpoint carries `mano_{r,l}.mano.<smplx buffer>` entries; same-named, same-shaped
    buffers override the module's asset, smplx's extra parameters are reported as unexpected."""
    from hands_amd.checkpoint import load_reference_checkpoint
    from hands_amd.mano import SMPLX_MANO_BUFFERS
    m = hands_amd.HandsLight()
    own = {k for k in m.state_dict() if k.startswith("mano_r.mano.")}
    assert own == {"mano_r.mano." + b for b in SMPLX_MANO_BUFFERS}
    sd = {"model." + k: v.clone() for k, v in recipe_model.state_dict().items()}
    sd["model.mano_r.mano.v_template"] = torch.full((778, 3), 0.5)
    sd["model.mano_r.mano.betas"] = torch.zeros(1, 10)                       # smplx nn.Parameter
    sd["model.mano_r.mano.vertex_joint_selector.extra_joints_idxs"] = torch.tensor([744, 320, 443, 554, 671])
    rep = load_reference_checkpoint(m, {"state_dict": sd})
    assert sorted(rep.unexpected_keys) == ["mano_r.mano.betas", "mano_r.mano.vertex_joint_selector.extra_joints_idxs"]
    assert torch.all(m.mano_r.mano.v_template == 0.5) and m._packed is None
    assert np.all(m.mano_r.mano.asset().v_template == 0.5)


def test_hamer_pretrained_argument_has_the_reference_semantics(tmp_path, monkeypatch):
    """src/models/hamer_light/model.py:33-44: ``args.get('pretrained', 'vit')``.  'vit' loads
    $DATA_DIR/hamer_training_data/vitpose_backbone.pth['state_dict'] into ``backbone`` with strict=False,
    'hamer' splits hamer.ckpt into ``backbone.`` / ``mano_head.`` and loads both strictly, a missing file
    raises, anything else ('none') keeps the initial weights.  The synthetic checkpoints store every tensor as
    an expanded one-element storage."""
    import hands_amd.hamer as hamer_mod
    from hands_amd.hamer import HAMER, _Args
    monkeypatch.setattr(hamer_mod, "VIT_DEPTH", 3)          # the loader does not depend on the depth; keeps the test in seconds
    mk = lambda **kw: _Args(pos_enc="center+corner_latent", n_freq_pos_enc=4, use_grasp_loss=True,
                            use_render_seg_loss=False, **kw)
    monkeypatch.setenv("DATA_DIR", str(tmp_path))
    base = HAMER(mk(pretrained="none"))
    fill = lambda t, v: torch.full((1,), v, dtype=t.dtype).expand(t.shape)
    # -- a missing file raises (reference: torch.load fails), also when the key is absent (default 'vit')
    with pytest.raises(FileNotFoundError):
        HAMER(mk())
    with pytest.raises(FileNotFoundError):
        HAMER(mk(pretrained="hamer"))
    monkeypatch.delenv("DATA_DIR")
    with pytest.raises(KeyError):
        HAMER(mk(pretrained="vit"))
    monkeypatch.setenv("DATA_DIR", str(tmp_path))
    # -- 'vit': ViTPose backbone file, strict=False: a subset of the keys + keys the model does not have
    os.makedirs(tmp_path / "hamer_training_data")
    vit_sd = {k: fill(v, 0.25) for k, v in base.backbone.state_dict().items() if not k.startswith("blocks.2.")}
    vit_sd["keypoint_head.final_layer.weight"] = torch.zeros(17, 8)
    torch.save({"state_dict": vit_sd, "meta": {"epoch": 210}}, tmp_path / "hamer_training_data" / "vitpose_backbone.pth")
    m = HAMER(mk())                                         # no 'pretrained' key -> 'vit'
    got = m.backbone.state_dict()
    assert torch.all(got["blocks.0.attn.qkv.weight"] == 0.25) and torch.all(got["pos_embed"] == 0.25)
    assert not torch.any(got["blocks.2.mlp.fc2.weight"] == 0.25)          # absent from the file: initial weights stay
    assert not torch.all(m.mano_head.decpose.weight == 0.25) and m._packed is None
    # -- 'hamer': strict on both halves
    ck = tmp_path / "hamer" / "_DATA" / "hamer_ckpts" / "checkpoints"
    os.makedirs(ck)
    sd = {"backbone." + k: fill(v, 0.5) for k, v in base.backbone.state_dict().items()}
    sd.update({"mano_head." + k: fill(v, 0.125) for k, v in base.mano_head.state_dict().items()})
    sd["discriminator.D_shape.weight"] = torch.zeros(3)     # neither half: ignored, as in the reference
    torch.save({"state_dict": sd}, ck / "hamer.ckpt")
    m = HAMER(mk(pretrained="hamer"))
    assert all(torch.all(v == 0.5) for v in m.backbone.state_dict().values())
    assert all(torch.all(v == 0.125) for v in m.mano_head.state_dict().values())
    assert not torch.all(m.kpe.feat_mlp[0].weight == 0.5)
    del sd["backbone.blocks.1.norm1.bias"]
    torch.save({"state_dict": sd}, ck / "hamer.ckpt")
    with pytest.raises(RuntimeError, match="blocks.1.norm1.bias"):
        HAMER(mk(pretrained="hamer"))
    sd["backbone.blocks.1.norm1.bias"] = torch.zeros(1280)
    sd["mano_head.extra.weight"] = torch.zeros(1)
    torch.save({"state_dict": sd}, ck / "hamer.ckpt")
    with pytest.raises(RuntimeError, match="extra.weight"):
        HAMER(mk(pretrained="hamer"))


def test_bench_launcher_stays_gpu_free_and_reports_the_failing_rank(monkeypatch, tmp_path):
    """``python bench.py --gpus N`` (WORLD_SIZE unset) is a pure launcher -- it must not import torch
    or touch the HIP runtime before (or after) starting its children; the GPU count comes from the *_VISIBLE_DEVICES lists
    / sysfs.  A rank that dies is reported with the tail of ITS stderr.  Here (no GPU) every rank fails at
    ``torch.cuda.set_device``: exactly the failure path."""
    import importlib.util
    import inspect
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    src = inspect.getsource(bench.launch_ranks) + inspect.getsource(bench.visible_gpu_count)
    assert "import torch" not in src and "torch." not in src
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and "only 0 GPU(s) are visible" in p.stderr
    env["HANDS_BENCH_SHARE_GPU"] = "1"           # skips the count check: the ranks start and fail on their own
    env["HANDS_BENCH_BACKEND"] = "gloo"
    probe = tmp_path / "sitecustomize.py"        # proves the launcher process itself never imported torch
    probe.write_text("import atexit, os, sys\n"
                     "if 'WORLD_SIZE' not in os.environ:\n"
                     "    atexit.register(lambda: sys.stderr.write('LAUNCHER_TORCH=%s\\n' % ('torch' in sys.modules)))\n")
    env["PYTHONPATH"] = str(tmp_path) + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode not in (0, 2), p.stderr[-1500:]
    assert "LAUNCHER_TORCH=False" in p.stderr
    assert "bench.py launcher: rank" in p.stderr and "stderr (tail)" in p.stderr and "Traceback" in p.stderr


def test_winograd_packer_layout_and_identity():
    """hands_pack_conv3x3_winograd_f64 (csrc/pack.cpp): U = G g G^T in fp64, rounded once, in the MFMA-A operand order
    [Cout/32][Cin/8][xi][nu][lane 64][4] that csrc/conv_wino.hip reads.  Checked against an independent numpy restatement
    and, through A^T [U (.) B^T d B] A evaluated in fp64 FROM THE PACKED BUFFER, against F.conv2d."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(9)
    Cout, Cin = 64, 32
    w = torch.randn(Cout, Cin, 3, 3, generator=g).double()
    pc = pack_conv(w, torch.randn(Cout, generator=g), 1, 1, "cpu")
    assert pc.wino is not None and pc.wino.dtype == torch.float32 and pc.wino.numel() == 16 * Cout * Cin
    assert pack_conv(w, None, 2, 1, "cpu").wino is None and pack_conv(w[:, :, :1, :1], None, 1, 0, "cpu").wino is None
    assert pack_conv(w[:24], None, 1, 1, "cpu").wino is None          # Cout % 32 != 0: the direct kernel keeps the layer
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    U = np.einsum("xi,ocij,nj->xnoc", G, w.numpy(), G)
    up = pc.wino.numpy().reshape(Cout // 32, Cin // 8, 4, 4, 2, 32, 4)      # nb, c8, xi, nu, half, o, e
    back = np.transpose(up, (2, 3, 0, 5, 1, 4, 6)).reshape(4, 4, Cout, Cin)     # xi, nu, (nb, o), (c8, half, e)
    assert np.array_equal(back, U.astype(np.float32))
    # the identity, from the packed weights
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    x = torch.randn(2, Cin, 6, 8, generator=g).double()
    xp = F.pad(x, (1, 1, 1, 1)).numpy()
    y = np.zeros((2, Cout, 6, 8))
    for ty in range(3):
        for tx in range(4):
            d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            V = np.einsum("xa,bcae,ne->bcxn", Bt, d, Bt)
            M = np.einsum("bcxn,xnoc->boxn", V, back.astype(np.float64))
            y[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ix,boxn,jn->boij", At, M, At)
    ref = F.conv2d(x, w, padding=1).numpy()
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max()           # U is rounded to fp32 once


def test_library_was_built_from_the_sources_in_the_tree():
    """hands_csrc_sha16() (embedded by csrc/Makefile) == the same hash over the tree (bench.csrc_tree_sha16): the in-tree .so is not
    stale.  bench.py validates stored counter summaries against the LOADED library's hash, so the two must agree for a
    summary taken on this tree to be usable."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod_hash", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    L = _lib.lib()
    assert L.hands_csrc_sha16().decode() == bench.csrc_tree_sha16() == bench.csrc_sha16()
