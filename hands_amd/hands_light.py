"""``HandsLight`` -- the WildHands forward path on MI355X, behind the reference's module API.

Same constructor, ``forward(inputs, meta_info) -> xdict`` contract, 22 output keys and
``state_dict`` names as the reference model (src/models/hands_light/model.py:15-437), but every
statement of ``forward`` runs as a hand-written gfx950 kernel from ``libhands_hip.so``:

    statement (model.py)                      kernel (include/hands_hip.h)
    :193,238-239  ResNet-50 trunks            hands_stem_conv_maxpool_nchw_f32 (conv1 + bn1 + relu + max-pool from the NCHW image),
                                              hands_conv2d_nhwc_f32 / hands_conv1x1_dual_nhwc_f32 (fp32 MFMA implicit GEMM, BN
                                              folded, bias / residual / ReLU epilogue), hands_conv3x3_winograd4_f32 (the 3x3 /
                                              stride-1 layers as Winograd F(4x4,3x3); hands_conv3x3_winograd_f32 = F(2x2) fallback)
    :196          sum-pool                    hands_sumpool_nhwc_f32
    :258-271      KPE + concat                hands_kpe_concat_f32
    :313-314      feature_conv                hands_conv2d_nhwc_f32 x4
    :320-321      HandHMR x2                  hands_conv2d_nhwc_f32 (cam_init, 3 x refine+decoders),
                                              hands_hmr_init_f32, hands_rot6d_to_matrix_f32
    :341-368      is_flipped swap             hands_flip_swap_f32 (per sample, no host sync)
    :378-390      MANOHead x2                 hands_mano_heads_f32 (both hands, one launch; the three-launch chain
                                              hands_mano_pose_f32 -> blend GEMM -> hands_mano_skin_f32 is its cross-check)
    :401-404      grasp classifier            hands_grasp_input_f32, hands_conv2d_nhwc_f32 x4

torch is used for parameter containers, device buffers and streams only.  Built configurations: resnet50,
tf_decoder=False with every pos_enc of model.py -- 'center+corner_latent' (shipped default), 'sinusoidal_cc', 'center', 'corner',
'center+corner', 'dense', 'dense_latent', 'cam_conv', 'pcl', 'perspective_correction', None -- ``no_crops`` (arctic_light), the
grasp head with / without the global feature vector or absent, ``separate_hands``, ``regress_center_corner``,
``use_glb_feat=False``, ``use_depth_loss`` (the depth head); tf_decoder, the ViT backbone and the renderer raise
``NotImplementedError``.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import ManoConsts, ManoOut, ManoSide, check, ptr
from .engine import DEFAULT_ENGINE, ConvEngine, EngineSwitches
from .mano import ManoLayer, build_mano_asset
from .packing import (HMR_VEC, PackedConv, fold_bn, hmr_state_columns, pack_conv, pack_conv1x1_dual,
                      pack_linear, pack_mano)
from .xdict import prefix_dict, stream_xdict, xdict

RESNET50_LAYERS = (3, 4, 6, 3)


# --------------------------------------------------------------------------------------------------
# parameter containers (names = the reference's state_dict keys; never called)
# --------------------------------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    """resnet.py:99-154 parameter layout (conv1/bn1/conv2/bn2/conv3/bn3/downsample.{0,1})."""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.stride = stride
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None


class ResNet50Params(nn.Module):
    """resnet.py:157-280 parameter layout of the ResNet-50 trunk (no avgpool/fc)."""

    def __init__(self, in_ch=3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_ch, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), RESNET50_LAYERS), start=1):
            blocks = []
            for bi in range(n):
                stride = 2 if (bi == 0 and li > 1) else 1
                blocks.append(_Bottleneck(inplanes, planes, stride, downsample=(bi == 0)))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))


class _HMRLayerParams(nn.Module):
    """hmr_layer.py:7-62 (tf_decoder=False)."""

    def __init__(self, feat_dim, mid_dim, specs):
        super().__init__()
        hmr_dim = feat_dim + sum(specs.values())
        self.refine = nn.Sequential(nn.Linear(hmr_dim, mid_dim), nn.ReLU(), nn.Dropout(),
                                    nn.Linear(mid_dim, mid_dim), nn.ReLU(), nn.Dropout())
        self.decoders = nn.ModuleDict({k: nn.Linear(mid_dim, v) for k, v in specs.items()})


class HandHMR(nn.Module):
    """hand_hmr.py:9-44 parameter layout."""

    def __init__(self, feat_dim, is_rhand, n_iter):
        super().__init__()
        self.is_rhand, self.n_iter = is_rhand, n_iter
        self.hand_specs = {"pose_6d": 96, "cam_t/wp": 3, "shape": 10}
        self.hmr_layer = _HMRLayerParams(feat_dim, 1024, self.hand_specs)
        self.cam_init = nn.Sequential(nn.Linear(feat_dim, 512), nn.ReLU(), nn.Linear(512, 512), nn.ReLU(),
                                      nn.Linear(512, 3))


class MANOHead(nn.Module):
    """mano_head.py:12-19: holds the MANO layer as ``.mano``."""

    def __init__(self, is_rhand, focal_length, img_res, asset=None):
        super().__init__()
        self.mano = ManoLayer(asset if asset is not None else build_mano_asset(is_rhand))
        self.focal_length, self.img_res, self.is_rhand = focal_length, img_res, is_rhand

    @property
    def faces(self):
        return self.mano.faces


_MANO_OUT_SHAPES = (("vertices", (778, 3)), ("v3d.cam", (778, 3)), ("joints3d", (21, 3)), ("j3d.cam", (21, 3)),
                    ("j2d.norm", (21, 2)), ("cam_t", (3,)))


def _mano_out_buffers(bz, dev):
    """The twelve output tensors of the two MANO heads as contiguous views of ONE allocation (one allocator call per
    forward); the two vertex arrays come first in each side's 16-byte aligned block (the kernel stores them 8 bytes wide)."""
    side_floats = (bz * 4839 + 3) // 4 * 4
    flat = torch.empty(2 * side_floats, device=dev)
    outs = []
    for side in range(2):
        o, off = {}, side * side_floats
        for name, shp in _MANO_OUT_SHAPES:
            n = bz
            for d_ in shp:
                n *= d_
            o[name] = flat[off:off + n].view((bz,) + shp)
            off += n
        outs.append(o)
    return outs


def _mano_output_dict(outs, rot, shape, cam, cam_init, bz):
    """Key order of mano_head.py:53-61 + the `cam_t.wp.init` / `mano.` prefixing of model.py:392-399."""
    output = xdict()
    for side, post in enumerate((".r", ".l")):
        ro = side * bz
        o = outs[side]
        md = xdict()
        md["cam_t.wp"] = cam[ro:ro + bz]
        md["cam_t"] = o["cam_t"]
        md["joints3d"] = o["joints3d"]
        md["vertices"] = o["vertices"]
        md["j3d.cam"] = o["j3d.cam"]
        md["v3d.cam"] = o["v3d.cam"]
        md["j2d.norm"] = o["j2d.norm"]
        md["beta"] = shape[ro:ro + bz]
        md["pose"] = rot[ro:ro + bz]
        md = md.postfix(post)
        md["cam_t.wp.init" + post] = cam_init[ro:ro + bz]       # model.py:392-393
        output.merge(prefix_dict(md, "mano."))                  # model.py:395-399
    return output


class ManoHeadsPlan:
    """``MANOHead.forward`` of both hands as ONE pre-bound C-ABI call (BASELINE configs[4], a serving loop over fixed
    buffers): the descriptor array of ``hands_mano_heads_f32`` -- input pointers, the twelve output views, both assets'
    constants -- is built once; ``launch(stream)`` is the single ctypes call per step and ``outputs`` the (fixed) result
    dict.  The caller owns ``rot`` (2bz,16,3,3) [or (2bz,48) axis-angle with ``aa_input``], ``shape`` (2bz,10),
    ``cam`` (2bz,3), ``K`` (bz,3,3) and rewrites them in place between steps."""

    def __init__(self, L, mano_r, mano_l, rot, shape, cam, K, img_res, bz, aa_input=False, cam_init=None):
        per = 48 if aa_input else 144
        assert rot.is_contiguous() and shape.is_contiguous() and cam.is_contiguous() and K.is_contiguous()
        assert rot.numel() == 2 * bz * per and shape.numel() == 2 * bz * 10 and cam.numel() == 2 * bz * 3 and K.numel() == bz * 9
        self.L, self.bz, self.img_res, self.aa = L, bz, float(img_res), int(bool(aa_input))
        self._keep = (mano_r, mano_l, rot, shape, cam, K)
        self._outs = _mano_out_buffers(bz, rot.device)
        self._sides = (ManoSide * 2)()
        for side, mp in enumerate((mano_r, mano_l)):
            ro, o = side * bz, self._outs[side]
            mo = ManoOut(ptr(o["vertices"]), ptr(o["joints3d"]), ptr(o["v3d.cam"]), ptr(o["j3d.cam"]), ptr(o["j2d.norm"]),
                         ptr(o["cam_t"]))
            self._sides[side] = ManoSide(mp["consts"], ptr(mp["blend"].w), ptr(mp["blend"].bias), ptr(rot, ro * per),
                                         ptr(shape, ro * 10), ptr(cam, ro * 3), mo)
        self._K = ptr(K)
        self.outputs = _mano_output_dict(self._outs, rot, shape, cam, cam if cam_init is None else cam_init, bz)

    def launch(self, stream):
        check(self.L.hands_mano_heads_f32(self._sides, 2, self._K, 10, self.img_res, 0.1, self.bz, self.aa, stream), "mano_heads")
        return self.outputs


def run_mano_heads(L, mano_r, mano_l, rot, shape, cam, cam_init, K, img_res, bz, stream, buf, engine=None):
    """MANOHead.forward for the right (rows [0,bz)) and left (rows [bz,2bz)) hands
    (src/nets/hand_heads/mano_head.py:21-65) + the `cam_t.wp.init` / `mano.` prefixing of
    model.py:392-399.  rot (2bz,16,3,3), shape (2bz,10), cam / cam_init (2bz,3), K (bz,3,3).
    One launch for both hands (hands_mano_heads_f32); ``engine.fuse_mano = False`` keeps the three-launch
    chain per side (pose -> blend GEMM -> skin), which the tests compare it with."""
    dev = rot.device
    engine = engine or DEFAULT_ENGINE
    fused = getattr(engine, "fuse_mano", True)
    outs = _mano_out_buffers(bz, dev)
    mouts = [ManoOut(ptr(o["vertices"]), ptr(o["joints3d"]), ptr(o["v3d.cam"]), ptr(o["j3d.cam"]),
                     ptr(o["j2d.norm"]), ptr(o["cam_t"])) for o in outs]
    if fused:
        sides = (ManoSide * 2)()
        for side, mp in enumerate((mano_r, mano_l)):
            ro = side * bz
            sides[side] = ManoSide(mp["consts"], ptr(mp["blend"].w), ptr(mp["blend"].bias), ptr(rot, ro * 144),
                                   ptr(shape, ro * 10), ptr(cam, ro * 3), mouts[side])
        check(L.hands_mano_heads_f32(sides, 2, ptr(K), 10, img_res, 0.1, bz, 0, stream), "mano_heads")
    else:
        blend_in = buf("blend_in", bz * 160)
        Abuf, j16 = buf("mano_A", bz * 192), buf("mano_j16", bz * 48)
        vposed = buf("vposed", bz * 2336)
        for side, mp in enumerate((mano_r, mano_l)):
            ro = side * bz
            check(L.hands_mano_pose_f32(C.byref(mp["consts"]), ptr(rot, ro * 144), ptr(shape, ro * 10), 10,
                                        ptr(blend_in), 160, ptr(Abuf), ptr(j16), bz, stream), "mano_pose")
            engine.conv(L, mp["blend"], blend_in, bz, 1, 1, vposed, False, stream)
            check(L.hands_mano_skin_f32(C.byref(mp["consts"]), ptr(vposed), 2336, ptr(Abuf), ptr(j16),
                                        ptr(cam, ro * 3), ptr(K), img_res, 0.1, C.byref(mouts[side]), bz, stream),
                  "mano_skin")
    return _mano_output_dict(outs, rot, shape, cam, cam_init, bz)


run_mano_heads.launches_per_step = 1


def mano_consts(m):
    return ManoConsts(ptr(m["pose_mean"]), ptr(m["J_template"]), ptr(m["J_shapedirs"]), ptr(m["lbs_weights"]),
                      ptr(m["tip_ids"]))


class _Args(dict):
    """attribute *and* ``.get`` access, like the reference's EasyDict args (parser.py:39-58)."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return self.get(k)


LATENT_ENC = ("center+corner_latent", "sinusoidal_cc")
IMAGE_ENC = {"center": 1, "corner": 2, "center+corner": 3}     # -> mode of hands_image_posenc_nhwc_f32

DEFAULT_ARGS = _Args(backbone="resnet50", pos_enc="center+corner_latent", n_freq_pos_enc=4,
                     use_glb_feat=True, separate_hands=False, tf_decoder=False, use_grasp_loss=True,
                     use_glb_feat_w_grasp=True, no_crops=False, use_depth_loss=False,
                     regress_center_corner=False, use_render_seg_loss=False, img_res=224,
                     focal_length=1000.0)


# --------------------------------------------------------------------------------------------------
class _PackedHolder:
    """Packed (kernel-layout) weights of one parameter set, shared by a model and its replicas: any of them
    invalidating it (load_state_dict, .to(), invalidate_packed) makes all of them repack on the next forward."""

    def __init__(self):
        self.packed, self.dev, self.version = None, None, 0

    def invalidate(self):
        self.packed = None
        self.version += 1


class HandsLight(EngineSwitches, nn.Module):
    def __init__(self, backbone="resnet50", focal_length=1000.0, img_res=224, args=None,
                 mano_assets=None):
        super().__init__()
        self.engine = ConvEngine()
        # Winograd F(4x4,3x3) (csrc/conv_wino4.hip, 2.25 multiplications per output) for the stride-1 3x3 convolutions of these
        # ResNet stages; the others keep F(2x2,3x3).  Round 5, one box, alternating: alone on the chip a launch takes the time of its
        # F(2x2) twin (layers 1-3) or 9 % less (layer 4), but it spends ~45 % fewer matrix-pipe cycles, and in the shipped multi-stream
        # mode that is +1.6 % on the forward with all four stages (+0.4 % with stage 4 only); end-to-end error over 200 random inputs
        # unchanged (median 2.4e-7 m, max 4.4e-7 against 2.5e-7 / 4.9e-7 with F(2x2): tools/hl_parity_sweep.py).  A packing-time
        # choice (36 Cout Cin floats per layer, 130 MB per trunk): call invalidate_packed() after changing it; engine.winograd4 = False
        # falls back to F(2x2) without repacking.
        self.winograd4_stages = (1, 2, 3, 4)
        self.engine.winograd4 = True
        self.trunk_chunks = (1, 2)    # (global, hand) trunk jobs, one HIP stream each
        self.async_tail = True        # tail of the forward on its own stream, joined at first use of the result
        self._calls = 0
        args = args if args is not None else DEFAULT_ARGS
        get = (lambda k, d=None: args.get(k, d)) if hasattr(args, "get") else (lambda k, d=None: getattr(args, k, d))
        self.args = args
        if backbone != "resnet50":
            raise NotImplementedError("hands_amd.HandsLight: only backbone='resnet50' is built")
        # Configuration switches (model.py:33-157).  Built: the latent KPE ('center+corner_latent', and 'sinusoidal_cc' whose
        # forward is the same code, model.py:258-271 / 288-304), the image-level encodings 'center' / 'corner' /
        # 'center+corner' (extra input channels of the hand trunk's conv1, model.py:60-77, 203-218), no encoding (None),
        # `no_crops` (both heads read the pooled global features, model.py:199-201, 316-318: the arctic_light configuration),
        # the grasp head with and without the global feature vector, or absent.
        pos_enc = get("pos_enc")
        self.pos_enc = pos_enc
        # 'dense' (per-pixel angle maps as extra conv1 channels, model.py:220-224), 'dense_latent' / 'cam_conv' (the maps resized
        # to the 7x7 feature map and concatenated, :244-256 / :276-288), 'pcl' / 'perspective_correction' (no encoding; the global
        # rotation is corrected after the heads, :330-334 / :370-376)
        self.enc_mode = ("latent" if pos_enc in LATENT_ENC else "image" if pos_enc in IMAGE_ENC else
                         "dense" if pos_enc == "dense" else "dense_latent" if pos_enc in ("dense_latent", "cam_conv") else
                         "none" if pos_enc in (None, "pcl", "perspective_correction") else None)
        self.rot_fix = {"pcl": 1, "perspective_correction": 2}.get(pos_enc, 0)
        self.use_depth_loss = bool(get("use_depth_loss", False))
        self.img_res_ds = int(get("img_res_ds", img_res) or img_res)
        self.no_crops = bool(get("no_crops", False))
        self.use_grasp_loss = bool(get("use_grasp_loss", False))
        self.use_glb_feat_w_grasp = bool(get("use_glb_feat_w_grasp", False))
        self.separate_hands = bool(get("separate_hands", False))       # model.py:40-50: own trunk weights per side
        self.regress_center_corner = bool(get("regress_center_corner", False))   # model.py:157-172, 426-433
        self.use_glb_feat = bool(get("use_glb_feat", False))
        unsupported = {
            f"pos_enc={pos_enc!r}": self.enc_mode is None,
            "tf_decoder": bool(get("tf_decoder", False)),
            "use_depth_loss with no_crops": self.use_depth_loss and self.no_crops,      # `depth_r` undefined in the reference (:308-310, 422)
            "use_render_seg_loss": bool(get("use_render_seg_loss", False)),
            # the reference itself fails on these combinations (`features` / `feat_vec` undefined, model.py:191-201, 402-404;
            # center_head on a 4-D map, :428)
            "use_glb_feat=False with no_crops": not self.use_glb_feat and self.no_crops,
            "use_glb_feat=False with use_glb_feat_w_grasp": (not self.use_glb_feat and self.use_grasp_loss and self.use_glb_feat_w_grasp),
            "regress_center_corner with no_crops": self.regress_center_corner and self.no_crops,
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"hands_amd.HandsLight: unsupported config switches {bad}")
        self.n_freq = int(get("n_freq_pos_enc", 4))
        feat_dim = 2048
        self.feat_dim = feat_dim
        self.backbone = ResNet50Params()      # (the reference builds it even with use_glb_feat = False: same state_dict keys)
        # model.py:60-77: conv1 of the hand trunk takes the image-level encoding as extra input channels
        self.enc_channels = ({1: 4, 2: 16, 3: 20}[IMAGE_ENC[pos_enc]] * self.n_freq if self.enc_mode == "image" else
                             4 * self.n_freq if self.enc_mode == "dense" else 0)
        if self.separate_hands:
            self.hand_backbone_r = ResNet50Params(3 + self.enc_channels)
            self.hand_backbone_l = ResNet50Params(3 + self.enc_channels)
        else:
            self.hand_backbone = ResNet50Params(3 + self.enc_channels)
        self.head_r = HandHMR(feat_dim, True, 3)
        self.head_l = HandHMR(feat_dim, False, 3)
        self.latent_channels = (5 * 4 * self.n_freq if self.enc_mode == "latent" else 4 * self.n_freq if pos_enc == "dense_latent" else
                                6 if pos_enc == "cam_conv" else 0)
        fc_dim = feat_dim + self.latent_channels                                           # model.py:79-88
        self.feature_conv = nn.Sequential(
            nn.Conv2d(fc_dim, 1024, 1, bias=False), nn.ReLU(inplace=True),
            nn.Conv2d(1024, 512, 3, bias=False), nn.ReLU(inplace=True),
            nn.Conv2d(512, 256, 3, bias=False), nn.ReLU(inplace=True),
            nn.Flatten(), nn.Linear(256 * 3 * 3, feat_dim), nn.ReLU(inplace=True))
        assets = mano_assets or (None, None)
        self.mano_r = MANOHead(True, focal_length, img_res, assets[0])
        self.mano_l = MANOHead(False, focal_length, img_res, assets[1])
        if self.use_grasp_loss:                                                            # model.py:113-125
            gdim = 10 + 144 + (feat_dim if self.use_glb_feat_w_grasp else 0)
            self.grasp_classifier = nn.Sequential(
                nn.Linear(gdim, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 512),
                nn.ReLU(inplace=True), nn.Linear(512, 128), nn.ReLU(inplace=True), nn.Linear(128, 9))
        if self.use_depth_loss:                                                            # model.py:133-155
            cv = lambda i, o: nn.Conv2d(i, o, 3, 1, 1)
            up = lambda k: nn.Upsample(scale_factor=k, mode="bilinear", align_corners=True)
            self.depth_mlp = nn.Sequential(
                cv(fc_dim + 2, 256), nn.ReLU(True), cv(256, 256), nn.ReLU(True), up(4), cv(256, 128), nn.ReLU(True),
                cv(128, 128), nn.ReLU(True), up(4), cv(128, 64), nn.ReLU(True), cv(64, 32), nn.ReLU(True), up(2),
                cv(32, 16), nn.ReLU(True), cv(16, 1))
        if self.regress_center_corner:
            mk = lambda n: nn.Sequential(nn.Linear(feat_dim, 512), nn.ReLU(inplace=True), nn.Linear(512, 128),
                                         nn.ReLU(inplace=True), nn.Linear(128, n))
            self.corner_head = mk(8)
            self.center_head = mk(2)
        self.mode = "train"
        self.img_res = img_res
        self.focal_length = focal_length
        self._holder = _PackedHolder()
        self._ws = {}
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate_packed())

    # ---- packing ------------------------------------------------------------------------------
    @property
    def _packed(self):
        return self._holder.packed

    def invalidate_packed(self):
        """Drop the packed weights (shared with every replica): call after writing parameters IN PLACE
        (``load_state_dict`` and ``.to()`` do it themselves; ``apply_recipe`` calls it)."""
        self._holder.invalidate()
        if any(isinstance(v, torch.Tensor) and v.is_cuda for v in self._ws.values()):
            torch.cuda.synchronize()      # an asynchronous tail of an earlier forward may still use the workspaces
        self._ws = {}

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def _pack_trunk(self, net: ResNet50Params, dev):
        cpu = lambda t: t.detach().cpu()
        bnp = lambda bn: (cpu(bn.weight), cpu(bn.bias), cpu(bn.running_mean), cpu(bn.running_var))
        P = {}
        w, b = fold_bn(cpu(net.conv1.weight), *bnp(net.bn1))
        if w.shape[1] == 3:
            P["stem"] = pack_conv(w, b, 2, 3, dev, cin_pad_to=4)
            # planar form for hands_stem_conv_maxpool_nchw_f32: k = plane * 52 + tap (49 taps + 3 zero columns per plane)
            col = [c * 52 + t for c in range(3) for t in range(49)]
            P["stem_planar"] = pack_linear(w.reshape(64, 147), b, dev, col_index=col, k_total=160)
            P["stem_planar"].macs_per_pixel = 64 * 147
        else:
            # widened conv1 (image-level encodings): the general implicit-GEMM route on an NHWC input whose channel count is
            # padded to a multiple of 16 (hands_image_posenc_nhwc_f32 writes the zeros)
            P["stem_wide"] = pack_conv(w, b, 2, 3, dev, cin_pad_to=(w.shape[1] + 15) // 16 * 16)
        blocks = []
        for li in range(1, 5):
            for blk in getattr(net, f"layer{li}"):
                e = {}
                w, b = fold_bn(cpu(blk.conv1.weight), *bnp(blk.bn1)); e["c1"] = pack_conv(w, b, 1, 0, dev)
                w, b = fold_bn(cpu(blk.conv2.weight), *bnp(blk.bn2))
                e["c2"] = pack_conv(w, b, blk.stride, 1, dev, winograd4=li in self.winograd4_stages)
                w, b = fold_bn(cpu(blk.conv3.weight), *bnp(blk.bn3)); e["c3"] = pack_conv(w, b, 1, 0, dev)
                if blk.downsample is not None:
                    w, b = fold_bn(cpu(blk.downsample[0].weight), *bnp(blk.downsample[1]))
                    e["ds"] = pack_conv(w, b, blk.stride, 0, dev)
                    # conv3 + downsample as ONE GEMM over K = planes + inplanes (hands_conv1x1_dual_nhwc_f32)
                    w3, b3 = fold_bn(cpu(blk.conv3.weight), *bnp(blk.bn3))
                    e["c3ds"] = pack_conv1x1_dual(w3, b3, w, b, dev)
                    e["c3ds_split"] = (w3.shape[1], w.shape[1], blk.stride)
                blocks.append(e)
        P["blocks"] = blocks
        return P

    @torch.no_grad()
    def _pack_head(self, head: HandHMR, dev):
        cpu = lambda t: t.detach().cpu()
        F = self.feat_dim
        P = {}
        ci = head.cam_init
        P["ci0"] = pack_linear(cpu(ci[0].weight), cpu(ci[0].bias), dev)
        P["ci2"] = pack_linear(cpu(ci[2].weight), cpu(ci[2].bias), dev)
        P["ci4"] = pack_linear(cpu(ci[4].weight), cpu(ci[4].bias), dev, n_total=4)
        rf = head.hmr_layer.refine
        P["r0"] = pack_linear(cpu(rf[0].weight), cpu(rf[0].bias), dev, col_index=hmr_state_columns(F),
                              k_total=F + HMR_VEC)
        P["r3"] = pack_linear(cpu(rf[3].weight), cpu(rf[3].bias), dev)
        d = head.hmr_layer.decoders
        wd = torch.cat([cpu(d["pose_6d"].weight), cpu(d["shape"].weight), cpu(d["cam_t/wp"].weight)], 0)
        bd = torch.cat([cpu(d["pose_6d"].bias), cpu(d["shape"].bias), cpu(d["cam_t/wp"].bias)], 0)
        rows = list(range(96)) + [96 + i for i in range(10)] + [108 + i for i in range(3)]
        P["dec"] = pack_linear(wd, bd, dev, row_index=rows, n_total=HMR_VEC)
        return P

    @torch.no_grad()
    def _pack(self, dev):
        cpu = lambda t: t.detach().cpu()
        F = self.feat_dim
        P = {"head_r": self._pack_head(self.head_r, dev), "head_l": self._pack_head(self.head_l, dev)}
        if self.use_glb_feat:
            P["backbone"] = self._pack_trunk(self.backbone, dev)
        if not self.no_crops:
            if self.separate_hands:
                P["hand_backbone_r"] = self._pack_trunk(self.hand_backbone_r, dev)
                P["hand_backbone_l"] = self._pack_trunk(self.hand_backbone_l, dev)
            else:
                P["hand_backbone"] = self._pack_trunk(self.hand_backbone, dev)
        if self.regress_center_corner:
            for nm, head in (("cc_center", self.center_head), ("cc_corner", self.corner_head)):
                P[nm] = [pack_linear(cpu(head[0].weight), cpu(head[0].bias), dev), pack_linear(cpu(head[2].weight), cpu(head[2].bias), dev),
                         pack_linear(cpu(head[4].weight), cpu(head[4].bias), dev, n_total=8)]
        fc = self.feature_conv
        # 'dense_latent' / 'cam_conv': the concatenated map is stored with its channel count padded to a multiple of 16 (zeros)
        P["fc0"] = pack_conv(cpu(fc[0].weight), None, 1, 0, dev,
                             cin_pad_to=self._cat_ld() if self.enc_mode == "dense_latent" else None)
        if self.use_depth_loss:
            dm = self.depth_mlp
            P["depth"] = [pack_conv(cpu(dm[i].weight), cpu(dm[i].bias), 1, 1, dev,
                                    cin_pad_to=self._depth_ld() if i == 0 else None) for i in (0, 2, 5, 7, 10, 12, 15, 17)]
            lin = torch.linspace(-1, 1, 7)                                # model.py:172-175 (`init_grid`, 'ij' meshgrid)
            xg, yg = torch.meshgrid(lin, lin, indexing="ij")
            P["depth_grid"] = torch.stack([xg, yg], dim=-1).reshape(49, 2).contiguous().to(dev)
        P["fc2"] = pack_conv(cpu(fc[2].weight), None, 1, 0, dev)
        P["fc4"] = pack_conv(cpu(fc[4].weight), None, 1, 0, dev)
        # nn.Flatten on NCHW (B,256,3,3): reference column c*9 + hw; NHWC buffer column hw*256 + c
        col = [(k % 9) * 256 + (k // 9) for k in range(256 * 9)]
        P["fc7"] = pack_linear(cpu(fc[7].weight), cpu(fc[7].bias), dev, col_index=col)
        if self.use_grasp_loss:
            g = self.grasp_classifier
            Fg = F if self.use_glb_feat_w_grasp else 0
            # reference cat([shape 10, rot 144(, feat_vec F)]) -> packed row [feat_vec F | rot 144 | shape 10]
            gcol = [Fg + 144 + i for i in range(10)] + [Fg + i for i in range(144)] + list(range(Fg))
            P["g0"] = pack_linear(cpu(g[0].weight), cpu(g[0].bias), dev, col_index=gcol, k_total=Fg + 154)
            P["g2"] = pack_linear(cpu(g[2].weight), cpu(g[2].bias), dev)
            P["g4"] = pack_linear(cpu(g[4].weight), cpu(g[4].bias), dev)
            P["g6"] = pack_linear(cpu(g[6].weight), cpu(g[6].bias), dev, n_total=12)
        P["mano_r"] = pack_mano(self.mano_r.mano.asset(), dev)
        P["mano_l"] = pack_mano(self.mano_l.mano.asset(), dev)
        for side in ("mano_r", "mano_l"):
            m = P[side]
            m["consts"] = mano_consts(m)
        return P

    def _cat_ld(self):
        """Row length of the map feature_conv reads (model.py:79-88): F + encoding channels, padded to 16 for the per-pixel modes."""
        n = self.feat_dim + self.latent_channels
        return (n + 15) // 16 * 16 if self.enc_mode == "dense_latent" else n

    def _depth_ld(self):
        return (self.feat_dim + self.latent_channels + 2 + 15) // 16 * 16

    def replica(self):
        """A second handle on the SAME parameters and packed weights with its own workspaces, side
        streams and engine switches, for a second request stream: two forwards in flight (one per torch
        stream, one replica each) overlap the tail of one with the trunks of the other.  Plain PyTorch
        stream semantics: each forward's outputs are valid on the stream it was called on.  The packed
        weights live in a holder both share, so ``load_state_dict`` / ``.to()`` / ``invalidate_packed`` on
        either one is seen by both."""
        self.packed(next(self.parameters()).device)
        import copy
        r = copy.copy(self)
        r._ws = {}
        r.engine = self.engine.clone_settings()
        return r

    def packed(self, dev):
        h = self._holder
        if h.packed is None or h.dev != dev:
            h.packed, h.dev = self._pack(dev), dev
        return h.packed

    # ---- buffers ------------------------------------------------------------------------------
    def _side_stream(self, dev, name="side_stream"):
        st = self._ws.get(name)
        if st is None or st.device != dev:
            st = torch.cuda.Stream(device=dev)
            self._ws[name] = st
        return st

    def _buf(self, name, numel, dev):
        t = self._ws.get(name)
        if t is None or t.numel() < numel or t.device != dev:
            if t is not None and t.is_cuda:
                # a workspace is being replaced (larger batch): an asynchronous tail or a side stream of an earlier
                # forward may still be using the old block, which the allocator would hand out again at once
                torch.cuda.synchronize(t.device)
            t = torch.empty(numel, dtype=torch.float32, device=dev)
            self._ws[name] = t
        return t

    # ---- kernel launch helpers ----------------------------------------------------------------
    # model-less launch helpers on the default engine (tests / tools driving a single layer)
    _conv = staticmethod(lambda *a, **kw: DEFAULT_ENGINE.conv(*a, **kw))
    _conv_dual = staticmethod(lambda *a, **kw: DEFAULT_ENGINE.conv_dual(*a, **kw))

    def _blocks(self, L, blocks, cur, nxt, t1, t2, ds, B, H, W, stream, final_dst, final_off):
        """A run of bottlenecks (resnet.py:134-154) on ping-pong buffers; the last one writes ``final_dst``
        at float offset ``final_off``.  Returns (H, W) of the output map."""
        n = len(blocks)
        for i, e in enumerate(blocks):
            self.engine.conv(L, e["c1"], cur, B, H, W, t1, True, stream)
            last = i + 1 == n
            dst, off = (final_dst, final_off) if last else (nxt, 0)
            H2, W2 = self.engine.conv(L, e["c2"], t1, B, H, W, t2, True, stream)
            if "ds" in e and self.engine.fuse_downsample:
                self.engine.conv_dual(L, e["c3ds"], e["c3ds_split"], t2, cur, B, H2, W2, H, W, dst, stream, out_off=off)
            else:
                if "ds" in e:
                    self.engine.conv(L, e["ds"], cur, B, H, W, ds, False, stream)
                    ident = ds
                else:
                    ident = cur
                self.engine.conv(L, e["c3"], t2, B, H2, W2, dst, True, stream, res=ident, out_off=off)
            H, W = H2, W2
            cur, nxt = dst, cur
        return H, W

    def _trunk(self, L, P, segs, B, res_in, stream, tag, cap_B, out=None, out_off=0):
        """ResNet-50 trunk on B images given as NCHW segments ``[(tensor, first image, n images), ...]`` (the
        reference's input layout, read in place); returns (B,7,7,2048) features (flat tensor).
        (Running stem + layer1 + layer2 per sub-batch of 32-128 images, to keep their HBM-bound 1x1 layers'
        tensors inside the 256 MB Infinity Cache, was measured 1-18 % SLOWER than whole-job launches.)"""
        dev = segs[0][0].device
        per = 112 * 112 * 64 * (res_in * res_in) // (224 * 224) + 64      # floats per image of the largest map
        cap = cap_B * per
        a = self._buf("trunk_a_" + tag, cap, dev); b = self._buf("trunk_b_" + tag, cap, dev)
        t1 = self._buf("trunk_t1_" + tag, cap, dev); t2 = self._buf("trunk_t2_" + tag, cap, dev)
        ds = self._buf("trunk_ds_" + tag, cap, dev)
        Hs, Ws = (res_in - 1) // 2 + 1, (res_in - 1) // 2 + 1                # stem conv map
        Hp, Wp = (Hs + 2 - 3) // 2 + 1, (Ws + 2 - 3) // 2 + 1                # after the max-pool
        img_floats = 3 * res_in * res_in
        done = 0
        for src, first, n in segs:
            if "stem_wide" in P:
                # widened conv1 (image-level encodings): `src` is the NHWC (.., res, res, Cpad) tensor hands_image_posenc_nhwc_f32
                # wrote; general implicit GEMM + the max-pool kernel
                Cp = P["stem_wide"].Cin
                self.engine.conv(L, P["stem_wide"], src, n, res_in, res_in, a, True, stream, x_off=first * res_in * res_in * Cp)
                check(L.hands_maxpool3x3s2_nhwc_f32(ptr(a), ptr(b, done * Hp * Wp * 64), n, Hs, Ws, 64, stream), "maxpool")
            elif self.engine.fuse_stem_pool:
                # conv1 + bn1 + relu + maxpool in one kernel straight from the NCHW image: neither the NHWC copy
                # of the input nor the 112x112x64 map ever reaches HBM
                self.engine.stem_pool_nchw(L, P["stem_planar"], src, first * img_floats, b, done * Hp * Wp * 64, n,
                                           res_in, res_in, 1, stream)
            else:
                x4 = self._buf("trunk_x4_" + tag, cap_B * res_in * res_in * 4, dev)
                check(L.hands_nchw3_to_nhwc4_f32(ptr(src, first * img_floats), ptr(x4), n, res_in, res_in, stream), "nchw->nhwc4")
                self.engine.conv(L, P["stem"], x4, n, res_in, res_in, a, True, stream)
                check(L.hands_maxpool3x3s2_nhwc_f32(ptr(a), ptr(b, done * Hp * Wp * 64), n, Hs, Ws, 64, stream), "maxpool")
            done += n
        assert done == B
        feat = out if out is not None else self._buf("feat_" + tag, B * 49 * P["blocks"][-1]["c3"].Cout, dev)
        H, W = self._blocks(L, P["blocks"], b, a, t1, t2, ds, B, Hp, Wp, stream, feat, out_off)
        return feat, H, W

    # ---- forward ------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, inputs, meta_info):
        L = _lib.lib()
        img = inputs["img"]
        dev = img.device
        if dev.type != "cuda":
            raise RuntimeError("hands_amd.HandsLight runs on a HIP device only (no CPU fallback); "
                               "move the module and its inputs to 'cuda'")
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        img, r_img, l_img = f32(img), f32(inputs["r_img"]), f32(inputs["l_img"])
        K = f32(meta_info["intrinsics"])
        assert K.shape[1:] == (3, 3)                               # transforms.py:322-326
        bz, c, res, res_w = img.shape
        assert c == 3 and res == res_w and r_img.shape == img.shape and l_img.shape == img.shape
        B2 = 2 * bz
        F = self.feat_dim
        P = self.packed(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        buf = lambda n, numel: self._buf(n, numel, dev)

        # -- trunks (model.py:193, 238-239).  r and l crops share `hand_backbone`, so the 2*bz crops
        #    are one batch; the 3*bz trunk passes are cut into `trunk_chunks` jobs that run on separate
        #    HIP streams: each conv launch is only 400-3000 workgroups on 256 CUs, and workgroups of
        #    the other streams fill its tail (and HBM-bound layers overlap with MFMA-bound ones).
        main = torch.cuda.current_stream(dev)
        # The tail of the forward (feature_conv, HMR heads, MANO, grasp: ~3 ms of launch-bound work) runs on its
        # own stream and is NOT joined back: the result is a stream_xdict that makes the consumer's stream
        # wait at first access, so the next forward's trunks start under this forward's tail.  The trunk
        # outputs the tail reads are double-buffered by call parity; everything the tail reads from the
        # caller (angles, intrinsics, flags) is copied on the caller's stream first.
        async_tail = bool(self.async_tail and self.engine.overlap and not torch.cuda.is_current_stream_capturing())
        par = self._calls & 1 if async_tail else 0
        self._calls += 1
        if not torch.cuda.is_current_stream_capturing():
            # the tail two calls ago read these feature buffers (long finished); a synchronous call waits for
            # every earlier asynchronous tail
            for q in ((par,) if async_tail else (0, 1)):
                prev_tail = self._ws.get(f"tail_done{q}")
                if prev_tail is not None:
                    main.wait_event(prev_tail)
        featg = buf(f"feat_g{par}", bz * 49 * F) if self.use_glb_feat else None
        feath = buf(f"feat_h{par}", B2 * 49 * F) if not self.no_crops else None
        gch, hch = self.trunk_chunks if self.engine.overlap else (1, 1)
        if self.no_crops:          # model.py:199-201: no hand trunks; the global job may as well be cut in two
            gch, hch = (2 if self.engine.overlap else 1), 0
        if self.separate_hands and hch:
            hch = 2                # one job per side: the sides have their own weights (model.py:226-228)
        gch, hch = (max(1, min(gch, bz)) if self.use_glb_feat else 0), min(hch, B2)
        need_cc = (not self.no_crops) and self.enc_mode in ("latent", "image")    # model.py:199-201: no_crops skips every encoding
        center = (torch.cat([f32(inputs["r_center_angle"]), f32(inputs["l_center_angle"])], 0)
                  if (need_cc or self.rot_fix == 2) else None)
        corner = torch.cat([f32(inputs["r_corner_angle"]), f32(inputs["l_corner_angle"])], 0) if need_cc else None
        rot_in = torch.cat([f32(inputs["r_rot"]), f32(inputs["l_rot"])], 0) if self.rot_fix == 1 else None    # 'pcl' (B2, 3, 3)
        wide = None
        dense_enc = None
        if self.enc_mode in ("dense", "dense_latent") and not self.no_crops:
            # model.py:462-481: per-pixel angle maps (bz, 2 | 6, w, h) + crop masks, encoded / masked / resized on the device
            ang = [f32(inputs["r_dense_angle"]), f32(inputs["l_dense_angle"])]
            msk = [f32(inputs["r_dense_mask"]), f32(inputs["l_dense_mask"])]
            Ca, Hs, Ws = ang[0].shape[1:]
            L_enc = 0 if self.pos_enc == "cam_conv" else self.n_freq
            assert (2 * L_enc * Ca if L_enc else Ca) == (self.enc_channels if self.enc_mode == "dense" else self.latent_channels)
            if self.enc_mode == "dense":
                # model.py:220-224: cat([crop, encoding]) as the NHWC input of the widened conv1
                assert res == self.img_res_ds, "pos_enc='dense': the crops must have the size args.img_res_ds"
                Cp = P["hand_backbone_r" if self.separate_hands else "hand_backbone"]["stem_wide"].Cin
                wide = buf("wide_in", B2 * res * res * Cp)
                for side, im in enumerate((r_img, l_img)):
                    check(L.hands_dense_posenc_f32(ptr(ang[side]), ptr(msk[side]), ptr(im), ptr(wide, side * bz * res * res * Cp),
                                                   bz, Ca, Hs, Ws, L_enc, self.img_res_ds, res, res, Cp, 3, stream), "dense_posenc")
            else:
                # model.py:244-249 / 276-281: resized to img_res_ds and then to the 7x7 feature map -- an input-only computation,
                # done here on the caller's stream so that the tail never reads the caller's maps
                Ce = self.latent_channels
                dense_enc = buf(f"dense_enc{par}", B2 * 49 * Ce)
                for side in (0, 1):
                    check(L.hands_dense_posenc_f32(ptr(ang[side]), ptr(msk[side]), None, ptr(dense_enc, side * bz * 49 * Ce),
                                                   bz, Ca, Hs, Ws, L_enc, self.img_res_ds, 7, 7, Ce, 0, stream), "dense_posenc")
        if self.enc_mode == "image" and not self.no_crops:
            # model.py:203-218: cat([crop, enc repeated over the pixels]) as the NHWC input of the widened conv1
            Cp = P["hand_backbone_r" if self.separate_hands else "hand_backbone"]["stem_wide"].Cin
            wide = buf("wide_in", B2 * res * res * Cp)
            mode = IMAGE_ENC[self.pos_enc]
            for side, im in enumerate((r_img, l_img)):
                check(L.hands_image_posenc_nhwc_f32(ptr(im), ptr(center, side * bz * 2), ptr(corner, side * bz * 8),
                                                    ptr(wide, side * bz * res * res * Cp), bz, res, res, self.n_freq, mode, Cp,
                                                    stream), "image_posenc")
        jobs = []   # (weights, NCHW segments [(tensor, first image, n)], first row of the job, n images, out buffer)
        for c in range(gch):
            lo, hi = c * bz // gch, (c + 1) * bz // gch
            jobs.append((P["backbone"], [(img, lo, hi - lo)], lo, hi - lo, featg))
        for c in range(hch):       # rows [0, bz) of the hand batch are the right crops, [bz, 2 bz) the left crops
            lo, hi = c * B2 // hch, (c + 1) * B2 // hch
            segs = []
            if wide is not None:
                segs.append((wide, lo, hi - lo))
            else:
                if lo < bz:
                    segs.append((r_img, lo, min(hi, bz) - lo))
                if hi > bz:
                    segs.append((l_img, max(lo, bz) - bz, hi - max(lo, bz)))
            jobs.append((P[("hand_backbone_r", "hand_backbone_l")[c]] if self.separate_hands else P["hand_backbone"], segs, lo, hi - lo, feath))
        ev0 = torch.cuda.Event()
        ev0.record(main)
        fh = fw = 7
        done = []
        for ji, (Pt, segs, lo, n, ob) in enumerate(jobs):
            # the largest job (last) stays on the caller's stream
            st = main if (ji == len(jobs) - 1 or not self.engine.overlap) else self._side_stream(dev, f"side{ji}")
            if st is not main:
                st.wait_event(ev0)
            _, fh, fw = self._trunk(L, Pt, segs, n, res, st.cuda_stream, f"j{ji}", n, out=ob, out_off=lo * 49 * F)
            if st is not main:
                ev = torch.cuda.Event()
                ev.record(st)
                done.append(ev)
        for ev in done:          # join only after every job has been enqueued
            main.wait_event(ev)
        assert fh * fw == 49
        HW = fh * fw
        # small per-sample inputs the tail reads: private copies made on the caller's stream (center / corner above)
        flipped = meta_info["is_flipped"].to(device=dev, dtype=torch.int64).contiguous()
        if async_tail:
            K, flipped = K.clone(), flipped.clone()
            tail = self._side_stream(dev, "tail")
            evt = torch.cuda.Event()
            evt.record(main)
            tail.wait_event(evt)
            for t in (center, corner, K, flipped, rot_in):
                if t is not None:
                    t.record_stream(tail)
        else:
            tail = main
        with torch.cuda.stream(tail):
            output = self._forward_tail(L, P, dev, tail, bz, fh, fw, featg, feath, center, corner, K, flipped, dense_enc, rot_in)
        if not async_tail:
            return output
        ready = torch.cuda.Event()
        ready.record(tail)
        self._ws[f"tail_done{par}"] = ready
        return stream_xdict(output, ready, dev)

    def _depth_head(self, L, P, dev, stream, cat, B2, bz, fh, fw):
        """`predict_depth` (model.py:177-185, 438-441) on the map feature_conv reads: (x, y) grid channels appended, eight 3x3 / pad 1
        convolutions, three align_corners upsamplings (7 -> 28 -> 112 -> 224).  Returns (depth.r, depth.l), each (bz, 224, 224)."""
        F = self.feat_dim
        Cr, lda, ldd = F + self.latent_channels, (self._cat_ld() if self.enc_mode in ("latent", "dense_latent") else F), self._depth_ld()
        d = P["depth"]
        out = torch.empty(B2, 16 * fh * 2, 16 * fw * 2, device=dev)
        step = 256                       # images per pass: the 224 x 224 x 32 map of 256 crops is 1.6 GB (and < 2^31 floats)
        for lo in range(0, B2, step):
            n = min(step, B2 - lo)
            buf = lambda nm, numel: self._buf(nm, numel, dev)
            din = buf("depth_in", n * fh * fw * ldd)
            check(L.hands_concat_nhwc_f32(ptr(cat, lo * fh * fw * lda), lda, Cr, None, 0, ptr(P["depth_grid"]), 0, 2, ptr(din), ldd,
                                          n, n, fh * fw, stream), "depth_in")
            h, w = fh, fw
            x = din
            for i, pc in enumerate(d):
                y = buf(f"depth_c{i}", n * h * w * pc.Cout)
                self.engine.conv(L, pc, x, n, h, w, y, i < 7, stream)
                x = y
                if i in (1, 3, 5):                            # nn.Upsample(x4, x4, x2; bilinear, align_corners=True)
                    k = 2 if i == 5 else 4
                    u = buf(f"depth_u{i}", n * h * k * w * k * pc.Cout)
                    check(L.hands_upsample_bilinear_ac_f32(ptr(x), ptr(u), n, h, w, h * k, w * k, pc.Cout, stream), "upsample_ac")
                    x, h, w = u, h * k, w * k
            out[lo:lo + n] = x[: n * h * w * 4].view(n, h, w, 4)[..., 0]      # Cout = 1 stored with a pixel stride of 4 floats
        return out[:bz], out[bz:]

    def _forward_tail(self, L, P, dev, main, bz, fh, fw, featg, feath, center, corner, K, flipped, dense_enc=None, rot_in=None):
        """Everything after the trunks (model.py:196-411), enqueued on ``main`` (the tail stream)."""
        B2, F, HW = 2 * bz, self.feat_dim, fh * fw
        stream = main.cuda_stream
        buf = lambda n, numel: self._buf(n, numel, dev)
        feat_vec = buf("feat_vec", bz * F)
        if self.use_grasp_loss and self.use_glb_feat_w_grasp:
            # sum-pool (model.py:196); only the grasp head reads it
            check(L.hands_sumpool_nhwc_f32(ptr(featg), ptr(feat_vec), bz, HW, F, F, stream), "sumpool")
        ld = F + HMR_VEC
        state = buf("state", B2 * ld)
        if self.no_crops:
            # model.py:316-318 -> HandHMR.forward(features, use_pool=True) (hand_hmr.py:73-78): both heads read the average-pooled
            # GLOBAL feature map -- written straight into the feat segment of the right and the left state rows
            for side in (0, 1):
                check(L.hands_avgpool_nhwc_f32(ptr(featg), ptr(state, side * bz * ld), bz, HW, F, ld, stream), "avgpool")
        else:
            if self.enc_mode == "latent":
                # -- KPE concat (model.py:258-271, 288-304) ------------------------------------------------
                Cc = F + 20 * self.n_freq
                cat = buf("cat", B2 * HW * Cc)
                check(L.hands_kpe_concat_f32(ptr(feath), ptr(featg) if self.use_glb_feat else None, ptr(center), ptr(corner),
                                             ptr(cat), B2, bz, HW, F, self.n_freq, stream), "kpe_concat")
            elif self.enc_mode == "dense_latent":
                # -- cat([features (+ global features), resized per-pixel maps]) (model.py:250-256, 282-288) -------------------
                Cc, Ce = self._cat_ld(), self.latent_channels
                cat = buf("cat", B2 * HW * Cc)
                check(L.hands_concat_nhwc_f32(ptr(feath), F, F, ptr(featg) if self.use_glb_feat else None, F, ptr(dense_enc),
                                              HW * Ce, Ce, ptr(cat), Cc, B2, bz, HW, stream), "concat")
            else:
                cat = feath            # pos_enc None / image-level: feature_conv reads the crop features as they are
            depth = self._depth_head(L, P, dev, stream, cat, B2, bz, fh, fw) if self.use_depth_loss else None
            # -- feature_conv (model.py:91-101, 313-314) -> HMR state rows ---------------------------
            f1 = buf("fc1", B2 * HW * 1024)
            self.engine.conv(L, P["fc0"], cat, B2, fh, fw, f1, True, stream)
            f2 = buf("fc2", B2 * (fh - 2) * (fw - 2) * 512)
            h2, w2 = self.engine.conv(L, P["fc2"], f1, B2, fh, fw, f2, True, stream)
            f3 = buf("fc3", B2 * (h2 - 2) * (w2 - 2) * 256)
            h3, w3 = self.engine.conv(L, P["fc4"], f2, B2, h2, w2, f3, True, stream, splitk_n=8)   # 3x3 output map: 72 tiles at bz=256, K=4608
            assert h3 * w3 * 256 == P["fc7"].Cin
            self.engine.conv(L, P["fc7"], f3, B2, 1, 1, state, True, stream, out_ps=ld, splitk=True)

        # -- HandHMR x2 (hand_hmr.py:73-92, hmr_layer.py:67-86) ----------------------------------
        caminit4 = buf("caminit4", B2 * 4)
        # the two heads are independent chains of latency-bound M = bz GEMMs: run them side by side
        evh = torch.cuda.Event()
        evh.record(main)
        joins = []
        for side, hp in ((0, P["head_r"]), (1, P["head_l"])):
            hs = main if (side == 1 or not self.engine.overlap) else self._side_stream(dev, "side_head")
            if hs is not main:
                hs.wait_event(evh)
            sh = hs.cuda_stream
            h512a, h512b = buf(f"h512a{side}", bz * 512), buf(f"h512b{side}", bz * 512)
            x1, x2 = buf(f"x1024a{side}", bz * 1024), buf(f"x1024b{side}", bz * 1024)
            so = side * bz * ld
            self.engine.conv(L, hp["ci0"], state, bz, 1, 1, h512a, True, sh, in_ps=ld, x_off=so, splitk=True)
            self.engine.conv(L, hp["ci2"], h512a, bz, 1, 1, h512b, True, sh, splitk=True)
            self.engine.conv(L, hp["ci4"], h512b, bz, 1, 1, caminit4, False, sh, out_off=side * bz * 4)
            check(L.hands_hmr_init_f32(ptr(state, so), ptr(caminit4, side * bz * 4), bz, ld, F, sh),
                  "hmr_init")
            for _ in range(3):
                self.engine.conv(L, hp["r0"], state, bz, 1, 1, x1, True, sh, in_ps=ld, x_off=so, splitk=True)
                self.engine.conv(L, hp["r3"], x1, bz, 1, 1, x2, True, sh, splitk=True)
                self.engine.conv(L, hp["dec"], x2, bz, 1, 1, state, False, sh, res=state, out_ps=ld,
                           res_ps=ld, out_off=so + F, res_off=so + F, splitk=True)
            if hs is not main:
                ev = torch.cuda.Event()
                ev.record(hs)
                joins.append(ev)
        for ev in joins:
            main.wait_event(ev)
        rotmat = buf("rotmat", B2 * 144)
        check(L.hands_rot6d_to_matrix_f32(ptr(state, F), ld, ptr(rotmat), B2, stream), "rot6d")
        if self.rot_fix == 1:      # 'pcl' (model.py:330-334): in place on the heads' output -- the flip swap and the grasp head see it
            check(L.hands_rot_leftmul_f32(ptr(rotmat), ptr(rot_in), B2, stream), "rot_leftmul")
        st = state[: B2 * ld].view(B2, ld)
        shape = st[:, F + 96:F + 106].contiguous()
        cam = st[:, F + 108:F + 111].contiguous()
        caminit = caminit4[: B2 * 4].view(B2, 4)[:, :3].contiguous()

        # -- is_flipped swap (model.py:341-368), per sample on device ----------------------------
        rot_m = torch.empty(B2, 16, 3, 3, device=dev)
        shape_m = torch.empty(B2, 10, device=dev)
        cam_m = torch.empty(B2, 3, device=dev)
        caminit_m = torch.empty(B2, 3, device=dev)
        check(L.hands_flip_swap_f32(ptr(flipped), ptr(rotmat), ptr(shape), ptr(cam), ptr(caminit),
                                    ptr(rot_m), ptr(shape_m), ptr(cam_m), ptr(caminit_m), bz, stream),
              "flip_swap")

        if self.rot_fix == 2:      # 'perspective_correction' (model.py:370-376): after the swap, see hands_hip.h for the grasp quirk
            check(L.hands_perspective_correction_f32(ptr(rot_m), ptr(rotmat), ptr(center), ptr(flipped), bz, stream), "persp")

        # -- MANOHead x2 (mano_head.py:21-65) -----------------------------------------------------
        output = run_mano_heads(L, P["mano_r"], P["mano_l"], rot_m, shape_m, cam_m, caminit_m, K,
                                float(self.img_res), bz, stream, buf, self.engine)

        # -- center / corner regression from the feature_conv vectors (model.py:426-433) -------------------
        extra = None
        if self.regress_center_corner:
            extra = xdict()
            for nm, key in (("cc_center", "center"), ("cc_corner", "corner")):
                l0, l1, l2 = P[nm]
                h1, h2 = buf("cc_h1", B2 * 512), buf("cc_h2", B2 * 128)
                o = torch.empty(B2, 8, device=dev)
                self.engine.conv(L, l0, state, B2, 1, 1, h1, True, stream, in_ps=ld, splitk=True)
                self.engine.conv(L, l1, h1, B2, 1, 1, h2, True, stream, splitk=True)
                self.engine.conv(L, l2, h2, B2, 1, 1, o, False, stream)
                n = 2 if key == "center" else 8
                extra[key + ".r"] = o[:bz, :n].contiguous()
                extra[key + ".l"] = o[bz:, :n].contiguous()
        # -- grasp classifier on the UN-flipped HMR outputs (model.py:401-411) -------------------
        if self.use_depth_loss and not self.no_crops:          # model.py:420-424 (merged after the grasp outputs, before center / corner)
            dx = xdict()
            dx["depth.r"], dx["depth.l"] = depth
            if extra is None:
                extra = dx
            else:
                dx.merge(extra)
                extra = dx
        if not self.use_grasp_loss:
            if extra is not None:
                output.merge(extra)
            return output
        Fg = F if self.use_glb_feat_w_grasp else 0
        gld = P["g0"].Cin
        gin = buf("grasp_in", B2 * gld)
        check(L.hands_grasp_input_f32(ptr(state, F + 96), ld, ptr(rotmat), ptr(feat_vec), ptr(gin), B2, bz,
                                      Fg, gld, stream), "grasp_input")
        g1, g2, g3 = buf("g1", B2 * 1024), buf("g2", B2 * 512), buf("g3", B2 * 128)
        g4 = torch.empty(B2, 12, device=dev)
        self.engine.conv(L, P["g0"], gin, B2, 1, 1, g1, True, stream, splitk=True)
        self.engine.conv(L, P["g2"], g1, B2, 1, 1, g2, True, stream, splitk=True)
        self.engine.conv(L, P["g4"], g2, B2, 1, 1, g3, True, stream, splitk=True)
        self.engine.conv(L, P["g6"], g3, B2, 1, 1, g4, False, stream)
        grasp = xdict()
        grasp["grasp.r"] = g4[:bz, :9].contiguous()
        grasp["grasp.l"] = g4[bz:, :9].contiguous()
        output.merge(grasp)
        if extra is not None:          # model.py:426-433: merged after the grasp outputs
            output.merge(extra)
        return output
