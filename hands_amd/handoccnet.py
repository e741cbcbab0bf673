"""``HandOccNet`` -- the handoccnet_light forward path on MI355X.

Mirror of the reference ``HandOccNet(focal_length, img_res, args)``
(src/models/handoccnet_light/model.py:17-129): same forward contract, 22 output keys and the same 906
``state_dict`` tensors (the parameter tree is materialised from ``manifests/handoccnet_light.json``).

    statement                                              kernel
    model.py:66-70    resize 224->256, cat                 hands_resize_crop_nchw3_to_nhwc4_f32
    backbone.py:44-53 LeakyReLU ResNet-50                  hands_conv2d_nhwc_f32 (BN folded, LEAKY epilogue)
    backbone.py:54-62 FPN top-down, smooth, avg-pool       conv + hands_upsample_bilinear_add_f32 + hands_pool2x2
    cbam.py:72-82     SpatialGate                          hands_channel_pool_f32, 7x7 conv, hands_gate_apply_f32
    transformer.py    FIT / SET (2 blocks each)            hands_add_embed2_f32, 1x1 convs, hands_flash_attention_f32
                                                           (fp32 MFMA, FIT gate fused), LayerNorm, GELU MLP, 3x3 fusion
    hand_head.py      hourglass, heat-maps, encoder        hands_bn_leaky_f32 + conv (pre-activation units),
                                                           hands_pool2x2, hands_upsample_nearest2x_add_f32,
                                                           hands_spatial_softmax_f32
    mano_head.py:190-207 MLP regressor, rot6d (columns)    GEMMs, hands_rot6d_to_matrix_cols_f32
    model.py:103-120  MANOHead x2, grasp MLP               shared with hands_light
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT_GELU, ACT_LEAKY_RELU, ACT_NONE, ACT_RELU, check, ptr
from .engine import ConvEngine, EngineSwitches
from .hands_light import MANOHead, _Args, mano_consts, run_mano_heads
from .packing import BN_EPS, PackedConv, fold_bn, pack_conv, pack_linear, pack_mano
from .param_tree import build_tree, load_manifest
from .xdict import stream_xdict, xdict

HANDOCC_DEFAULT_ARGS = _Args(pos_enc="center+corner_latent", n_freq_pos_enc=4, use_grasp_loss=True,
                             use_render_seg_loss=False, img_res=224, focal_length=1000.0,
                             method="handoccnet_light")
NTOK, CF, HEADS = 1024, 256, 4
STAGES = ("resnet", "fpn", "fit", "set", "hourglass", "reghead", "encoder", "mlp")


def stage_of(p: str) -> str:
    """Stage of a reference parameter prefix: the units ``HandOccNet.acc64_stages`` is expressed in."""
    if p.startswith("backbone.layer"):
        return "resnet"                    # backbone.py:68-119 (stem + the four stages)
    if p.startswith("backbone."):
        return "fpn"                       # backbone.py:54-62 laterals / smoothing, cbam.py:72-82 gate
    if p.startswith("FIT"):
        return "fit"
    if p.startswith("SET"):
        return "set"
    if ".hand_regHead.hg" in p:
        return "hourglass"                 # hand_head.py:217-235
    if ".hand_regHead" in p:
        return "reghead"                   # hand_head.py:75-94: res unit, fc, score -> the heat-map logits
    if ".hand_Encoder" in p:
        return "encoder"                   # hand_head.py:266-280
    return "mlp"                           # mano_head.py:190-207, the KPE MLP, the grasp classifier


def _conv_fns(L, stream, new, engine, small_map_splitk=True):
    """(conv, hconv) launchers bound to one HIP stream; hconv = per-sample rows (split-K head GEMMs)."""
    def conv(pc: PackedConv, x, B, H, W, act=ACT_NONE, res=None, out=None, **kw):
        Ho = (H + 2 * pc.pad - pc.KH) // pc.stride + 1
        Wo = (W + 2 * pc.pad - pc.KW) // pc.stride + 1
        out = out if out is not None else new(B, Ho, Wo, pc.Cout)
        if small_map_splitk in (True, "deep") and H * W > 1 and Ho * Wo <= 64 and "splitk" not in kw:
            # 8x8 ... 2x2 maps (layer4, the deep hourglass / encoder levels): a handful of output tiles
            # walking K = 1152-4608 serially.  The slice count depends on the map size and K only, never
            # on the batch, so every output bit stays independent of the batch size.
            S = 4 if Ho * Wo > 16 else 8
            kw["splitk_n"] = max(1, min(S, pc.Kpad // 128))
        elif (small_map_splitk in (True, "16x16") and H * W > 1 and Ho * Wo <= 256 and pc.Cout <= 128 and pc.Kpad >= 1024
              and "splitk" not in kw):
            kw["splitk_n"] = 2          # 16x16 maps, one n-tile: 2 output tiles per crop
        engine.conv(L, pc, x, B, H, W, out, act, stream, res=res, **kw)
        return out, Ho, Wo

    return conv, (lambda *a, **kw: conv(*a, splitk=True, **kw))


class HandOccNet(EngineSwitches, nn.Module):
    graph_private_buffers = True    # forward allocates every activation per call: GraphedForward(depth=2) is safe

    def __init__(self, focal_length=1000.0, img_res=224, args=None, mano_assets=None):
        super().__init__()
        args = args if args is not None else HANDOCC_DEFAULT_ARGS
        get = args.get if hasattr(args, "get") else (lambda k, d=None: getattr(args, k, d))
        self.args = args
        # built: pos_enc 'center+corner_latent' (shipped) or None (model.py:74-89: no KPE anywhere), grasp head on or off;
        # 'dense_latent' fails in the reference's own PositionalEncoding (hamer_light/pos_emb.py:41 calls compute_dense_pos_enc
        # without its `size` argument)
        if get("pos_enc") not in ("center+corner_latent", None) or get("use_render_seg_loss", False):
            raise NotImplementedError("hands_amd.HandOccNet: pos_enc must be 'center+corner_latent' or None, renderer off")
        self.n_freq = int(get("n_freq_pos_enc", 4))
        self.input_size = (256, 256)
        self.pos_enc = get("pos_enc")
        self.use_grasp_loss = bool(get("use_grasp_loss", False))
        skip = (("kpe.",) if self.pos_enc is None else ()) + (("grasp_classifier.",) if not self.use_grasp_loss else ())
        build_tree(self, load_manifest("handoccnet_light"), skip_prefixes=skip)
        assets = mano_assets or (None, None)
        self.mano_r = MANOHead(True, focal_length, img_res, assets[0])
        self.mano_l = MANOHead(False, focal_length, img_res, assets[1])
        # hand_regHead grid buffers (hand_head.py:24-28); only consumed by the unused 2-D read-out
        rng = torch.arange(32).float()
        vv, uu = torch.meshgrid(rng, rng, indexing="ij")
        self.regressor.hand_regHead.uu.copy_((uu + 0.5) / 32)
        self.regressor.hand_regHead.vv.copy_((vv + 0.5) / 32)
        self.img_res, self.focal_length = img_res, focal_length
        self._packed = None
        self._packed_dev = None
        self.engine = ConvEngine()
        # Numerics of this network (DESIGN.md "Conditioning note"): it amplifies ANY fp32 re-association -- the fp32 reference itself
        # sits a median 2.9e-7 m / at most 6.3e-7 m from an fp64 evaluation and moves by 2-5e-7 m with ATen's thread count -- so the
        # bar (1e-6 m against the reference, however it is run) needs a HIP path that is CLOSER to fp64 than the reference is.
        # Round 6 (tools/hon_error_stages.py, tools/experiments/hon_stage_budget_cpu.py, tools/hon_parity_ab.py):
        #   * where the error is made: the heat-map head (hand_head.py:75-94: res unit -> fc -> score -> spatial softmax) carries
        #     ~3/4 of the reference's own excess over an exact-accumulation evaluator, the MLPs (mano_head.py:190-207, KPE) most of
        #     the rest; the ResNet stages and the hourglass nothing.  Those two stages (2 % of the FLOPs) accumulate in fp64
        #     (acc64_stages -> HANDS_ACC_F64: v_mfma_f64_16x16x4_f64, correctly rounded outputs), the spatial softmax runs in fp64
        #     inside, and flash attention sums P V in 32-key blocks (its 1024-key chains were the whole FIT / SET excess);
        #   * everything else as in round 5: winograd_scope "all" (chains Cin long instead of 9 Cin) and engine.chain_limit = 64,
        #     chain_in_kernel (HANDS_SUM_BLOCK64) in every stage (taking the blocks out of the ResNet stages costs 0.74 -> 0.91).
        # Result: median err(HIP, fp64) / err(reference fp32, fp64) = 0.74 (1.02 in round 5), max 3.7e-7 m against fp64 (6.9e-7),
        # at -3 % throughput (32 samples).  An all-fp64 path reaches 0.59 at -55 %: DESIGN.md has the table.
        # Call invalidate_packed() after changing any of these.
        self.engine.winograd = True
        self.winograd_scope = "all"        # "all" | "backbone+fit" | "backbone" (trunk + FPN smoothing) | "trunk"
        self.engine.chain_limit, self.engine.chain_min_k, self.engine.chain_in_kernel = 64, 0, True
        self.block_stages = frozenset(STAGES)   # stages whose direct launches sum in engine.chain_limit blocks (the others: single chains)
        self.wino_stages = None            # None: winograd_scope decides; else the stages whose 3x3 / s1 layers take Winograd F(2x2)
        self.acc64_stages = frozenset(("reghead", "mlp"))   # stages (STAGES) whose convolutions / linear layers accumulate in fp64
        self.wino4_stages = frozenset()    # stages whose Winograd layers take F(4x4,3x3) instead of F(2x2) (needs engine.winograd4 = True;
                                           # 10-20x F(2x2)'s per-layer rounding: off -- tools/hon_parity_ab.py arm "+w4:resnet")
        self.acc64_3x3 = True              # False: the 3x3 layers of the fp64 stages keep their fp32 route (Winograd / blocked direct)
        self.small_map_splitk = False  # True / "deep" / "16x16": call-site constant split-K on maps of <= 8x8 pixels and on the
                                       # one-tile 16x16 layers (see _conv_fns).  Off since round 5: with three forwards in flight the
                                       # chip is filled by other forwards, and the reduce launches cost more than the slices gain
                                       # (tools/experiments/ab_small_map_splitk.py: +2.3 % at 32 samples, +2.9 % at 256); True shortens ONE
                                       # synchronous forward at 2 samples by 6 % (6.5 against 6.9 ms)
        self.chunks = 2   # a forward that runs ALONE (async_forward off) splits its 2*bz crops into this many jobs on separate HIP
                          # streams (1 = single stream); pipelined forwards run one job each
        self.async_forward = True   # two forwards in flight: call i runs on pipeline stream i & 1 and is joined at the first
                                    # use of its result (stream_xdict), so the launches of consecutive calls fill each other's
                                    # tails -- at 32 samples per GPU a launch is 1-4 tiles per CU, all in phase when alone
        self.pipeline_depth = "auto"  # forwards in flight in the pipelined mode: an int, or "auto" = 3 up to 64 samples per call, 2
                                    # above (round 5, one box, alternating: 32 samples 3833-3844 hands/s with 3 against 3738 with 2
                                    # and 3755 with 4; 256 samples 3786-3808 with 3 against 3832-3838 with 2) -- a launch schedule,
                                    # the arithmetic of a forward does not depend on it
        self._calls = 0
        self._pipe_done = {}
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate_packed())

    def _side_stream(self, dev, i):
        key = ("side_stream", i)
        st = self.__dict__.setdefault("_streams", {}).get(key)
        if st is None or st.device != dev:
            st = self.__dict__["_streams"][key] = torch.cuda.Stream(device=dev)
        return st

    def invalidate_packed(self):
        # forwards still in flight on the pipeline / side streams read the packed weights: wait for them before the tensors
        # are dropped (the caching allocator only knows the stream they were allocated on)
        for ev in self._pipe_done.values():
            ev.synchronize()
        self._pipe_done.clear()
        self._packed = None

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    # ---- packing ------------------------------------------------------------------------------
    @torch.no_grad()
    def _pack(self, dev):
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}

        def bn_affine(p, eps=BN_EPS):
            # scale / shift of an eval BatchNorm through the library's fp64 fold (w = 1 -> w_folded = scale)
            ones = torch.ones(sd[p + ".weight"].shape[0], 1, 1, 1)
            s, t = fold_bn(ones, sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"], eps)
            return s.view(-1), t

        def conv(p, stride=1, pad=0, bn=None, cin_pad_to=None):
            """conv (+ optional bias) followed by an optional eval-BatchNorm, folded."""
            w = sd[p + ".weight"].double()
            b = sd[p + ".bias"].double() if (p + ".bias") in sd else torch.zeros(w.shape[0], dtype=torch.float64)
            if bn is not None:
                s, t = bn_affine(bn)
                w, b = w * s.view(-1, 1, 1, 1), b * s + t
            sc = self.winograd_scope
            wino = stage_of(p) in self.wino_stages if self.wino_stages is not None else (sc == "all" or (sc == "trunk" and p.startswith("backbone.layer")) or
                    (sc in ("backbone", "backbone+fit") and p.startswith("backbone.")) or (sc == "backbone+fit" and p.startswith("FIT.")))
            pc = pack_conv(w, b, stride, pad, dev, cin_pad_to=cin_pad_to, winograd=wino,
                           winograd4=wino and stage_of(p) in self.wino4_stages)
            pc.acc64 = stage_of(p) in self.acc64_stages and (self.acc64_3x3 or w.shape[-1] == 1)
            pc.sum_block = -1 if stage_of(p) in self.block_stages else 0
            return pc

        def lin(p, **kw):
            pc = pack_linear(sd[p + ".weight"], sd[p + ".bias"], dev, **kw)
            pc.acc64 = stage_of(p) in self.acc64_stages and not p.startswith("grasp")
            pc.sum_block = -1 if stage_of(p) in self.block_stages else 0
            return pc

        def preact(p):
            s, t = bn_affine(p)
            return s.float().to(dev), t.float().to(dev)

        P = {"stem": conv("backbone.layer0.0", 2, 3, bn="backbone.layer0.1", cin_pad_to=4), "layers": []}
        for li, n in enumerate((3, 4, 6, 3), start=1):
            blocks = []
            for bi in range(n):
                p = f"backbone.layer{li}.0.{bi}"
                stride = 2 if (bi == 0 and li > 1) else 1
                e = {"c1": conv(p + ".conv1", bn=p + ".bn1"), "c2": conv(p + ".conv2", stride, 1, bn=p + ".bn2"),
                     "c3": conv(p + ".conv3", bn=p + ".bn3")}
                if (p + ".downsample.0.weight") in sd:
                    e["ds"] = conv(p + ".downsample.0", stride, 0, bn=p + ".downsample.1")
                blocks.append(e)
            P["layers"].append(blocks)
        for name in ("toplayer", "latlayer1", "latlayer2", "latlayer3"):
            P[name] = conv("backbone." + name)
        P["smooth3"] = conv("backbone.smooth3", 1, 1)
        P["gate"] = conv("backbone.attention_module.spatial.conv", 1, 3, bn="backbone.attention_module.spatial.bn",
                         cin_pad_to=4)
        if self.pos_enc is not None:
            P["kpe0"], P["kpe2"] = lin("kpe.feat_mlp.0"), lin("kpe.feat_mlp.2")
        for T, inj in (("FIT", True), ("SET", False)):
            layers = []
            for i in range(2):
                p = f"{T}.layers.{i}"
                tokmaj = lambda e: e[0].permute(1, 2, 0).reshape(NTOK, CF).contiguous().to(dev)
                e = {"v": conv(p + ".encode_value"), "q": conv(p + ".encode_query"), "k": conv(p + ".encode_key"),
                     "qemb": tokmaj(sd[p + ".q_embedding"]), "kemb": tokmaj(sd[p + ".k_embedding"]),
                     "n2": (sd[p + ".norm2.weight"].to(dev), sd[p + ".norm2.bias"].to(dev)),
                     "fc1": lin(p + ".mlp.fc1"), "fc2": lin(p + ".mlp.fc2")}
                if inj:
                    e["q2"], e["k2"] = conv(p + ".encode_query2"), conv(p + ".encode_key2")
                layers.append(e)
            P[T] = layers
        P["fit_c10"], P["fit_c12"] = conv("FIT.conv1.0", 1, 1), conv("FIT.conv1.2", 1, 1)
        P["fit_c20"] = conv("FIT.conv2.0")

        def hg_unit(p):      # hand_head.py:152-190
            return {"pre": preact(p + ".bn1"), "c1": conv(p + ".conv1", bn=p + ".bn2"),
                    "c2": conv(p + ".conv2", 1, 1, bn=p + ".bn3"), "c3": conv(p + ".conv3")}

        def enc_unit(p):     # hand_head.py:117-149
            return {"pre": preact(p + ".bn"), "c1": conv(p + ".conv1", bn=p + ".bn1"),
                    "c2": conv(p + ".conv2", 1, 1, bn=p + ".bn2"), "c3": conv(p + ".conv3")}

        hp = "regressor.hand_regHead"
        P["hg"] = [[hg_unit(f"{hp}.hg.0.hg.{lvl}.{j}.0") for j in range(4 if lvl == 0 else 3)] for lvl in range(4)]
        P["res"] = hg_unit(hp + ".res.0.0")
        P["fc"] = conv(hp + ".fc.0.block.0", bn=hp + ".fc.0.block.1")
        P["score"] = conv(hp + ".score.0")
        P["betas"] = sd[hp + ".betas"].reshape(-1).contiguous().to(dev)
        ep = "regressor.hand_Encoder"
        P["hm_conv"] = conv(ep + ".heatmap_conv", cin_pad_to=32)
        P["enc_conv"] = conv(ep + ".encoding_conv")
        P["enc"] = [enc_unit(f"{ep}.reg.{i}") for i in range(8)]
        mp = "regressor.mano_regHead"
        # NCHW flatten of (B,256,2,2): reference column c*4 + hw; NHWC buffer column hw*256 + c
        col = [(k % 4) * 256 + (k // 4) for k in range(1024)]
        P["base0"] = lin(mp + ".mano_base_layer.0", col_index=col)
        P["base2"] = lin(mp + ".mano_base_layer.2")
        wd = torch.cat([sd[mp + ".pose_reg.weight"], sd[mp + ".shape_reg.weight"], sd[mp + ".cam_reg.weight"]], 0)
        bd = torch.cat([sd[mp + ".pose_reg.bias"], sd[mp + ".shape_reg.bias"], sd[mp + ".cam_reg.bias"]], 0)
        rows = list(range(96)) + [96 + i for i in range(10)] + [108 + i for i in range(3)]
        P["regs"] = pack_linear(wd, bd, dev, row_index=rows, n_total=112)
        P["regs"].acc64 = "mlp" in self.acc64_stages
        if self.use_grasp_loss:
            gcol = [144 + i for i in range(10)] + list(range(144))
            P["g0"] = lin("grasp_classifier.0", col_index=gcol, k_total=154)
            P["g2"], P["g4"] = lin("grasp_classifier.2"), lin("grasp_classifier.4")
            P["g6"] = lin("grasp_classifier.6", n_total=12)
        for side, head in (("mano_r", self.mano_r), ("mano_l", self.mano_l)):
            m = pack_mano(head.mano.asset(), dev)
            m["consts"] = mano_consts(m)
            P[side] = m
        return P

    def packed(self, dev):
        if self._packed is None or self._packed_dev != dev:
            self._packed = self._pack(dev)
            self._packed_dev = dev
        return self._packed

    # ---- forward ------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, inputs, meta_info, targets=None):
        L = _lib.lib()
        r_img = inputs["r_img"]
        dev = r_img.device
        if dev.type != "cuda":
            raise RuntimeError("hands_amd.HandOccNet runs on a HIP device only (no CPU fallback)")
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        r_img, l_img, K = f32(r_img), f32(inputs["l_img"]), f32(meta_info["intrinsics"])
        bz, c, Hin, Win = r_img.shape
        assert c == 3 and l_img.shape == r_img.shape and K.shape[1:] == (3, 3)
        B2 = 2 * bz
        P = self.packed(dev)
        main = torch.cuda.current_stream(dev)
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        # -- everything that READS the caller's tensors runs on the caller's stream: the caller may overwrite its inputs
        #    as soon as forward returns -----------------------------------------------------------------------------
        # model.py:66-70: resize to 256x256, cat(r, l) -> NHWC4
        S = 256
        x4 = new(B2, S, S, 4)
        for side, im in enumerate((r_img, l_img)):
            check(L.hands_resize_crop_nchw3_to_nhwc4_f32(ptr(im), ptr(x4, side * bz * S * S * 4), bz, Hin, Win, S, 0, S,
                                                         main.cuda_stream), "resize")
        # KPE angles (hamer_light/pos_emb.py:28-64, feat_dim 256): private copies
        if self.pos_enc is not None:
            center = torch.cat([f32(inputs["r_center_angle"]), f32(inputs["l_center_angle"])], 0)
            corner = torch.cat([f32(inputs["r_corner_angle"]), f32(inputs["l_corner_angle"])], 0)
        else:
            center = corner = torch.zeros(B2, 8, dtype=torch.float32, device=dev)      # unused
        dbg = self.__dict__.get("_debug")
        capturing = self.engine._capturing(L, main.cuda_stream)
        pipelined = bool(self.async_forward and self.engine.overlap and dbg is None and self.engine.hook is None and not capturing)
        if pipelined:
            depth = (3 if bz <= 64 else 2) if self.pipeline_depth == "auto" else max(1, int(self.pipeline_depth))
            par = self._calls % depth
            self._calls += 1
            st = self._side_stream(dev, f"pipe{par}")
            K = K.clone()
            ev = torch.cuda.Event()
            ev.record(main)
            st.wait_event(ev)
            for t in (x4, center, corner, K):
                t.record_stream(st)
        else:
            st = main
            if not capturing:                        # (GraphedForward drains the device before it captures)
                for evd in self._pipe_done.values():     # a synchronous call is ordered after every forward still in flight
                    main.wait_event(evd)
        with torch.cuda.stream(st):
            output = self._forward_body(L, P, dev, x4, center, corner, K, bz, pipelined)
        if not pipelined:
            return output
        ready = torch.cuda.Event()
        ready.record(st)
        self._pipe_done[par] = ready
        return stream_xdict(output, ready, dev)

    def _forward_body(self, L, P, dev, x4, center, corner, K, bz, pipelined=False):
        """Everything after the input resize (model.py:72-129), enqueued on torch's CURRENT stream."""
        B2, S = 2 * bz, 256
        stream = torch.cuda.current_stream(dev).cuda_stream
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)

        conv, hconv = _conv_fns(L, stream, new, self.engine, self.small_map_splitk)

        def pipeline(x4, center, corner, B2, stream):
            """Everything per crop, from the NHWC image to the 112-vector of the regressor; rows are
            independent, so the 2*bz crops may be cut into chunks that run on separate HIP streams."""
            npix = B2 * NTOK
            conv, hconv = _conv_fns(L, stream, new, self.engine, self.small_map_splitk)

            def group(*jobs):
                """Independent layers (pc, x, B, H, W[, act[, res[, pre]]]) -> their outputs; pointwise ones of one kernel class
                leave as ONE launch (ConvEngine.conv_group), the rest as their own."""
                if self.small_map_splitk:                      # the call-site split-K rules live in conv()
                    return [conv(j[0], j[1], j[2], j[3], j[4], *(j[5:6] or (ACT_NONE,)), res=(j[6] if len(j) > 6 else None),
                                 **({"pre": j[7]} if len(j) > 7 and j[7] is not None else {}))[0] for j in jobs]
                js = []
                for j in jobs:
                    pc, x, B_, H_, W_ = j[:5]
                    Ho = (H_ + 2 * pc.pad - pc.KH) // pc.stride + 1
                    Wo = (W_ + 2 * pc.pad - pc.KW) // pc.stride + 1
                    d = {"pc": pc, "x": x, "B": B_, "H": H_, "W": W_, "out": new(B_, Ho, Wo, pc.Cout), "relu": j[5] if len(j) > 5 else ACT_NONE}
                    if len(j) > 6 and j[6] is not None:
                        d["res"] = j[6]
                    if len(j) > 7 and j[7] is not None:
                        d["pre"] = j[7]
                    js.append(d)
                self.engine.conv_group(L, js, stream)
                return [d["out"] for d in js]

            if self.pos_enc is not None:
                enc = new(B2, P["kpe0"].Cin)
                check(L.hands_kpe_encode_f32(ptr(center), ptr(corner), ptr(enc), B2, P["kpe0"].Cin, self.n_freq, stream), "kpe")
                k1, _, _ = conv(P["kpe0"], enc, B2, 1, 1, ACT_RELU)
                kpe, _, _ = conv(P["kpe2"], k1, B2, 1, 1, ACT_RELU)
                kpe = kpe.view(B2, CF)
            else:
                # pos_enc None (model.py:74-89: no KPE term anywhere): a zero vector through the same kernels -- x + 0 == x
                kpe = torch.zeros(B2, CF, dtype=torch.float32, device=dev)
            # -- LeakyReLU ResNet-50 (backbone.py:44-53) ------------------------------------------------
            H, W = (S - 1) // 2 + 1, (S - 1) // 2 + 1
            Hp, Wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            cur = new(B2, Hp, Wp, 64)
            if self.engine.fuse_stem_pool:         # conv + BN + LeakyReLU + max-pool in one kernel (bit-identical)
                self.engine.stem_pool(L, P["stem"], x4, 0, cur, B2, S, S, ACT_LEAKY_RELU, stream)
            else:
                a, H, W = conv(P["stem"], x4, B2, S, S, ACT_LEAKY_RELU)
                check(L.hands_maxpool3x3s2_nhwc_f32(ptr(a), ptr(cur), B2, H, W, 64, stream), "maxpool")
            H, W = Hp, Wp
            stages = []
            for blocks in P["layers"]:
                for e in blocks:
                    if "ds" in e:        # conv1 and the downsample branch read the same map: one launch where their tiles agree
                        t1, ident = group((e["c1"], cur, B2, H, W, ACT_LEAKY_RELU), (e["ds"], cur, B2, H, W))
                    else:
                        t1, ident = conv(e["c1"], cur, B2, H, W, ACT_LEAKY_RELU)[0], cur
                    t2, H2, W2 = conv(e["c2"], t1, B2, H, W, ACT_LEAKY_RELU)
                    cur, _, _ = conv(e["c3"], t2, B2, H2, W2, ACT_LEAKY_RELU, res=ident)
                    H, W = H2, W2
                stages.append((cur, H, W))
            (c2, h2, w2), (c3, h3, w3), (c4, h4, w4), (c5, h5, w5) = stages
            # -- FPN top-down (backbone.py:54-62) ------------------------------------------------------
            def up_add(x, h, w, y, Hh, Ww):
                out = new(B2, Hh, Ww, CF)
                check(L.hands_upsample_bilinear_add_f32(ptr(x), ptr(y), ptr(out), B2, h, w, Hh, Ww, CF, stream), "up_add")
                return out
            # the top layer and the three laterals are independent of the top-down chain: one launch
            p5, l4, l3, l2 = group((P["toplayer"], c5, B2, h5, w5), (P["latlayer1"], c4, B2, h4, w4),
                                   (P["latlayer2"], c3, B2, h3, w3), (P["latlayer3"], c2, B2, h2, w2))
            p4 = up_add(p5, h5, w5, l4, h4, w4)
            p3 = up_add(p4, h4, w4, l3, h3, w3)
            p2 = up_add(p3, h3, w3, l2, h2, w2)
            p2, _, _ = conv(P["smooth3"], p2, B2, h2, w2)          # smooth2(p3) of the reference is dead code
            Hf, Wf = h2 // 2, w2 // 2
            assert Hf * Wf == NTOK
            pooled = new(B2, Hf, Wf, CF)
            check(L.hands_pool2x2_nhwc_f32(ptr(p2), ptr(pooled), B2, h2, w2, CF, 0, stream), "avgpool")
            # -- SpatialGate (cbam.py:72-82) ------------------------------------------------------------
            npix = B2 * NTOK
            comp = new(npix, 4)
            check(L.hands_channel_pool_f32(ptr(pooled), ptr(comp), npix, CF, stream), "channel_pool")
            logit, _, _ = conv(P["gate"], comp, B2, Hf, Wf)
            primary, secondary = new(npix, CF), new(npix, CF)
            check(L.hands_gate_apply_f32(ptr(pooled), ptr(logit), 4, ptr(primary), ptr(secondary), npix, CF, stream), "gate")

            # -- FIT / SET (transformer.py:26-35,117-157) ------------------------------------------------
            scale = float((CF // HEADS) ** -0.5)

            def block(e, query, key, injection):
                qe, ke = new(npix, CF), new(npix, CF)
                check(L.hands_add_embed2_f32(ptr(query), ptr(key), ptr(e["qemb"]), ptr(e["kemb"]), ptr(kpe), ptr(qe), ptr(ke),
                                             B2, NTOK, CF, stream), "add_embed")
                # the three (five) projections of a block: one launch
                if injection:
                    v, q, k, q2, k2 = group((e["v"], key, npix, 1, 1), (e["q"], qe, npix, 1, 1), (e["k"], ke, npix, 1, 1),
                                            (e["q2"], qe, npix, 1, 1), (e["k2"], ke, npix, 1, 1))
                else:
                    v, q, k = group((e["v"], key, npix, 1, 1), (e["q"], qe, npix, 1, 1), (e["k"], ke, npix, 1, 1))
                x = new(npix, CF)
                if injection:
                    k2sum = new(B2, CF)
                    check(L.hands_token_sum_f32(ptr(k2), ptr(k2sum), B2, NTOK, CF, stream), "token_sum")
                    check(L.hands_flash_attention_f32(ptr(q), ptr(k), ptr(v), ptr(q2), ptr(k2sum), None, ptr(x), B2, NTOK,
                                                      HEADS, CF // HEADS, scale, stream), "flash_attention")
                else:
                    check(L.hands_flash_attention_f32(ptr(q), ptr(k), ptr(v), None, None, ptr(query), ptr(x), B2, NTOK,
                                                      HEADS, CF // HEADS, scale, stream), "flash_attention")
                y = new(npix, CF)
                check(L.hands_layernorm_f32(ptr(x), ptr(e["n2"][0]), ptr(e["n2"][1]), ptr(y), None, 1, npix, CF, 1e-5, stream),
                      "layernorm")
                hdn = conv(e["fc1"], y, npix, 1, 1, ACT_GELU)[0]
                conv(e["fc2"], hdn, npix, 1, 1, res=x, out=x)
                return x

            out = secondary
            for e in P["FIT"]:
                out = block(e, out, primary, True)
            cat = torch.cat([primary.view(npix, CF), out.view(npix, CF)], dim=1)          # (npix, 512)
            c20 = conv(P["fit_c20"], cat, B2, Hf, Wf)[0]
            c10 = conv(P["fit_c10"], cat, B2, Hf, Wf, ACT_RELU)[0]
            feats = conv(P["fit_c12"], c10, B2, Hf, Wf, res=c20)[0].view(npix, CF)
            if dbg is not None:
                dbg.update(primary=primary, secondary=secondary, fit=feats, c5=c5, pooled=pooled)
            key = feats
            out = feats
            for e in P["SET"]:
                out = block(e, out, key, False)
            feats = new(npix, CF)
            check(L.hands_add_rowvec_f32(ptr(out), ptr(kpe), ptr(feats), B2, NTOK, CF, stream), "add_kpe")   # model.py:88-89

            # -- regressor (hand_head.py, mano_head.py:190-207) -------------------------------------------
            def unit(u, x, H, W):
                n = B2 * H * W
                if self.engine.fuse_pre and self.engine.math == "fp32":
                    # BatchNorm -> LeakyReLU of the pre-activation unit applied to conv1's operand on its way into LDS
                    t1 = conv(u["c1"], x, B2, H, W, ACT_LEAKY_RELU, pre=u["pre"])[0]
                else:
                    t0 = new(n, CF)
                    check(L.hands_bn_leaky_f32(ptr(x), ptr(u["pre"][0]), ptr(u["pre"][1]), ptr(t0), n, CF, stream), "bn_leaky")
                    t1 = conv(u["c1"], t0, B2, H, W, ACT_LEAKY_RELU)[0]
                t2 = conv(u["c2"], t1, B2, H, W, ACT_LEAKY_RELU)[0]
                return conv(u["c3"], t2, B2, H, W, res=x)[0]

            def pool(x, H, W, mode):
                o = new(B2, H // 2, W // 2, CF)
                check(L.hands_pool2x2_nhwc_f32(ptr(x), ptr(o), B2, H, W, CF, mode, stream), "pool2x2")
                return o

            def unit2(ua, xa, Ha, Wa, ub, xb, Hb, Wb):
                """Two independent pre-activation units (the up and low branches of an hourglass level): their conv1 pair and their
                conv3 pair as one launch each, the 3x3 layers on their own."""
                if not (self.engine.fuse_pre and self.engine.math == "fp32"):
                    return unit(ua, xa, Ha, Wa), unit(ub, xb, Hb, Wb)
                t1a, t1b = group((ua["c1"], xa, B2, Ha, Wa, ACT_LEAKY_RELU, None, ua["pre"]),
                                 (ub["c1"], xb, B2, Hb, Wb, ACT_LEAKY_RELU, None, ub["pre"]))
                t2a = conv(ua["c2"], t1a, B2, Ha, Wa, ACT_LEAKY_RELU)[0]
                t2b = conv(ub["c2"], t1b, B2, Hb, Wb, ACT_LEAKY_RELU)[0]
                return group((ua["c3"], t2a, B2, Ha, Wa, ACT_NONE, xa), (ub["c3"], t2b, B2, Hb, Wb, ACT_NONE, xb))

            def hourglass(n, x, H, W):                                  # hand_head.py:217-235
                lv = P["hg"][n - 1]
                up1, low1 = unit2(lv[0], x, H, W, lv[1], pool(x, H, W, 1), H // 2, W // 2)
                low2 = hourglass(n - 1, low1, H // 2, W // 2) if n > 1 else unit(lv[3], low1, H // 2, W // 2)
                low3 = unit(lv[2], low2, H // 2, W // 2)
                o = new(B2, H, W, CF)
                check(L.hands_upsample_nearest2x_add_f32(ptr(low3), ptr(up1), ptr(o), B2, H // 2, W // 2, CF, stream), "up2x")
                return o

            if dbg is not None:
                dbg["set"] = out
            y = hourglass(4, feats, Hf, Wf)
            if dbg is not None:
                dbg["hourglass"] = y
            y = unit(P["res"], y, Hf, Wf)
            y = conv(P["fc"], y, B2, Hf, Wf, ACT_LEAKY_RELU)[0]
            lat = conv(P["score"], y, B2, Hf, Wf)[0]                     # (B2,32,32,24): 21 joints + pad
            heat = new(npix, 32)
            check(L.hands_spatial_softmax_f32(ptr(lat), P["score"].Cout, ptr(P["betas"]), ptr(heat), 32, B2, NTOK, 21, stream),
                  "spatial_softmax")
            if dbg is not None:
                dbg.update(heat=heat, lat=lat)
            hm = conv(P["hm_conv"], heat, B2, Hf, Wf)[0]
            x = conv(P["enc_conv"], y, B2, Hf, Wf, res=hm)[0]
            H, W = Hf, Wf
            for i in range(4):
                x = unit(P["enc"][2 * i], x, H, W)
                x = unit(P["enc"][2 * i + 1], x, H, W)
                x = pool(x, H, W, 1)
                H, W = H // 2, W // 2
            if dbg is not None:
                dbg["enc"] = x
            f = hconv(P["base0"], x, B2, 1, 1, ACT_LEAKY_RELU)[0]         # x: (B2,2,2,256) read as (B2,1024)
            f = hconv(P["base2"], f, B2, 1, 1, ACT_LEAKY_RELU)[0]
            pred = hconv(P["regs"], f, B2, 1, 1)[0].view(B2, 112)
            return pred

        dbg = self.__dict__.get("_debug")
        main = torch.cuda.current_stream(dev)
        # measured (bz=32 -> 64 crops): 2 chunks without split-K lose 8 % (smaller launches), with
        # latency_mode they gain 4 %; at 512 crops they gain 3 %
        # round 5, several forwards in flight (tools/experiments/ab_pipeline_depth.py): the other forwards fill the chip and one job per
        # forward is 0.6-1.4 % faster at 512 crops -- the crop jobs are for a forward that runs alone
        nch = self.chunks if (dbg is None and self.engine.overlap and not pipelined
                              and (B2 >= 128 or self.engine.latency_mode)) else 1
        nch = max(1, min(nch, B2))
        if nch == 1:
            pred = pipeline(x4, center, corner, B2, stream)
        else:
            # chunks of crops on separate streams: a launch is a few hundred workgroups on 256 CUs, and
            # the other chunk's workgroups fill its tail (same idea as HandsLight.trunk_chunks)
            ev0 = torch.cuda.Event()
            ev0.record(main)
            parts, done = [], []
            for ci in range(nch):
                lo, hi = ci * B2 // nch, (ci + 1) * B2 // nch
                st = main if ci == nch - 1 else self._side_stream(dev, ci)
                if st is not main:
                    st.wait_event(ev0)
                with torch.cuda.stream(st):
                    pc_ = pipeline(x4[lo:hi], center[lo:hi].contiguous(), corner[lo:hi].contiguous(), hi - lo, st.cuda_stream)
                if st is not main:
                    pc_.record_stream(main)
                    ev = torch.cuda.Event()
                    ev.record(st)
                    done.append(ev)
                parts.append(pc_)
            for ev in done:
                main.wait_event(ev)
            pred = torch.cat(parts, 0)
        rot = new(B2, 16, 3, 3)
        check(L.hands_rot6d_to_matrix_cols_f32(ptr(pred), 112, ptr(rot), B2, stream), "rot6d_cols")
        if dbg is not None:
            dbg["pred"] = pred
        shape = pred[:, 96:106].contiguous()
        cam = pred[:, 108:111].contiguous()
        # -- MANOHead x2, grasp (model.py:103-120) ------------------------------------------------------
        ws = {}

        def buf(name, numel):
            if name not in ws:
                ws[name] = new(numel)
            return ws[name]

        output = run_mano_heads(L, P["mano_r"], P["mano_l"], rot, shape, cam, cam, K, float(self.img_res), bz, stream, buf, self.engine)
        if not self.use_grasp_loss:
            return output
        gld = P["g0"].Cin
        gin = new(B2, gld)
        check(L.hands_grasp_input_f32(ptr(shape), 10, ptr(rot), ptr(shape), ptr(gin), B2, bz, 0, gld, stream), "grasp_in")
        g = hconv(P["g0"], gin, B2, 1, 1, ACT_RELU)[0]
        g = hconv(P["g2"], g, B2, 1, 1, ACT_RELU)[0]
        g = conv(P["g4"], g, B2, 1, 1, ACT_RELU)[0]
        g4 = conv(P["g6"], g, B2, 1, 1)[0].view(B2, 12)
        grasp = xdict()
        grasp["grasp.r"] = g4[:bz, :9].contiguous()
        grasp["grasp.l"] = g4[bz:, :9].contiguous()
        output.merge(grasp)
        return output
