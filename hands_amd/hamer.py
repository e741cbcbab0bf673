"""``HAMER`` -- the hamer_light forward path (ViT-H/16 + cross-attention MANO decoder) on MI355X.

Mirror of the reference ``HAMER(args, focal_length, img_res)`` (src/models/hamer_light/model.py:19-151):
same ``forward(inputs, meta_info) -> xdict`` contract, the same 22 output keys and the same 515
``state_dict`` tensors, every statement running as a gfx950 kernel of ``libhands_hip.so``:

    statement                                         kernel
    model.py:82-100  resize 224->256, cat, crop       hands_resize_crop_nchw3_to_nhwc4_f32
    vit.py:154-176   PatchEmbed conv16x16 s16 p2      hands_conv2d_nhwc_f32 (RGB0 path, K = 1024)
    pos_emb.py:28-64 KPE MLP                          hands_kpe_encode_f32 + 2 GEMMs
    vit.py:326-330   + pos_embed + kpe                hands_add_pos_f32
    vit.py:128-151   32 x Block                       hands_layernorm_f32, GEMM qkv, hands_attention_f32
                                                      (fp32 MFMA), GEMM proj (+residual), GEMM fc1 (GELU
                                                      epilogue), GEMM fc2 (+residual)
    vit.py:338, model.py:102-104 last_norm + kpe      hands_layernorm_f32 (fused add)
    mano_head.py:58-112 decoder, 6 layers             LN, GEMMs, hands_cross_attention_1q_f32
    geometry.py:47-62 rot6d (columns)                 hands_rot6d_to_matrix_cols_f32
    model.py:125-139 MANOHead x2, grasp MLP           shared with hands_light

The single decoder token attends only to itself in the self-attention (softmax over one key is
exactly 1), so that sub-layer is ``to_out(W_v LN(x)) + x`` -- bit-identical, two GEMMs.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, check, ptr
from .engine import ConvEngine, EngineSwitches
from .hands_light import MANOHead, _Args, mano_consts, run_mano_heads
from .packing import pack_conv, pack_linear, pack_mano
from .weights import synthetic_mano_mean_params
from .xdict import xdict

VIT_DIM, VIT_DEPTH, VIT_HEADS, VIT_HDIM = 1280, 32, 16, 80
DEC_DIM, DEC_DEPTH, DEC_HEADS, DEC_HDIM = 1024, 6, 8, 64
TOKENS_H, TOKENS_W = 16, 12


# ---- parameter containers (reference state_dict names; never called) -------------------------------
class _VitAttn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _VitMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _VitBlock(nn.Module):
    """vit.py:128-151."""

    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _VitAttn(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _VitMlp(dim, dim * 4)


class _PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=16, stride=16, padding=2)     # vit.py:168 (ratio=1 -> pad 2)
        self.patch_shape = (TOKENS_H, TOKENS_W)


class ViTParams(nn.Module):
    """vit.py:201-263 parameter layout of ViT-H/16 at 256x192."""

    def __init__(self):
        super().__init__()
        self.patch_embed = _PatchEmbed(VIT_DIM)
        self.pos_embed = nn.Parameter(torch.zeros(1, TOKENS_H * TOKENS_W + 1, VIT_DIM))
        self.blocks = nn.ModuleList([_VitBlock(VIT_DIM) for _ in range(VIT_DEPTH)])
        self.last_norm = nn.LayerNorm(VIT_DIM, eps=1e-6)


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class _SelfAttn(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0))


class _CrossAttn(nn.Module):
    def __init__(self, dim, inner, context_dim):
        super().__init__()
        self.to_kv = nn.Linear(context_dim, inner * 2, bias=False)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0))


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), nn.Linear(hidden, dim),
                                 nn.Dropout(0.0))


class _TransformerCrossAttn(nn.Module):
    def __init__(self):
        super().__init__()
        inner = DEC_HEADS * DEC_HDIM
        self.layers = nn.ModuleList([nn.ModuleList([
            _PreNorm(DEC_DIM, _SelfAttn(DEC_DIM, inner)),
            _PreNorm(DEC_DIM, _CrossAttn(DEC_DIM, inner, VIT_DIM)),
            _PreNorm(DEC_DIM, _FeedForward(DEC_DIM, DEC_DIM))]) for _ in range(DEC_DEPTH)])


class _TransformerDecoder(nn.Module):
    """pose_transformer.py:301-357."""

    def __init__(self):
        super().__init__()
        self.to_token_embedding = nn.Linear(1, DEC_DIM)
        self.pos_embedding = nn.Parameter(torch.randn(1, 1, DEC_DIM))
        self.transformer = _TransformerCrossAttn()


class MANOTransformerDecoderHead(nn.Module):
    """mano_head.py:12-56 parameter layout; mean parameters from mano_mean_params.npz."""

    def __init__(self, mean_params=None):
        super().__init__()
        self.transformer = _TransformerDecoder()
        self.decpose = nn.Linear(DEC_DIM, 96)
        self.decshape = nn.Linear(DEC_DIM, 10)
        self.deccam = nn.Linear(DEC_DIM, 3)
        mp = mean_params if mean_params is not None else load_mano_mean_params()
        self.register_buffer("init_hand_pose", torch.from_numpy(mp["pose"].astype(np.float32))[None])
        self.register_buffer("init_betas", torch.from_numpy(mp["shape"].astype(np.float32))[None])
        self.register_buffer("init_cam", torch.from_numpy(mp["cam"].astype(np.float32))[None])


class _KPE(nn.Module):
    """pos_emb.py:6-26."""

    def __init__(self, n_freq):
        super().__init__()
        self.feat_mlp = nn.Sequential(nn.Linear(20 * n_freq, VIT_DIM), nn.ReLU(inplace=True),
                                      nn.Linear(VIT_DIM, VIT_DIM), nn.ReLU(inplace=True))


def load_mano_mean_params(allow_synthetic=None):
    """$DATA_DIR/hamer/_DATA/data/mano_mean_params.npz (mano_head.py:49-50).  The reference fails when
    the file is missing; so does this unless the synthetic stand-in was asked for explicitly
    (argument or HANDS_SYNTHETIC_MANO=1, see hands_amd.mano.synthetic_allowed)."""
    from .mano import synthetic_allowed
    fn = os.path.join(os.environ.get("DATA_DIR", ""), "hamer", "_DATA", "data", "mano_mean_params.npz")
    if os.path.isfile(fn):
        d = np.load(fn)
        return {k: d[k] for k in ("pose", "shape", "cam")}
    if synthetic_allowed(allow_synthetic):
        return synthetic_mano_mean_params()
    raise FileNotFoundError(f"hands_amd: {fn!r} not found; set $DATA_DIR, pass mean_params=..., or opt in to the "
                            "synthetic mean parameters with HANDS_SYNTHETIC_MANO=1 (tests / benchmarks only)")


HAMER_DEFAULT_ARGS = _Args(pos_enc="center+corner_latent", n_freq_pos_enc=4, use_grasp_loss=True,
                           use_render_seg_loss=False, pretrained="none", img_res=224, focal_length=1000.0,
                           method="hamer_light")


class HAMER(EngineSwitches, nn.Module):
    def __init__(self, args=None, focal_length=1000.0, img_res=224, mano_assets=None, mean_params=None):
        super().__init__()
        args = args if args is not None else HAMER_DEFAULT_ARGS
        get = args.get if hasattr(args, "get") else (lambda k, d=None: getattr(args, k, d))
        self.args = args
        # built: pos_enc 'center+corner_latent' (shipped) or None (no KPE: model.py:91-97,102-104), grasp head on or off
        # (model.py:59-72,136-143).  'dense_latent' cannot run in the reference either: PositionalEncoding.forward calls
        # compute_dense_pos_enc(angle, mask) without its `size` argument (hamer_light/pos_emb.py:41 vs :66) -> TypeError.  The
        # renderer is not built
        if get("pos_enc") not in ("center+corner_latent", None) or get("use_render_seg_loss", False):
            raise NotImplementedError("hands_amd.HAMER: pos_enc must be 'center+corner_latent' or None, renderer off")
        self.n_freq = int(get("n_freq_pos_enc", 4))
        self.vit_input_size = (256, 192)
        self.backbone = ViTParams()
        self.mano_head = MANOTransformerDecoderHead(mean_params)
        assets = mano_assets or (None, None)
        self.mano_r = MANOHead(True, focal_length, img_res, assets[0])
        self.mano_l = MANOHead(False, focal_length, img_res, assets[1])
        self.pos_enc = get("pos_enc")
        if self.pos_enc is not None:
            self.kpe = _KPE(self.n_freq)
        self.use_grasp_loss = bool(get("use_grasp_loss", False))
        if self.use_grasp_loss:
            self.grasp_classifier = nn.Sequential(
                nn.Linear(10 + 144, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 512), nn.ReLU(inplace=True),
                nn.Linear(512, 128), nn.ReLU(inplace=True), nn.Linear(128, 9))
        self.img_res, self.focal_length = img_res, focal_length
        self._packed = None
        self._packed_dev = None
        self.engine = ConvEngine()
        self.chunks = 2   # the 2*bz crops run as this many jobs on separate HIP streams (1 = single stream)
        self._ws = {}
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate_packed())
        self._load_pretrained(get("pretrained", "vit"))

    def _load_pretrained(self, which):
        """model.py:33-44 with the reference's semantics: ``args.get('pretrained', 'vit')`` -- an args object
        WITHOUT the key asks for the ViTPose backbone; a missing ``$DATA_DIR`` / file raises as ``torch.load``
        does there; any other value (the configs use ``'none'``) keeps the initial weights.
        ``'vit'``:   ``$DATA_DIR/hamer_training_data/vitpose_backbone.pth['state_dict']`` into ``backbone``,
                     ``strict=False`` (the ViTPose file carries no ``kpe`` / decoder entries and may carry extra ones);
        ``'hamer'``: ``$DATA_DIR/hamer/_DATA/hamer_ckpts/checkpoints/hamer.ckpt['state_dict']``: the keys that contain
                     ``backbone`` / ``mano_head`` with that prefix removed, both loaded STRICTLY."""
        if which == "vit":
            fn = f"{os.environ['DATA_DIR']}/hamer_training_data/vitpose_backbone.pth"
            sd = torch.load(fn, map_location="cpu", weights_only=False)["state_dict"]
            self.backbone.load_state_dict(sd, strict=False)
        elif which == "hamer":
            fn = f"{os.environ['DATA_DIR']}/hamer/_DATA/hamer_ckpts/checkpoints/hamer.ckpt"
            sd = torch.load(fn, map_location="cpu", weights_only=False)["state_dict"]
            self.backbone.load_state_dict({k.replace("backbone.", ""): v for k, v in sd.items() if "backbone" in k})
            self.mano_head.load_state_dict({k.replace("mano_head.", ""): v for k, v in sd.items() if "mano_head" in k})
        self.invalidate_packed()

    def invalidate_packed(self):
        self._packed = None
        self._ws = {}

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def _pack(self, dev):
        cpu = lambda t: t.detach().cpu()
        lin = lambda m, **kw: pack_linear(cpu(m.weight), cpu(m.bias) if m.bias is not None else None, dev, **kw)
        vit = self.backbone
        P = {"patch": pack_conv(cpu(vit.patch_embed.proj.weight), cpu(vit.patch_embed.proj.bias), 16, 2, dev,
                                cin_pad_to=4),
             "pos": cpu(vit.pos_embed)[0].contiguous().to(dev),
             "blocks": [], "dec": []}
        if self.pos_enc is not None:
            P["kpe0"], P["kpe2"] = lin(self.kpe.feat_mlp[0]), lin(self.kpe.feat_mlp[2])
        ln = lambda m: (cpu(m.weight).to(dev), cpu(m.bias).to(dev))
        for blk in vit.blocks:
            P["blocks"].append({"n1": ln(blk.norm1), "qkv": lin(blk.attn.qkv), "proj": lin(blk.attn.proj),
                                "n2": ln(blk.norm2), "fc1": lin(blk.mlp.fc1), "fc2": lin(blk.mlp.fc2)})
        P["last"] = ln(vit.last_norm)
        td = self.mano_head.transformer
        inner = DEC_HEADS * DEC_HDIM
        for sa, ca, ff in td.transformer.layers:
            wv = cpu(sa.fn.to_qkv.weight)[2 * inner:3 * inner]          # only V matters for one token
            P["dec"].append({
                "n0": ln(sa.norm), "v": pack_linear(wv, None, dev), "o0": lin(sa.fn.to_out[0]),
                "n1": ln(ca.norm), "q": lin(ca.fn.to_q), "kv": lin(ca.fn.to_kv), "o1": lin(ca.fn.to_out[0]),
                "n2": ln(ff.norm), "f0": lin(ff.fn.net[0]), "f3": lin(ff.fn.net[3])})
        # token = Linear(1,1024)(0) + pos_embedding  (mano_head.py:80, pose_transformer.py:352-355)
        tok0 = (torch.zeros(1, 1) @ cpu(td.to_token_embedding.weight).T + cpu(td.to_token_embedding.bias)) \
            + cpu(td.pos_embedding)[0, :1]
        P["tok0"] = tok0.to(dev)
        mh = self.mano_head
        wd = torch.cat([cpu(mh.decpose.weight), cpu(mh.decshape.weight), cpu(mh.deccam.weight)], 0)
        bd = torch.cat([cpu(mh.decpose.bias), cpu(mh.decshape.bias), cpu(mh.deccam.bias)], 0)
        rows = list(range(96)) + [96 + i for i in range(10)] + [108 + i for i in range(3)]
        P["decout"] = pack_linear(wd, bd, dev, row_index=rows, n_total=112)
        init = torch.zeros(112)
        init[:96], init[96:106], init[108:111] = cpu(mh.init_hand_pose)[0], cpu(mh.init_betas)[0], cpu(mh.init_cam)[0]
        P["init"] = init.to(dev)
        if self.use_grasp_loss:
            g = self.grasp_classifier
            gcol = [144 + i for i in range(10)] + list(range(144))          # reference cat([shape, pose])
            P["g0"] = pack_linear(cpu(g[0].weight), cpu(g[0].bias), dev, col_index=gcol, k_total=154)
            P["g2"], P["g4"] = lin(g[2]), lin(g[4])
            P["g6"] = lin(g[6], n_total=12)
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

    def _side_stream(self, dev, i):
        key = ("side_stream", i)
        st = self._ws.get(key)
        if st is None or st.device != dev:
            st = self._ws[key] = torch.cuda.Stream(device=dev)
        return st

    def _buf(self, name, numel, dev):
        t = self._ws.get(name)
        if t is None or t.numel() < numel or t.device != dev:
            t = torch.empty(numel, dtype=torch.float32, device=dev)
            self._ws[name] = t
        return t

    @torch.no_grad()
    def forward(self, inputs, meta_info, targets=None):
        L = _lib.lib()
        r_img = inputs["r_img"]
        dev = r_img.device
        if dev.type != "cuda":
            raise RuntimeError("hands_amd.HAMER runs on a HIP device only (no CPU fallback)")
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        r_img, l_img, K = f32(r_img), f32(inputs["l_img"]), f32(meta_info["intrinsics"])
        bz, c, Hin, Win = r_img.shape
        assert c == 3 and l_img.shape == r_img.shape and K.shape[1:] == (3, 3)
        B2, T, Cd = 2 * bz, TOKENS_H * TOKENS_W, VIT_DIM
        M = B2 * T
        P = self.packed(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        buf = lambda n, numel: self._buf(n, numel, dev)
        gemm = lambda pc, x, rows, out, act=ACT_NONE, **kw: self.engine.conv(L, pc, x, rows, 1, 1, out, act, stream, **kw)
        hgemm = lambda *a, **kw: gemm(*a, splitk=True, **kw)     # per-sample rows: latency-bound head GEMMs
        lnorm = lambda x, gb, out, rows, Cc, eps, addvec=None, rpv=1: check(
            L.hands_layernorm_f32(ptr(x), ptr(gb[0]), ptr(gb[1]), ptr(out), ptr(addvec), rpv, rows, Cc, eps, stream),
            "layernorm")

        # -- model.py:82-100: resize to 256x256, cat(r,l), keep columns 32..223 -> NHWC4 ----------
        S, Wc = max(self.vit_input_size), min(self.vit_input_size)
        x4 = buf("x4", B2 * S * Wc * 4)
        for side, im in enumerate((r_img, l_img)):
            check(L.hands_resize_crop_nchw3_to_nhwc4_f32(ptr(im), ptr(x4, side * bz * S * Wc * 4), bz, Hin, Win, S,
                                                         (S - Wc) // 2, Wc, stream), "resize_crop")
        # -- KPE embedding (pos_emb.py:28-64) -------------------------------------------------------
        kpe = None
        if self.pos_enc is not None:
            center = torch.cat([f32(inputs["r_center_angle"]), f32(inputs["l_center_angle"])], 0)
            corner = torch.cat([f32(inputs["r_corner_angle"]), f32(inputs["l_corner_angle"])], 0)
            kld = P["kpe0"].Cin
            enc, k1, kpe = buf("kpe_enc", B2 * kld), buf("kpe_h", B2 * Cd), buf("kpe", B2 * Cd)
            check(L.hands_kpe_encode_f32(ptr(center), ptr(corner), ptr(enc), B2, kld, self.n_freq, stream), "kpe_encode")
            hgemm(P["kpe0"], enc, B2, k1, ACT_RELU)
            hgemm(P["kpe2"], k1, B2, kpe, ACT_RELU)
        # -- ViT-H/16 + decoder head, per chunk of crops -----------------------------------------------
        pred = torch.empty(B2, 112, device=dev)
        inner = DEC_HEADS * DEC_HDIM
        scale = float(VIT_HDIM ** -0.5)
        dscale = float(DEC_HDIM ** -0.5)

        def pipeline(lo, nB, st, tag):
            """Crops [lo, lo+nB): everything from the patch embedding to the 112-vector of the decoder.
            Rows are independent, so chunks of crops run on separate HIP streams: the attention /
            layer-norm launches of one chunk overlap the GEMMs of the other."""
            sh = st.cuda_stream
            Mc = nB * T
            cbuf = lambda n, numel: self._buf(f"{n}@{tag}", numel, dev)
            gemm = lambda pc, x, rows, out, act=ACT_NONE, **kw: self.engine.conv(L, pc, x, rows, 1, 1, out, act, sh, **kw)
            hgemm = lambda *a, **kw: gemm(*a, splitk=True, **kw)
            lnorm = lambda x, gb, out, rows, Cc, eps, addvec=None, rpv=1: check(
                L.hands_layernorm_f32(ptr(x), ptr(gb[0]), ptr(gb[1]), ptr(out), addvec, rpv, rows, Cc, eps, sh), "layernorm")
            # ViT-H/16 (vit.py:320-342)
            x = cbuf("vit_x", Mc * Cd)
            ho, wo = self.engine.conv(L, P["patch"], x4, nB, S, Wc, x, ACT_NONE, sh, x_off=lo * S * Wc * 4)
            assert (ho, wo) == (TOKENS_H, TOKENS_W)
            check(L.hands_add_pos_f32(ptr(x), ptr(P["pos"]), ptr(kpe, lo * Cd) if kpe is not None else None, nB, T, Cd, sh), "add_pos")
            y, qkv, att, hid = cbuf("vit_y", Mc * Cd), cbuf("vit_qkv", Mc * 3 * Cd), cbuf("vit_att", Mc * Cd), cbuf("vit_h", Mc * 4 * Cd)
            for blk in P["blocks"]:
                lnorm(x, blk["n1"], y, Mc, Cd, 1e-6)
                gemm(blk["qkv"], y, Mc, qkv)
                check(L.hands_attention_f32(ptr(qkv), ptr(att), nB, T, VIT_HEADS, VIT_HDIM, scale, sh), "attention")
                gemm(blk["proj"], att, Mc, x, res=x)
                lnorm(x, blk["n2"], y, Mc, Cd, 1e-6)
                gemm(blk["fc1"], y, Mc, hid, ACT_GELU)
                gemm(blk["fc2"], hid, Mc, x, res=x)
            feat = cbuf("vit_feat", Mc * Cd)
            lnorm(x, P["last"], feat, Mc, Cd, 1e-6, addvec=ptr(kpe, lo * Cd) if kpe is not None else None, rpv=T)   # last_norm, + kpe (model.py:102-104)
            # decoder head (mano_head.py:58-112)
            xd = cbuf("dec_x", nB * DEC_DIM)
            xd[: nB * DEC_DIM].view(nB, DEC_DIM).copy_(P["tok0"].expand(nB, DEC_DIM))
            yd, v512, q512, o512 = cbuf("dec_y", nB * DEC_DIM), cbuf("dec_v", nB * inner), cbuf("dec_q", nB * inner), cbuf("dec_o", nB * inner)
            kv, hd = cbuf("dec_kv", Mc * 2 * inner), cbuf("dec_h", nB * DEC_DIM)
            for lay in P["dec"]:
                lnorm(xd, lay["n0"], yd, nB, DEC_DIM, 1e-5)
                hgemm(lay["v"], yd, nB, v512)
                hgemm(lay["o0"], v512, nB, xd, res=xd)
                lnorm(xd, lay["n1"], yd, nB, DEC_DIM, 1e-5)
                hgemm(lay["q"], yd, nB, q512)
                gemm(lay["kv"], feat, Mc, kv)
                check(L.hands_cross_attention_1q_f32(ptr(q512), ptr(kv), ptr(o512), nB, T, DEC_HEADS, DEC_HDIM, dscale, sh),
                      "cross_attention")
                hgemm(lay["o1"], o512, nB, xd, res=xd)
                lnorm(xd, lay["n2"], yd, nB, DEC_DIM, 1e-5)
                hgemm(lay["f0"], yd, nB, hd, ACT_GELU)
                hgemm(lay["f3"], hd, nB, xd, res=xd)
            hgemm(P["decout"], xd, nB, pred, res=P["init"], res_ps=0, out_off=lo * 112)   # dec*(token) + mean params

        main = torch.cuda.current_stream(dev)
        nch = self.chunks if self.engine.overlap else 1
        nch = max(1, min(nch, B2))
        if nch == 1:
            pipeline(0, B2, main, "c0")
        else:
            ev0 = torch.cuda.Event()
            ev0.record(main)
            done = []
            for ci in range(nch):
                lo, hi = ci * B2 // nch, (ci + 1) * B2 // nch
                st = main if ci == nch - 1 else self._side_stream(dev, ci)
                if st is not main:
                    st.wait_event(ev0)
                with torch.cuda.stream(st):
                    pipeline(lo, hi - lo, st, f"c{ci}")
                if st is not main:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    done.append(ev)
            for ev in done:
                main.wait_event(ev)
        rot = torch.empty(B2, 16, 3, 3, device=dev)
        check(L.hands_rot6d_to_matrix_cols_f32(ptr(pred), 112, ptr(rot), B2, stream), "rot6d_cols")
        shape = pred[:, 96:106].contiguous()
        cam = pred[:, 108:111].contiguous()
        # -- MANOHead x2 (model.py:125-129) ----------------------------------------------------------
        output = run_mano_heads(L, P["mano_r"], P["mano_l"], rot, shape, cam, cam, K, float(self.img_res), bz,
                                stream, buf, self.engine)
        # -- grasp classifier (model.py:136-143) -----------------------------------------------------
        if not self.use_grasp_loss:
            return output
        gld = P["g0"].Cin
        gin = buf("grasp_in", B2 * gld)
        check(L.hands_grasp_input_f32(ptr(shape), 10, ptr(rot), ptr(shape), ptr(gin), B2, bz, 0, gld, stream),
              "grasp_input")
        g1, g2, g3 = buf("g1", B2 * 1024), buf("g2", B2 * 512), buf("g3", B2 * 128)
        g4 = torch.empty(B2, 12, device=dev)
        hgemm(P["g0"], gin, B2, g1, ACT_RELU)
        hgemm(P["g2"], g1, B2, g2, ACT_RELU)
        hgemm(P["g4"], g2, B2, g3, ACT_RELU)
        hgemm(P["g6"], g3, B2, g4)
        grasp = xdict()
        grasp["grasp.r"] = g4[:bz, :9].contiguous()
        grasp["grasp.l"] = g4[bz:, :9].contiguous()
        output.merge(grasp)
        return output
