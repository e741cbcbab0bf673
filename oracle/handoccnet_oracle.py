"""CPU oracle for the handoccnet_light forward path.  TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of ``HandOccNet.forward`` (reference: src/models/handoccnet_light/model.py:60-129)
from a flat ``state_dict`` with the reference's key names.  Pinned against fixtures produced by the
imported reference (tests/golden/make_golden_handoccnet.py); MANO / matrix->axis-angle shared with
oracle/hands_oracle.py (a9 unpinned).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import hands_oracle as O
from .hamer_oracle import rot6d_to_rotmat_columns

LEAK = 0.01


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=eps)


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


# ---- backbone.py:68-119,166-203: ResNet-50 with LeakyReLU(0.01), stride on the 3x3 --------------------
def res_bottleneck(x, sd, p, stride):
    out = F.leaky_relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1"), LEAK)
    out = F.leaky_relu(_bn(_conv(out, sd, p + ".conv2", stride, 1), sd, p + ".bn2"), LEAK)
    out = _bn(_conv(out, sd, p + ".conv3"), sd, p + ".bn3")
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(_conv(x, sd, p + ".downsample.0", stride), sd, p + ".downsample.1")
    return F.leaky_relu(out + res, LEAK)


def fpn(x, sd, p="backbone"):
    """backbone.py:44-65."""
    c1 = F.max_pool2d(F.leaky_relu(_bn(_conv(x, sd, p + ".layer0.0", 2, 3), sd, p + ".layer0.1"), LEAK), 3, 2, 1)
    feats = []
    c = c1
    for li, n in enumerate((3, 4, 6, 3), start=1):
        for bi in range(n):
            c = res_bottleneck(c, sd, f"{p}.layer{li}.0.{bi}", 2 if (bi == 0 and li > 1) else 1)
        feats.append(c)
    c2, c3, c4, c5 = feats
    up_add = lambda a, b: F.interpolate(a, size=b.shape[-2:], mode="bilinear", align_corners=False) + b
    p5 = _conv(c5, sd, p + ".toplayer")
    p4 = up_add(p5, _conv(c4, sd, p + ".latlayer1"))
    p3 = up_add(p4, _conv(c3, sd, p + ".latlayer2"))
    p2 = up_add(p3, _conv(c2, sd, p + ".latlayer3"))
    p2 = _conv(p2, sd, p + ".smooth3", 1, 1)          # smooth2(p3) is computed by the reference but unused
    p2s = p2
    p2 = F.avg_pool2d(p2, 2, 2)
    # SpatialGate (cbam.py:72-82): 7x7 conv on [max_c, mean_c] -> BN -> sigmoid
    comp = torch.cat([p2.max(1)[0].unsqueeze(1), p2.mean(1).unsqueeze(1)], dim=1)
    g = _bn(_conv(comp, sd, p + ".attention_module.spatial.conv", 1, 3), sd, p + ".attention_module.spatial.bn")
    scale = torch.sigmoid(g)
    return p2 * scale, p2 * (1 - scale), {"c5": c5, "p2_smooth": p2s}


# ---- transformer.py:71-157: FIT / SET blocks -----------------------------------------------------------
def attention(q, k, v, q2, k2, heads, use_sigmoid):
    B, N, C = q.shape
    sp = lambda t: t.reshape(B, N, heads, C // heads).permute(0, 2, 1, 3)
    scale = (C // heads) ** -0.5
    attn = (torch.matmul(sp(q), sp(k).transpose(-2, -1)) * scale).softmax(dim=-1)
    if use_sigmoid:
        a2 = torch.matmul(sp(q2), sp(k2).transpose(-2, -1)) * scale
        attn = attn * torch.sigmoid(a2.sum(dim=-1)).unsqueeze(3)
    return torch.matmul(attn, sp(v)).transpose(1, 2).reshape(B, N, C)


def block(query, key, kpe_map, sd, p, injection, heads=4):
    b, c, h, w = query.shape
    q_embed = query + sd[p + ".q_embedding"] + kpe_map
    k_embed = key + sd[p + ".k_embedding"] + kpe_map
    tok = lambda t: t.view(b, c, -1).permute(0, 2, 1)
    v = tok(_conv(key, sd, p + ".encode_value"))
    q = tok(_conv(q_embed, sd, p + ".encode_query"))
    k = tok(_conv(k_embed, sd, p + ".encode_key"))
    x = tok(query)
    if injection:
        q2 = tok(_conv(q_embed, sd, p + ".encode_query2"))
        k2 = tok(_conv(k_embed, sd, p + ".encode_key2"))
        x = attention(q, k, v, q2, k2, heads, True)
    else:
        x = x + attention(q, k, v, None, None, heads, False)
    y = F.layer_norm(x, (c,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    x = x + _lin(F.gelu(_lin(y, sd, p + ".mlp.fc1")), sd, p + ".mlp.fc2")
    return x.permute(0, 2, 1).contiguous().view(b, c, h, w)


def transformer(query, key, kpe_map, sd, p, injection, probes=None):
    out = query
    for i in range(2):
        out = block(out, key, kpe_map, sd, f"{p}.layers.{i}", injection)
        if probes is not None and i == 0:
            probes[p + "_block0"] = out
    if injection:
        cat = torch.cat([key, out], dim=1)
        c1 = _conv(F.relu(_conv(cat, sd, p + ".conv1.0", 1, 1)), sd, p + ".conv1.2", 1, 1)
        out = c1 + _conv(cat, sd, p + ".conv2.0")
    return out


# ---- hand_head.py: pre-activation residual units, hourglass, encoder -----------------------------------
def hg_bottleneck(x, sd, p):
    """hand_head.py:152-190 (expansion 2, no skip here)."""
    out = _conv(F.leaky_relu(_bn(x, sd, p + ".bn1"), LEAK), sd, p + ".conv1")
    out = _conv(F.leaky_relu(_bn(out, sd, p + ".bn2"), LEAK), sd, p + ".conv2", 1, 1)
    out = _conv(F.leaky_relu(_bn(out, sd, p + ".bn3"), LEAK), sd, p + ".conv3")
    return out + x


def hourglass(n, x, sd, p):
    """hand_head.py:217-235 (depth 4, one unit per residual module)."""
    up1 = hg_bottleneck(x, sd, f"{p}.hg.{n - 1}.0.0")
    low1 = hg_bottleneck(F.max_pool2d(x, 2, stride=2), sd, f"{p}.hg.{n - 1}.1.0")
    low2 = hourglass(n - 1, low1, sd, p) if n > 1 else hg_bottleneck(low1, sd, f"{p}.hg.{n - 1}.3.0")
    low3 = hg_bottleneck(low2, sd, f"{p}.hg.{n - 1}.2.0")
    return up1 + F.interpolate(low3, scale_factor=2)


def enc_residual(x, sd, p):
    """hand_head.py:117-149 (numIn == numOut)."""
    out = _conv(F.leaky_relu(_bn(x, sd, p + ".bn"), LEAK), sd, p + ".conv1")
    out = _conv(F.leaky_relu(_bn(out, sd, p + ".bn1"), LEAK), sd, p + ".conv2", 1, 1)
    out = _conv(F.leaky_relu(_bn(out, sd, p + ".bn2"), LEAK), sd, p + ".conv3")
    return out + x


def regressor(feats, sd, p="regressor"):
    """regressor.py:14-19, hand_head.py:75-94,266-280, mano_head.py:190-207."""
    hp = p + ".hand_regHead"
    y = hourglass(4, feats, sd, hp + ".hg.0")
    hg_out = y
    y = hg_bottleneck(y, sd, hp + ".res.0.0")
    y = F.leaky_relu(_bn(_conv(y, sd, hp + ".fc.0.block.0"), sd, hp + ".fc.0.block.1"), LEAK)
    lat = _conv(y, sd, hp + ".score.0")
    B = lat.shape[0]
    heat = (lat.view(B, 21, -1) * sd[hp + ".betas"]).softmax(dim=2).view(B, 21, 32, 32)
    ep = p + ".hand_Encoder"
    x = _conv(heat, sd, ep + ".heatmap_conv") + _conv(y, sd, ep + ".encoding_conv")
    for i in range(4):
        for j in range(2):
            x = enc_residual(x, sd, f"{ep}.reg.{i * 2 + j}")
        x = F.max_pool2d(x, 2, 2)
    enc = x.view(B, -1)
    mp = p + ".mano_regHead"
    f = F.leaky_relu(_lin(enc, sd, mp + ".mano_base_layer.0"), LEAK)
    f = F.leaky_relu(_lin(f, sd, mp + ".mano_base_layer.2"), LEAK)
    pose6d = _lin(f, sd, mp + ".pose_reg")
    rot = rot6d_to_rotmat_columns(pose6d.view(-1, 6)).view(-1, 16, 3, 3)
    return rot, _lin(f, sd, mp + ".shape_reg"), _lin(f, sd, mp + ".cam_reg"), \
        {"hourglass": hg_out, "heatmaps": heat, "mano_encoding": enc, "pose6d": pose6d}


def kpe_embedding(inputs, prefix, sd, n_freq=4):
    """hamer_light/pos_emb.py:28-64 with feat_dim = 256."""
    enc = torch.cat([O.pos_enc(inputs[prefix + "center_angle"], n_freq),
                     O.pos_enc(inputs[prefix + "corner_angle"], n_freq)], dim=1)
    enc = enc.to(sd["kpe.feat_mlp.0.weight"].dtype)      # fp64 when the oracle is run as its own numeric reference
    return F.relu(_lin(F.relu(_lin(enc, sd, "kpe.feat_mlp.0")), sd, "kpe.feat_mlp.2"))


@torch.no_grad()
def handoccnet_forward(sd, asset_r, asset_l, inputs, meta_info, img_res=224, n_freq=4, return_intermediates=False,
                       pos_enc="center+corner_latent", use_grasp_loss=True):
    K = meta_info["intrinsics"]
    bz = inputs["r_img"].shape[0]
    r = F.interpolate(inputs["r_img"], size=256, mode="bilinear", align_corners=False)
    l = F.interpolate(inputs["l_img"], size=256, mode="bilinear", align_corners=False)
    x = torch.cat([r, l], dim=0)
    if pos_enc is not None:
        kpe = torch.cat([kpe_embedding(inputs, "r_", sd, n_freq), kpe_embedding(inputs, "l_", sd, n_freq)], 0)
    else:                                                # model.py:74-89: no KPE term anywhere (x + 0 == x)
        kpe = torch.zeros(x.shape[0], 256, dtype=x.dtype)
    kpe_map = kpe[:, :, None, None]                      # constant over the 32x32 map (pos_emb.py:44)
    primary, secondary, inter = fpn(x, sd)
    probes = {}
    feats = transformer(secondary, primary, kpe_map, sd, "FIT", True, probes)      # model.py:85
    inter["fit"] = feats
    feats = transformer(feats, feats, kpe_map, sd, "SET", False, probes)           # model.py:86
    inter["set"] = feats
    feats = feats + kpe_map                                                        # model.py:88-89
    rot, shape, cam, rinter = regressor(feats, sd)
    inter.update(rinter)
    inter.update(primary=primary, secondary=secondary, fit_block0=probes["FIT_block0"], kpe=kpe)
    out = {}
    mr = O.mano_head(rot[:bz], shape[:bz], cam[:bz], K, asset_r, img_res, ".r")
    ml = O.mano_head(rot[bz:], shape[bz:], cam[bz:], K, asset_l, img_res, ".l")
    mr["cam_t.wp.init.r"] = cam[:bz]
    ml["cam_t.wp.init.l"] = cam[bz:]
    out.update({"mano." + k: v for k, v in mr.items()})
    out.update({"mano." + k: v for k, v in ml.items()})

    def grasp(sh, pose):
        g = torch.cat([sh, pose.reshape(bz, -1)], dim=1)
        for i in (0, 2, 4):
            g = F.relu(_lin(g, sd, f"grasp_classifier.{i}"))
        return _lin(g, sd, "grasp_classifier.6")

    if use_grasp_loss:                                   # model.py:113-120
        out["grasp.r"] = grasp(shape[:bz], rot[:bz])
        out["grasp.l"] = grasp(shape[bz:], rot[bz:])
    return (out, inter) if return_intermediates else out
