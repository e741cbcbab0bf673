"""CPU oracle for the hamer_light forward path.  TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of ``HAMER.forward`` (reference: src/models/hamer_light/model.py:75-151) from a
flat ``state_dict`` with the reference's key names.  Pinned against fixtures produced by the imported
reference (tests/golden/make_golden_hamer.py -> tests/golden/hamer_light_*.npz); the MANO layer and
matrix->axis-angle are shared with oracle/hands_oracle.py (same pinning status: a9 unpinned).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import hands_oracle as O

VIT_DEPTH, VIT_DIM, VIT_HEADS = 32, 1280, 16
DEC_DEPTH, DEC_DIM, DEC_HEADS, DEC_DHEAD = 6, 1024, 8, 64


def _lin(x, sd, p, bias=True):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def kpe_embedding(inputs, prefix, sd, n_freq=4):
    """pos_emb.py:28-64: MLP(80 -> 1280 -> 1280, ReLU after both) of cat(center_enc, corner_enc)."""
    enc = torch.cat([O.pos_enc(inputs[prefix + "center_angle"], n_freq),
                     O.pos_enc(inputs[prefix + "corner_angle"], n_freq)], dim=1)
    enc = enc.to(sd["kpe.feat_mlp.0.weight"].dtype)      # fp64 when the oracle is run as its own numeric reference
    x = F.relu(_lin(enc, sd, "kpe.feat_mlp.0"))
    return F.relu(_lin(x, sd, "kpe.feat_mlp.2"))                  # (bz, 1280); repeated over tokens


def vit_attention(x, sd, p, heads=VIT_HEADS):
    """vit.py:89-126: q pre-scaled by head_dim**-0.5, softmax(q k^T) v, proj."""
    B, N, C = x.shape
    qkv = _lin(x, sd, p + ".qkv").reshape(B, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (C // heads) ** -0.5
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    return _lin(out, sd, p + ".proj")


def vit_forward(x_img, kpe_emb, sd, p="backbone", depth=VIT_DEPTH, return_blocks=False):
    """vit.py:320-342: PatchEmbed conv16x16 s16 pad 2 -> +pos_embed[:,1:]+pos_embed[:,:1] -> +kpe ->
    32 x Block (LN eps 1e-6) -> last_norm.  Returns tokens (B, 192, 1280) (= the reference's
    (B,C,Hp,Wp) map rearranged 'b c h w -> b (h w) c', mano_head.py:62)."""
    x = F.conv2d(x_img, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=16, padding=2)
    x = x.flatten(2).transpose(1, 2)
    pos = sd[p + ".pos_embed"]
    x = x + pos[:, 1:] + pos[:, :1]
    if kpe_emb is not None:                      # vit.py:329-330: only with a KPE embedding (model.py:91-97)
        x = x + kpe_emb[:, None, :]
    probes = []
    for i in range(depth):
        b = f"{p}.blocks.{i}"
        x = x + vit_attention(_ln(x, sd, b + ".norm1", 1e-6), sd, b + ".attn")
        h = F.gelu(_lin(_ln(x, sd, b + ".norm2", 1e-6), sd, b + ".mlp.fc1"))
        x = x + _lin(h, sd, b + ".mlp.fc2")
        if return_blocks:
            probes.append(x)
    x = _ln(x, sd, p + ".last_norm", 1e-6)
    return (x, probes) if return_blocks else x


def decoder_forward(context, sd, p="mano_head", depth=DEC_DEPTH):
    """mano_head.py:58-112 + pose_transformer.py:301-357,160-201,89-124: one zero token, 6 x
    {PreNorm self-attn, PreNorm cross-attn to the 192x1280 context, PreNorm FF(GELU)}."""
    B = context.shape[0]
    t = p + ".transformer"
    token = torch.zeros(B, 1, 1, dtype=context.dtype)          # (fp64 evaluations of the oracle: tools/hm_parity_ab.py)
    x = _lin(token, sd, t + ".to_token_embedding") + sd[t + ".pos_embedding"][:, :1]
    scale = DEC_DHEAD ** -0.5
    split = lambda z: z.view(B, -1, DEC_HEADS, DEC_DHEAD).transpose(1, 2)
    for i in range(depth):
        lp = f"{t}.transformer.layers.{i}"
        # self-attention (one token: softmax over a single key is exactly 1)
        y = _ln(x, sd, lp + ".0.norm", 1e-5)
        q, k, v = _lin(y, sd, lp + ".0.fn.to_qkv", bias=False).chunk(3, dim=-1)
        a = (torch.matmul(split(q), split(k).transpose(-1, -2)) * scale).softmax(dim=-1)
        o = torch.matmul(a, split(v)).transpose(1, 2).reshape(B, -1, DEC_HEADS * DEC_DHEAD)
        x = _lin(o, sd, lp + ".0.fn.to_out.0") + x
        # cross-attention (context is NOT normalised)
        y = _ln(x, sd, lp + ".1.norm", 1e-5)
        k, v = _lin(context, sd, lp + ".1.fn.to_kv", bias=False).chunk(2, dim=-1)
        q = _lin(y, sd, lp + ".1.fn.to_q", bias=False)
        a = (torch.matmul(split(q), split(k).transpose(-1, -2)) * scale).softmax(dim=-1)
        o = torch.matmul(a, split(v)).transpose(1, 2).reshape(B, -1, DEC_HEADS * DEC_DHEAD)
        x = _lin(o, sd, lp + ".1.fn.to_out.0") + x
        # feed-forward
        y = _ln(x, sd, lp + ".2.norm", 1e-5)
        x = _lin(F.gelu(_lin(y, sd, lp + ".2.fn.net.0")), sd, lp + ".2.fn.net.3") + x
    tok = x.squeeze(1)
    pose6d = _lin(tok, sd, p + ".decpose") + sd[p + ".init_hand_pose"]
    betas = _lin(tok, sd, p + ".decshape") + sd[p + ".init_betas"]
    cam = _lin(tok, sd, p + ".deccam") + sd[p + ".init_cam"]
    return pose6d, betas, cam, tok


def rot6d_to_rotmat_columns(x):
    """geometry.py:47-62: a1 = x[:3], a2 = x[3:], Gram-Schmidt, b1,b2,b3 stacked as COLUMNS."""
    x = x.reshape(-1, 2, 3).permute(0, 2, 1)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


def preprocess(inputs):
    """model.py:79-100: bilinear 224 -> 256 (align_corners=False), cat(r,l), keep columns 32..223."""
    r = F.interpolate(inputs["r_img"], size=256, mode="bilinear", align_corners=False)
    l = F.interpolate(inputs["l_img"], size=256, mode="bilinear", align_corners=False)
    return torch.cat([r, l], dim=0)[:, :, :, 32:-32]


@torch.no_grad()
def hamer_forward(sd, asset_r, asset_l, inputs, meta_info, img_res=224, n_freq=4, vit_depth=VIT_DEPTH,
                  return_intermediates=False, pos_enc="center+corner_latent", use_grasp_loss=True):
    K = meta_info["intrinsics"]
    bz = inputs["r_img"].shape[0]
    x = preprocess(inputs)
    kpe = None
    if pos_enc is not None:                                                      # model.py:91-97
        kpe = torch.cat([kpe_embedding(inputs, "r_", sd, n_freq), kpe_embedding(inputs, "l_", sd, n_freq)], 0)
    tokens, blocks = vit_forward(x, kpe, sd, depth=vit_depth, return_blocks=True)
    feats = tokens + kpe[:, None, :] if kpe is not None else tokens             # model.py:102-104
    pose6d, betas, cam, tok = decoder_forward(feats, sd)
    rotmat = rot6d_to_rotmat_columns(pose6d).view(2 * bz, 16, 3, 3)
    out = {}
    mr = O.mano_head(rotmat[:bz], betas[:bz], cam[:bz], K, asset_r, img_res, ".r")     # model.py:125
    ml = O.mano_head(rotmat[bz:], betas[bz:], cam[bz:], K, asset_l, img_res, ".l")     # model.py:126
    mr["cam_t.wp.init.r"] = cam[:bz]
    ml["cam_t.wp.init.l"] = cam[bz:]
    out.update({"mano." + k: v for k, v in mr.items()})
    out.update({"mano." + k: v for k, v in ml.items()})

    def grasp(shape, pose):                                                      # model.py:137-139
        g = torch.cat([shape, pose.reshape(bz, -1)], dim=1)
        for i in (0, 2, 4):
            g = F.relu(_lin(g, sd, f"grasp_classifier.{i}"))
        return _lin(g, sd, "grasp_classifier.6")

    if use_grasp_loss:                                                           # model.py:136-143
        out["grasp.r"] = grasp(betas[:bz], rotmat[:bz])
        out["grasp.l"] = grasp(betas[bz:], rotmat[bz:])
    if return_intermediates:
        return out, {"x": x, "kpe": kpe, "tokens": tokens, "blocks": blocks, "pose6d": pose6d,
                     "betas": betas, "cam": cam, "token_out": tok}
    return out
