"""Deterministic weight recipe shared by tests, fixtures, ``smoke()`` and ``bench.py``.

There is no network, so neither the torchvision ResNet-50 weights the reference downloads
(reference: src/nets/backbone/resnet.py:286-291) nor a trained checkpoint exist here. Every
``state_dict`` entry is instead filled from a generator seeded by the CRC-32 of its key, so the
reference model (imported under shims when the golden fixtures were generated) and this package's
``HandsLight`` receive bit-identical parameters without 288 MB of weights being committed.

Scales keep activations O(1) through 16 residual blocks (the last BN of every bottleneck is damped)
so that absolute tolerances on metre-valued outputs are meaningful.
"""
from __future__ import annotations

import math
import zlib

import torch


def _gen(key: str) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed(zlib.crc32(key.encode()))
    return g


def _is_bn(key: str) -> bool:
    parts = key.split(".")
    leaf_parent = parts[-2] if len(parts) >= 2 else ""
    if leaf_parent.startswith("bn"):
        return True
    # torchvision naming: downsample.0 = conv, downsample.1 = BN
    if len(parts) >= 3 and parts[-3] == "downsample" and leaf_parent == "1":
        return True
    # handoccnet_light: backbone.layer0 = Sequential(conv1, bn1, ...), BasicBlock.block = Sequential(conv, bn, ..)
    return len(parts) >= 3 and leaf_parent == "1" and parts[-3] in ("layer0", "block")


def recipe_tensor(key: str, ref: torch.Tensor) -> torch.Tensor | None:
    """Value for ``state_dict[key]`` (same shape/dtype as ``ref``); ``None`` = leave untouched."""
    if ".mano." in key or key.startswith("mano_") and ".mano" in key:
        return None  # MANO buffers come from the asset, not from the recipe
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros_like(ref)
    g = _gen(key)
    shape = tuple(ref.shape)
    parts = key.split(".")
    if leaf.startswith("init_"):
        return None  # hamer mean-parameter buffers come from mano_mean_params, not from the recipe
    if leaf == "pos_embed":
        return (0.2 * torch.randn(shape, generator=g)).to(ref.dtype)
    if leaf == "pos_embedding":
        return torch.randn(shape, generator=g).to(ref.dtype)
    if leaf in ("q_embedding", "k_embedding"):          # handoccnet FIT/SET learned 256x32x32 embeddings
        return (0.5 * torch.randn(shape, generator=g)).to(ref.dtype)
    if leaf == "betas":                                 # hand_regHead spatial-softmax temperatures (21,1)
        return (1.0 + 0.2 * torch.randn(shape, generator=g)).to(ref.dtype)
    if leaf in ("uu", "vv"):
        return None
    if len(parts) >= 2 and parts[-2] in ("norm", "norm1", "norm2", "last_norm"):   # LayerNorm
        if leaf == "weight":
            return (1.0 + 0.1 * torch.randn(shape, generator=g)).to(ref.dtype)
        return (0.1 * torch.randn(shape, generator=g)).to(ref.dtype)
    if _is_bn(key):
        if leaf == "weight":
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
            if ".bn3." in key:
                w = 0.25 * w  # damp the residual branch: keeps the trunk O(1) over 16 blocks
            return w.to(ref.dtype)
        if leaf == "bias":
            return (0.1 * torch.randn(shape, generator=g)).to(ref.dtype)
        if leaf == "running_mean":
            return (0.1 * torch.randn(shape, generator=g)).to(ref.dtype)
        if leaf == "running_var":
            return (1.0 + 0.1 * torch.rand(shape, generator=g)).to(ref.dtype)
        return None
    if leaf == "weight" and ref.ndim >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        w = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        if ".decoders." in key:
            # the reference initialises the decoders with xavier gain 0.01 (hmr_layer.py:62-65): the
            # regressed vectors move mostly through the biases below, which keeps the outputs'
            # sensitivity to fp32 summation order at the level of a trained model
            w = 0.02 * w
        if key.endswith("cam_init.4.weight"):
            w = 0.1 * w
        if key.endswith(("attn.proj.weight", "mlp.fc2.weight", "to_out.0.weight", "fn.net.3.weight")):
            w = 0.5 * w    # transformer residual branches (hamer_light)
        # handoccnet_light: keep FIT/SET, hourglass and encoder activations O(1-10)
        if key.startswith("regressor.") and key.endswith(".conv3.weight"):
            w = 0.2 * w          # residual branches of the 13 + 1 + 8 pre-activation units
        if key.endswith(("encode_value.weight", "encoding_conv.weight", "heatmap_conv.weight")):
            w = 0.3 * w
        if key.endswith(("encode_query.weight", "encode_key.weight")):
            w = 0.05 * w         # attention logits O(1-10) as in a trained net (He-scale q,k give one-hot softmax)
        if key.endswith(("encode_query2.weight", "encode_key2.weight")):
            w = 0.01 * w         # the FIT gate logit is a SUM over 1024 keys (transformer.py:88): keep it O(1)
        if key.startswith(("FIT.conv", "SET.conv")):
            w = 0.5 * w
        if key.endswith(("decpose.weight", "decshape.weight", "deccam.weight", "pose_reg.weight",
                         "shape_reg.weight", "cam_reg.weight")):
            w = 0.02 * w
        if key.startswith("feature_conv.0") or key.startswith("grasp_classifier.0"):
            w = 0.25 * w  # inputs are sums of O(1) feature maps (crop+glb, 49-pixel sum-pool)
        return w.to(ref.dtype)
    if leaf == "bias":
        b = 0.01 * torch.randn(shape, generator=g)
        if ".decoders.pose_6d." in key:
            b = 0.1 * torch.randn(shape, generator=g)   # x3 iterations: joint rotations of ~0.3 rad
        elif ".decoders.shape." in key:
            b = 0.3 * torch.randn(shape, generator=g)   # betas O(1)
        if key.endswith(("decpose.bias", "pose_reg.bias")):
            b = 0.3 * torch.randn(shape, generator=g)
            if key.endswith("pose_reg.bias"):
                # handoccnet regresses the 6D pose directly (no mean-pose init, mano_head.py:190-195):
                # centre it on the identity so that Gram-Schmidt stays well conditioned
                b = b + torch.tensor([1.0, 0, 0, 0, 1.0, 0]).repeat(shape[0] // 6)
        elif key.endswith(("decshape.bias", "shape_reg.bias")):
            b = 0.5 * torch.randn(shape, generator=g)
        elif key.endswith("cam_reg.bias"):
            b = b + torch.tensor([1.0, 0.0, 0.0])
        if key.endswith("cam_init.4.bias"):
            b = b + torch.tensor([1.0, 0.0, 0.0])  # weak-perspective scale near 1
        return b.to(ref.dtype)
    return None


@torch.no_grad()
def apply_recipe(module: torch.nn.Module) -> torch.nn.Module:
    sd = module.state_dict()
    for key in sorted(sd.keys()):
        val = recipe_tensor(key, sd[key])
        if val is not None:
            sd[key].copy_(val)
    if hasattr(module, "invalidate_packed"):     # parameters were written in place: drop the packed copies
        module.invalidate_packed()
    return module


def synthetic_mano_mean_params():
    """Stand-in for hamer's ``mano_mean_params.npz`` (pose (96,), shape (10,), cam (3,)); the real
    file lives under $DATA_DIR/hamer/_DATA/data (src/models/hamer_light/mano_head.py:49-56)."""
    import numpy as np
    return {"pose": np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), 16), "shape": np.zeros(10, np.float32),
            "cam": np.array([0.9, 0.0, 0.0], np.float32)}


def synthetic_inputs(bz: int, seed: int = 0, img_res: int = 224, device="cpu"):
    """Synthetic batch in the reference's input contract."""
    g = torch.Generator().manual_seed(seed)
    inputs = {
        "img": torch.randn(bz, 3, img_res, img_res, generator=g),
        "r_img": torch.randn(bz, 3, img_res, img_res, generator=g),
        "l_img": torch.randn(bz, 3, img_res, img_res, generator=g),
        "r_center_angle": 0.3 * torch.randn(bz, 2, generator=g),
        "l_center_angle": 0.3 * torch.randn(bz, 2, generator=g),
        "r_corner_angle": 0.3 * torch.randn(bz, 8, generator=g),
        "l_corner_angle": 0.3 * torch.randn(bz, 8, generator=g),
    }
    K = torch.tensor([[1000.0, 0.0, img_res / 2], [0.0, 1000.0, img_res / 2], [0.0, 0.0, 1.0]])
    meta_info = {
        "intrinsics": K[None].repeat(bz, 1, 1).contiguous(),
        "is_flipped": torch.zeros(bz, dtype=torch.long),
    }
    inputs = {k: v.to(device) for k, v in inputs.items()}
    meta_info = {k: v.to(device) for k, v in meta_info.items()}
    return inputs, meta_info


def synthetic_dense_inputs(bz: int, seed: int = 0, pos_enc: str = "dense", img_res: int = 224, device="cpu"):
    """The extra per-sample inputs of the non-default encodings, shaped as the reference's datasets build them
    (src/datasets/assembly_dataset.py:496-548, 569-681): ``{r,l}_dense_angle`` (bz, 2 | 6, img_res, img_res) -- the per-pixel
    viewing angles of the crop box (plus, for 'cam_conv', the centred pixel offsets and the normalised coordinates), zero outside
    the box -- with ``{r,l}_dense_mask`` (bz, img_res, img_res), and for 'pcl' the virtual-to-original rotations ``{r,l}_rot``."""
    g = torch.Generator().manual_seed(10_000 + seed)
    out = {}
    if pos_enc == "pcl":
        for side in "rl":
            w = 0.2 * torch.randn(bz, 3, generator=g)
            skew = torch.zeros(bz, 3, 3)
            skew[:, 0, 1], skew[:, 0, 2], skew[:, 1, 2] = -w[:, 2], w[:, 1], -w[:, 0]
            skew = skew - skew.transpose(1, 2)
            out[f"{side}_rot"] = torch.linalg.matrix_exp(skew.double()).float()
        return {k: v.to(device) for k, v in out.items()}
    f, c0 = 1000.0, img_res / 2
    for side in "rl":
        nch = 6 if pos_enc == "cam_conv" else 2
        ang = torch.zeros(bz, nch, img_res, img_res)
        msk = torch.zeros(bz, img_res, img_res)
        for b in range(bz):
            x0, y0 = (int(v) for v in torch.randint(0, 400, (2,), generator=g))
            w, h = (int(v) for v in torch.randint(100, img_res + 1, (2,), generator=g))
            xs = torch.arange(x0, x0 + w, dtype=torch.float64)[:, None].expand(w, h)
            ys = torch.arange(y0, y0 + h, dtype=torch.float64)[None, :].expand(w, h)
            maps = [torch.atan2(xs - c0, torch.tensor(f, dtype=torch.float64)), torch.atan2(ys - c0, torch.tensor(f, dtype=torch.float64))]
            if nch == 6:
                maps += [xs - c0, ys - c0, 2 * xs / img_res - 1, 2 * ys / img_res - 1]
            for i, m in enumerate(maps):
                ang[b, i, :w, :h] = m.float()
            msk[b, :w, :h] = 1.0
        out[f"{side}_dense_angle"], out[f"{side}_dense_mask"] = ang, msk
    return {k: v.to(device) for k, v in out.items()}
