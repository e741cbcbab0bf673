"""INDEPENDENT pure-torch restatement of the packed layouts (test infrastructure; the product packs through
the C ABI, csrc/pack.cpp): tests/test_host.py checks that a weight packed by C equals this one bit for bit.

One-time host-side packing of reference-layout parameters into the layouts the HIP kernels read.

Not on the hot path: runs once per ``load_state_dict`` / device move.

* conv + eval BatchNorm2d -> folded weight/bias (reference applies them separately:
  src/nets/backbone/resnet.py:137-149);
* weights to ``[Cout_pad][Kpad]`` with k ordered (kh, kw, cin) -- see include/hands_hip.h;
* column permutations that keep every operand segment 16-byte aligned (HMR state row, grasp row,
  NCHW ``nn.Flatten`` order of feature_conv's Linear).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from hands_amd.mano import ManoAsset, TIP_IDS

BN_EPS = 1e-5


def _round_up(x, m):
    return (x + m - 1) // m * m


@dataclass
class PackedConv:
    w: torch.Tensor       # (Cout_pad, Kpad) fp32 device
    bias: torch.Tensor    # (Cout_pad,) fp32 device
    Cin: int              # channels the kernel sees (padded)
    Cout: int             # channels the kernel stores (multiple of 4)
    KH: int
    KW: int
    stride: int
    pad: int
    Kpad: int
    macs_per_pixel: int = 0   # ALGORITHMIC multiply-accumulates per output pixel (true, unpadded dims)


def fold_bn(w, bn_w, bn_b, bn_mean, bn_var, eps=BN_EPS):
    """(Cout,Cin,KH,KW) conv weight + BN stats -> folded weight, bias (fp64 math, fp32 result)."""
    # numpy's sqrt is correctly rounded (IEEE); torch.sqrt on float64 goes through MKL VML here and is off by
    # one ulp in ~1 % of the elements, so the comparison with the C packer uses the IEEE one
    scale = bn_w.double() / torch.from_numpy(np.sqrt(bn_var.double().numpy() + eps))
    wf = w.double() * scale.view(-1, 1, 1, 1)
    bf = bn_b.double() - bn_mean.double() * scale
    return wf, bf


def pack_conv(w, bias, stride, pad, device, cin_pad_to=None) -> PackedConv:
    """w: (Cout, Cin, KH, KW) (any float dtype, CPU); bias: (Cout,) or None."""
    Cout, Cin, KH, KW = w.shape
    Cin_p = cin_pad_to or Cin
    wk = torch.zeros(Cout, KH, KW, Cin_p, dtype=torch.float64)
    wk[..., :Cin] = w.double().permute(0, 2, 3, 1)
    K = KH * KW * Cin_p
    Kpad = _round_up(K, 16)
    Cout_s = _round_up(Cout, 4)
    Cout_pad = _round_up(Cout, 128)
    wp = torch.zeros(Cout_pad, Kpad, dtype=torch.float32)
    wp[:Cout, :K] = wk.reshape(Cout, K).float()
    bp = torch.zeros(Cout_pad, dtype=torch.float32)
    if bias is not None:
        bp[:Cout] = bias.float()
    return PackedConv(wp.to(device), bp.to(device), Cin_p, Cout_s, KH, KW, stride, pad, Kpad,
                      macs_per_pixel=Cout * Cin * KH * KW)


def pack_linear(w, bias, device, col_index=None, k_total=None, row_index=None, n_total=None) -> PackedConv:
    """nn.Linear weight (N, K) -> 1x1 'conv'.  ``col_index[k_ref] = k_packed`` places reference input
    column k_ref at packed column k_packed (row of width ``k_total``); ``row_index`` likewise for
    output rows (``n_total`` stored outputs)."""
    N, K = w.shape
    kt = k_total or K
    nt = n_total or N
    w2 = torch.zeros(nt, kt, dtype=torch.float64)
    ci = torch.arange(K) if col_index is None else torch.as_tensor(col_index)
    ri = torch.arange(N) if row_index is None else torch.as_tensor(row_index)
    tmp = torch.zeros(N, kt, dtype=torch.float64)
    tmp[:, ci] = w.double()
    w2[ri] = tmp
    b2 = torch.zeros(nt, dtype=torch.float64)
    if bias is not None:
        b2[ri] = bias.double()
    kp = _round_up(kt, 16)
    pc = pack_conv(w2.view(nt, kt, 1, 1), b2, 1, 0, device, cin_pad_to=kp)
    pc.macs_per_pixel = N * K
    return pc


# HMR state row layout (see hands_hmr_init_f32): feat | pose6d 96 | shape 10 | 2 pad | cam 3 | 1 pad
def hmr_state_columns(F):
    """packed column of each reference concat column [feat F, pose_6d 96, shape 10, cam 3]
    (src/nets/hmr_layer.py:80 with dict order from src/nets/hand_heads/hand_hmr.py:66-69)."""
    cols = list(range(F)) + [F + i for i in range(96)] + [F + 96 + i for i in range(10)] + \
           [F + 108 + i for i in range(3)]
    return cols


HMR_VEC = 112  # width of the vector part of the state row


def pack_mano(asset: ManoAsset, device):
    """MANO constants for the pose / blend-GEMM / skin kernels."""
    vt = asset.v_template.astype(np.float64)
    sd = asset.shapedirs.astype(np.float64)              # (778,3,10)
    Jr = asset.J_regressor.astype(np.float64)            # (16,778)
    J_template = (Jr @ vt).astype(np.float32)            # (16,3)
    J_shapedirs = np.einsum("jv,vck->jck", Jr, sd).reshape(48, 10).astype(np.float32)
    pose_mean = np.concatenate([np.zeros(3, np.float32), asset.hands_mean.astype(np.float32)])
    # blend matrix as a Linear weight (N=2334 outputs, K=146 inputs [beta | pose_feature | 1]): column 145 is v_template, which
    # hands_mano_heads_f32 multiplies with a constant 1 (include/hands_hip.h); the bias vector serves the three-launch chain
    Wb = np.concatenate([sd.reshape(-1, 10), asset.posedirs.astype(np.float64).T, vt.reshape(-1, 1)], axis=1)  # (2334,146)
    blend = pack_linear(torch.from_numpy(Wb), torch.from_numpy(vt.reshape(-1)), device,
                        k_total=146, n_total=2336)
    blend.macs_per_pixel = 2334 * 145                   # the algorithmic contraction (the template column is the bias)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return {
        "pose_mean": t(pose_mean), "J_template": t(J_template), "J_shapedirs": t(J_shapedirs),
        "lbs_weights": t(asset.lbs_weights.astype(np.float32)),
        "tip_ids": t(np.asarray(TIP_IDS, np.int32)), "blend": blend,
        "faces": torch.from_numpy(asset.faces.copy()),
    }
