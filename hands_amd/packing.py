"""One-time host-side packing of reference-layout parameters into the layouts the HIP kernels read.

Thin ctypes wrapper over the HOST entry points of libhands_hip.so (``hands_pack_*``, csrc/pack.cpp,
declared in include/hands_hip.h): a non-Python host packs a reference checkpoint with the same calls.
Not on the hot path: runs once per ``load_state_dict`` / device move.

* conv + eval BatchNorm2d -> folded weight/bias (reference applies them separately:
  src/nets/backbone/resnet.py:137-149), fp64 fold, one rounding to fp32;
* weights to ``[Cout_pad][Kpad]`` with k ordered (kh, kw, cin) -- see include/hands_hip.h;
* column permutations that keep every operand segment 16-byte aligned (HMR state row, grasp row,
  NCHW ``nn.Flatten`` order of feature_conv's Linear).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import PackedDims, check
from .mano import ManoAsset, TIP_IDS

BN_EPS = 1e-5


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
    wino: torch.Tensor = None  # 3x3 / stride 1 / pad 1 layers: Winograd F(2x2,3x3) weights (hands_pack_conv3x3_winograd_f64)
    sum_block: int = -1   # blocked fp32 summation of THIS layer: -1 = the engine's chain_limit, 0 = a single chain, 64 / 128 = blocks
    acc64: bool = False   # the owning model wants this layer accumulated in fp64 (HANDS_ACC_F64; ConvEngine.conv honours it)
    wino4: torch.Tensor = None  # the same layers: Winograd F(4x4,3x3) weights (hands_pack_conv3x3_winograd4_f64), when asked for


def _f64(t):
    """contiguous float64 numpy view of a CPU tensor / array (None stays None)."""
    if t is None:
        return None
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(t):
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fold_bn(w, bn_w, bn_b, bn_mean, bn_var, eps=BN_EPS):
    """(Cout,Cin,KH,KW) conv weight + BN stats -> folded weight, bias (fp64 tensors; hands_fold_bn_f32)."""
    wn = _f32(w)
    Cout = wn.shape[0]
    per = int(wn.size // Cout)
    wf = np.empty(wn.shape, np.float64)
    bf = np.empty(Cout, np.float64)
    g, b, m, v = (_f32(t) for t in (bn_w, bn_b, bn_mean, bn_var))
    check(_lib.lib().hands_fold_bn_f32(Cout, per, _p(wn), _p(g), _p(b), _p(m), _p(v), float(eps), _p(wf), _p(bf)),
          "hands_fold_bn_f32")
    return torch.from_numpy(wf), torch.from_numpy(bf)


def pack_conv(w, bias, stride, pad, device, cin_pad_to=None, winograd=True, winograd4=False) -> PackedConv:
    """w: (Cout, Cin, KH, KW) (any float dtype, CPU); bias: (Cout,) or None.  hands_pack_conv_f64.
    3x3 / stride 1 / pad 1 layers additionally get the Winograd form of the same folded weight (``pc.wino``: F(2x2,3x3);
    ``winograd4=True``: also ``pc.wino4``, the F(4x4,3x3) form, 36 Cout Cin floats)."""
    L = _lib.lib()
    Cout, Cin, KH, KW = w.shape
    d = PackedDims()
    check(L.hands_pack_conv_dims(Cout, Cin, KH, KW, int(cin_pad_to or 0), C.byref(d)), "hands_pack_conv_dims")
    wn, bn = _f64(w), _f64(bias)
    wp = np.empty((d.Cout_pad, d.Kpad), np.float32)
    bp = np.empty(d.Cout_pad, np.float32)
    check(L.hands_pack_conv_f64(Cout, Cin, KH, KW, int(cin_pad_to or 0), _p(wn), _p(bn), _p(wp), _p(bp)),
          "hands_pack_conv_f64")
    pc = PackedConv(torch.from_numpy(wp).to(device), torch.from_numpy(bp).to(device), d.Cin, d.Cout, KH, KW,
                    stride, pad, d.Kpad, macs_per_pixel=Cout * Cin * KH * KW)
    if winograd and KH == 3 and KW == 3 and stride == 1 and pad == 1 and not cin_pad_to:
        n = L.hands_pack_conv3x3_winograd_floats(Cout, Cin)
        if n > 0:
            up = np.empty(n, np.float32)
            check(L.hands_pack_conv3x3_winograd_f64(Cout, Cin, _p(wn), _p(up)), "hands_pack_conv3x3_winograd_f64")
            pc.wino = torch.from_numpy(up).to(device)
        n4 = L.hands_pack_conv3x3_winograd4_floats(Cout, Cin) if winograd4 else 0
        if n4 > 0:
            up = np.empty(n4, np.float32)
            check(L.hands_pack_conv3x3_winograd4_f64(Cout, Cin, _p(wn), _p(up)), "hands_pack_conv3x3_winograd4_f64")
            pc.wino4 = torch.from_numpy(up).to(device)
    return pc


def pack_conv1x1_dual(w0, b0, w1, b1, device) -> PackedConv:
    """conv3 + downsample of a stage's first bottleneck as one two-source GEMM: inputs are the fp64 folds
    (Cout,K0,1,1), (Cout,K1,1,1) and their biases.  hands_pack_conv1x1_dual_f64."""
    Cout, K0, K1 = w0.shape[0], w0.shape[1], w1.shape[1]
    Kpad, Cout_pad = (K0 + K1 + 15) // 16 * 16, (Cout + 127) // 128 * 128
    wp = np.empty((Cout_pad, Kpad), np.float32)
    bp = np.empty(Cout_pad, np.float32)
    a0, c0, a1, c1 = _f64(w0).reshape(Cout, K0), _f64(b0), _f64(w1).reshape(Cout, K1), _f64(b1)
    check(_lib.lib().hands_pack_conv1x1_dual_f64(Cout, K0, K1, _p(a0), _p(c0), _p(a1), _p(c1), _p(wp), _p(bp)),
          "hands_pack_conv1x1_dual_f64")
    return PackedConv(torch.from_numpy(wp).to(device), torch.from_numpy(bp).to(device), K0 + K1, (Cout + 3) // 4 * 4, 1, 1,
                      1, 0, Kpad, macs_per_pixel=Cout * (K0 + K1))


def pack_linear(w, bias, device, col_index=None, k_total=None, row_index=None, n_total=None) -> PackedConv:
    """nn.Linear weight (N, K) -> 1x1 'conv'.  ``col_index[k_ref] = k_packed`` places reference input
    column k_ref at packed column k_packed (row of width ``k_total``); ``row_index`` likewise for
    output rows (``n_total`` stored outputs).  hands_pack_linear_f64."""
    N, K = w.shape
    kt, nt = int(k_total or K), int(n_total or N)
    Kpad, Cout_pad = (kt + 15) // 16 * 16, (nt + 127) // 128 * 128
    wp = np.empty((Cout_pad, Kpad), np.float32)
    bp = np.empty(Cout_pad, np.float32)
    ci = None if col_index is None else np.ascontiguousarray(np.asarray(col_index), dtype=np.int32)
    ri = None if row_index is None else np.ascontiguousarray(np.asarray(row_index), dtype=np.int32)
    wn, bn = _f64(w), _f64(bias)
    check(_lib.lib().hands_pack_linear_f64(N, K, _p(wn), _p(bn), _p(ci), kt, _p(ri), nt, _p(wp), _p(bp)),
          "hands_pack_linear_f64")
    return PackedConv(torch.from_numpy(wp).to(device), torch.from_numpy(bp).to(device), Kpad, (nt + 3) // 4 * 4, 1, 1, 1, 0,
                      Kpad, macs_per_pixel=N * K)


# HMR state row layout (see hands_hmr_init_f32): feat | pose6d 96 | shape 10 | 2 pad | cam 3 | 1 pad
def hmr_state_columns(F):
    """packed column of each reference concat column [feat F, pose_6d 96, shape 10, cam 3]
    (src/nets/hmr_layer.py:80 with dict order from src/nets/hand_heads/hand_hmr.py:66-69)."""
    cols = list(range(F)) + [F + i for i in range(96)] + [F + 96 + i for i in range(10)] + \
           [F + 108 + i for i in range(3)]
    return cols


HMR_VEC = 112  # width of the vector part of the state row


def pack_mano(asset: ManoAsset, device):
    """MANO constants for the pose / blend-GEMM / skin kernels.  hands_pack_mano_f32."""
    vt, sd, pd, Jr, hm = (_f32(a) for a in (asset.v_template, asset.shapedirs, asset.posedirs, asset.J_regressor,
                                            asset.hands_mean))
    pose_mean, J_template, J_shapedirs = np.empty(48, np.float32), np.empty((16, 3), np.float32), np.empty((48, 10), np.float32)
    bw, bb = np.empty((2432, 160), np.float32), np.empty(2432, np.float32)
    check(_lib.lib().hands_pack_mano_f32(_p(vt), _p(sd), _p(pd), _p(Jr), _p(hm), _p(pose_mean), _p(J_template),
                                         _p(J_shapedirs), _p(bw), _p(bb)), "hands_pack_mano_f32")
    blend = PackedConv(torch.from_numpy(bw).to(device), torch.from_numpy(bb).to(device), 160, 2336, 1, 1, 1, 0, 160,
                       macs_per_pixel=2334 * 145)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return {
        "pose_mean": t(pose_mean), "J_template": t(J_template), "J_shapedirs": t(J_shapedirs),
        "lbs_weights": t(asset.lbs_weights.astype(np.float32)),
        "tip_ids": t(np.asarray(TIP_IDS, np.int32)), "blend": blend,
        "faces": torch.from_numpy(asset.faces.copy()),
    }
