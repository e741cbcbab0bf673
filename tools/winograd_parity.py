"""Experiment (CPU, test infrastructure): what does Winograd F(2x2,3x3) in fp32 on every stride-1 padded 3x3
convolution do to the end-to-end parity of hands_light?  Patches the ORACLE's conv2d and compares vertices."""
import sys, os, json
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hands_oracle as O
from hands_amd.mano import synthetic_mano_asset
from hands_amd.weights import synthetic_inputs
import hands_amd

torch.set_num_threads(8)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
_orig = F.conv2d
COUNT = [0]

def wino(x, w):
    dt = x.dtype
    B, C, H, W = x.shape
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # B,C,nh,nw,4,4
    bt = Bt.to(dt)
    V = torch.einsum('xa,bchwae,ye->bchwxy', bt, d, bt)          # fp32 adds
    U = torch.einsum('xa,ocae,ye->ocxy', G, w.double(), G).to(dt)  # offline in fp64, rounded once
    M = torch.einsum('bchwxy,ocxy->bohwxy', V, U)
    at = At.to(dt)
    Y = torch.einsum('ix,bohwxy,jy->bohwij', at, M, at)          # B,O,nh,nw,2,2
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], Hp, Wp)
    return Y[:, :, :H, :W].contiguous()

def patched(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if w.shape[2:] == (3, 3) and stride in (1, (1, 1)) and padding in (1, (1, 1)) and bias is None:
        COUNT[0] += 1
        return wino(x, w)
    return _orig(x, w, bias, stride, padding, dilation, groups)

def run(sd, bz, seed, dtype, use_wino):
    inputs, meta = synthetic_inputs(bz, seed)
    ar, al = synthetic_mano_asset(True), synthetic_mano_asset(False)
    cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t
    sd2 = {k: cast(v) for k, v in sd.items()}
    inputs = {k: cast(v) for k, v in inputs.items()}
    meta = {k: cast(v) for k, v in meta.items()}
    conv = patched if use_wino else _orig
    O.F.conv2d = conv
    try:
        out = O.hands_light_forward(sd2, ar, al, inputs, meta)
    finally:
        O.F.conv2d = _orig
    return {k: v.double() for k, v in out.items() if torch.is_tensor(v)}

if __name__ == '__main__':
    os.environ.setdefault('HANDS_SYNTHETIC_MANO','1')
    m = hands_amd.HandsLight(); hands_amd.apply_recipe(m); m.eval()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for bz, seed in [(2, 0), (2, 1), (2, 2)]:
        ref64 = run(sd, bz, seed, torch.float64, False)
        d32 = run(sd, bz, seed, torch.float32, False)
        COUNT[0] = 0
        w32 = run(sd, bz, seed, torch.float32, True)
        n = COUNT[0]
        for k in ('mano.vertices.r', 'mano.vertices.l', 'mano.joints3d.r', 'mano.joints3d.l'):
            if k not in ref64: continue
            e_d = (d32[k] - ref64[k]).abs().max().item()
            e_w = (w32[k] - ref64[k]).abs().max().item()
            e_dw = (w32[k] - d32[k]).abs().max().item()
            print(f'bz{bz} seed{seed} {k}: direct32-vs-64 {e_d:.3e}  wino32-vs-64 {e_w:.3e}  wino-vs-direct {e_dw:.3e}  ({n} convs)')
