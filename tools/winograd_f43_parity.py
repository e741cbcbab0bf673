#!/usr/bin/env python3
"""Experiment (CPU, test infrastructure): end-to-end parity of hands_light with Winograd F(4x4,3x3) in fp32 on the stride-1 padded
3x3 convolutions (36 multiplications per 4x4 outputs against 64 for F(2x2,3x3) and 144 direct), next to F(2x2,3x3) and the direct
algorithm -- against an fp64 forward and against the fp32 oracle (= the reference's output, the parity bar).  Interpolation points
0, +-1, +-2, inf (Lavin & Gray); G g G^T in fp64 rounded once, transforms and products in fp32.
usage: python tools/winograd_f43_parity.py [seeds, default 4] [scope: all | wide (only the layers with >= 128 channels)]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
import torch.nn.functional as F
import hands_amd
from hands_amd.mano import synthetic_mano_asset
from hands_amd.weights import synthetic_inputs
from oracle import hands_oracle as O

torch.set_num_threads(8)
D = torch.float64
MATS = {
    2: (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=D),
        torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=D),
        torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=D)),
    4: (torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                      [0, 4, 0, -5, 0, 1]], dtype=D),
        torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], dtype=D),
        torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=D)),
}
_orig = F.conv2d
STATE = {"m": 0, "min_c": 0}


def wino(x, w, m):
    Bt, G, At = MATS[m]
    dt = x.dtype
    B, C, H, W = x.shape
    Hp, Wp = -(-H // m) * m, -(-W // m) * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = xp.unfold(2, m + 2, m).unfold(3, m + 2, m)                   # B, C, nh, nw, m+2, m+2
    bt, at = Bt.to(dt), At.to(dt)
    V = torch.einsum("xa,bchwae,ye->bchwxy", bt, d, bt)
    U = torch.einsum("xa,ocae,ye->ocxy", G, w.double(), G).to(dt)    # offline in fp64, rounded once
    M = torch.einsum("bchwxy,ocxy->bohwxy", V, U)
    Y = torch.einsum("ix,bohwxy,jy->bohwij", at, M, at)
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], Hp, Wp)
    return Y[:, :, :H, :W].contiguous()


def patched(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if STATE["m"] and w.shape[2:] == (3, 3) and stride in (1, (1, 1)) and padding in (1, (1, 1)) and bias is None:
        m = STATE["m"] if w.shape[1] >= STATE["min_c"] else 2          # narrow layers stay on F(2x2,3x3)
        return wino(x, w, m)
    return _orig(x, w, bias, stride, padding, dilation, groups)


def run(sd, seed, dtype, m, min_c=0):
    inputs, meta = synthetic_inputs(2, seed)
    cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t
    STATE["m"], STATE["min_c"] = m, min_c
    O.F.conv2d = patched
    try:
        out = O.hands_light_forward({k: cast(v) for k, v in sd.items()}, synthetic_mano_asset(True), synthetic_mano_asset(False),
                                    {k: cast(v) for k, v in inputs.items()}, {k: cast(v) for k, v in meta.items()})
    finally:
        O.F.conv2d = _orig
        STATE["m"] = 0
    return torch.cat([out["mano.vertices.r"], out["mano.vertices.l"]]).double()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    min_c = 128 if len(sys.argv) > 2 and sys.argv[2] == "wide" else 0
    model = hands_amd.apply_recipe(hands_amd.HandsLight()).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    worst = {}
    for seed in range(n):
        r64 = run(sd, seed, torch.float64, 0)
        o32 = run(sd, seed, torch.float32, 0)
        row = []
        for name, m in (("direct", 0), ("F(2,3)", 2), ("F(4,3)", 4)):
            v = o32 if m == 0 else run(sd, seed, torch.float32, m, min_c)
            e64, eo = (v - r64).abs().max().item(), (v - o32).abs().max().item()
            row.append(f"{name}: vs fp64 {e64:.2e}, vs fp32 oracle {eo:.2e}")
            worst[name] = max(worst.get(name, (0, 0)), (e64, eo), key=lambda t: t[1] if name != "direct" else t[0])
        print(f"seed {seed}: " + " | ".join(row), flush=True)
    print("worst (vs fp64, vs fp32 oracle):", {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in worst.items()})
