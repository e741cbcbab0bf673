#!/usr/bin/env python3
"""HandOccNet: what an evaluator with EXACT accumulation and fp32 storage would sit at (CPU-only dev experiment, round 6).

Every conv / linear / matmul of the oracle is evaluated in fp64 on its fp32 operands and rounded to fp32 once ("ideal"): the
only error left against the fp64 network is the storage rounding of each layer's output, the floor any fp32-activation path
has.  Prints per seed the max vertex error of ref32 (the reference's arithmetic) and of ideal against fp64, and ideal vs ref32.
usage: python tools/experiments/hon_floor_cpu.py [n_seeds] [threads]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch
import torch.nn.functional as F

import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import handoccnet_oracle as HO

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
m = hands_amd.apply_recipe(hands_amd.HandOccNet())
sd = {k: v.clone() for k, v in m.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
c64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}

_conv, _lin, _mm = HO._conv, HO._lin, torch.matmul
MODE = {"ideal": False}


def conv_ideal(x, sd_, p, stride=1, padding=0):
    if not MODE["ideal"]:
        return _conv(x, sd_, p, stride, padding)
    b = sd_.get(p + ".bias")
    return F.conv2d(x.double(), sd_[p + ".weight"].double(), None if b is None else b.double(), stride=stride,
                    padding=padding).float()


def lin_ideal(x, sd_, p):
    if not MODE["ideal"]:
        return _lin(x, sd_, p)
    return F.linear(x.double(), sd_[p + ".weight"].double(), sd_[p + ".bias"].double()).float()


class _T:
    """torch proxy for the oracle module: matmul in fp64 when ideal."""
    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def matmul(a, b):
        if MODE["ideal"] and a.dtype == torch.float32:
            return _mm(a.double(), b.double()).float()
        return _mm(a, b)


HO._conv, HO._lin, HO.torch = conv_ideal, lin_ideal, _T()
verts = lambda o: torch.cat([o["mano.vertices.r"], o["mano.vertices.l"]], 0).double()
rows = []
for seed in range(n):
    ci, cm = synthetic_inputs(2, seed)
    MODE["ideal"] = False
    v32 = verts(HO.handoccnet_forward(sd, ar, al, ci, cm))
    v64 = verts(HO.handoccnet_forward(sd64, ar, al, c64(ci), c64(cm)))
    MODE["ideal"] = True
    vid = verts(HO.handoccnet_forward(sd, ar, al, ci, cm))
    e = lambda a, b: (a - b).abs().max().item()
    rows.append((e(v32, v64), e(vid, v64), e(vid, v32)))
    print(f"seed {seed}: ref32-fp64 {rows[-1][0]:.2e}  ideal-fp64 {rows[-1][1]:.2e}  ideal-ref32 {rows[-1][2]:.2e}", flush=True)
r = np.array(rows)
print("median", np.median(r, 0), "max", r.max(0), "median ratio ideal/ref32", np.median(r[:, 1] / r[:, 0]))
