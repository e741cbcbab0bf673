#!/usr/bin/env python3
"""HandOccNet per-stage fp32 error budget on the CPU (round 6 dev experiment).

Every GEMM-shaped op (conv / linear / matmul) of the oracle is "ideal" (fp64 on fp32 operands, one rounding) EXCEPT in one
stage, which runs the reference's fp32 arithmetic (ATen).  The max vertex error against the fp64 network then prices what that
stage's fp32 accumulation contributes on top of the storage-rounding floor ("none" = every stage ideal, "all" = reference).
usage: python tools/experiments/hon_stage_budget_cpu.py [n_seeds]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.set_num_threads(8)
m = hands_amd.apply_recipe(hands_amd.HandOccNet())
sd = {k: v.clone() for k, v in m.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
c64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
_conv, _lin, _mm = HO._conv, HO._lin, torch.matmul
FP32_STAGE = {"s": None, "cur": None}      # s: the stage left in fp32 ("all": everything, None: nothing)


def stage_of(p):
    if p.startswith("backbone.layer"):
        return "resnet"
    if p.startswith("backbone."):
        return "fpn"
    if p.startswith("FIT"):
        return "fit"
    if p.startswith("SET"):
        return "set"
    if ".hand_regHead.hg" in p:
        return "hourglass"
    if ".hand_regHead" in p:
        return "reghead"
    if ".hand_Encoder" in p:
        return "encoder"
    if ".mano_regHead" in p or p.startswith("kpe") or p.startswith("grasp"):
        return "mlp"
    raise KeyError(p)


def ideal(p):
    s = FP32_STAGE["s"]
    FP32_STAGE["cur"] = stage_of(p)
    return s != "all" and s != stage_of(p)


def conv_x(x, sd_, p, stride=1, padding=0):
    if x.dtype == torch.float64 or not ideal(p):
        return _conv(x, sd_, p, stride, padding)
    b = sd_.get(p + ".bias")
    return F.conv2d(x.double(), sd_[p + ".weight"].double(), None if b is None else b.double(), stride=stride,
                    padding=padding).float()


def lin_x(x, sd_, p):
    if x.dtype == torch.float64 or not ideal(p):
        return _lin(x, sd_, p)
    return F.linear(x.double(), sd_[p + ".weight"].double(), sd_[p + ".bias"].double()).float()


class _T:
    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def matmul(a, b):      # attention matmuls: stage of the last conv seen (FIT / SET)
        s = FP32_STAGE["s"]
        if a.dtype == torch.float32 and s != "all" and s != FP32_STAGE["cur"]:
            return _mm(a.double(), b.double()).float()
        return _mm(a, b)


HO._conv, HO._lin, HO.torch = conv_x, lin_x, _T()
verts = lambda o: torch.cat([o["mano.vertices.r"], o["mano.vertices.l"]], 0).double()
stages = [None, "resnet", "fpn", "fit", "set", "hourglass", "reghead", "encoder", "mlp", "all"]
res = {s: [] for s in stages}
for seed in range(n):
    ci, cm = synthetic_inputs(2, seed)
    v64 = verts(HO.handoccnet_forward(sd64, ar, al, c64(ci), c64(cm)))
    for s in stages:
        FP32_STAGE["s"] = s
        v = verts(HO.handoccnet_forward(sd, ar, al, ci, cm))
        res[s].append((v - v64).abs().max().item())
    print(seed, " ".join(f"{s}:{res[s][-1]:.2e}" for s in stages), flush=True)
print("median / max over seeds")
for s in stages:
    r = np.array(res[s])
    print(f"  fp32 in {str(s):10s} median {np.median(r):.2e} max {r.max():.2e}  excess variance vs floor "
          f"{np.median(r)**2 - np.median(np.array(res[None]))**2:.2e}")
