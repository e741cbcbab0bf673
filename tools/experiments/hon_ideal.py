"""Shared by the round-6 HandOccNet error experiments: the oracle with every GEMM-shaped op (conv / linear / matmul) evaluated in
fp64 on its fp32 operands and rounded once ("ideal": exact accumulation, fp32 storage), optionally leaving some stages in the
reference's fp32 arithmetic.  Dev tooling only."""
import torch
import torch.nn.functional as F

from hands_amd.handoccnet import stage_of
from oracle import handoccnet_oracle as HO

_conv, _lin, _mm = HO._conv, HO._lin, torch.matmul
STATE = {"on": False, "fp32": frozenset(), "cur": None}


def _ideal(p):
    STATE["cur"] = stage_of(p)
    return STATE["on"] and STATE["cur"] not in STATE["fp32"]


def _conv_x(x, sd, p, stride=1, padding=0):
    if x.dtype == torch.float64 or not _ideal(p):
        return _conv(x, sd, p, stride, padding)
    b = sd.get(p + ".bias")
    return F.conv2d(x.double(), sd[p + ".weight"].double(), None if b is None else b.double(), stride=stride,
                    padding=padding).float()


def _lin_x(x, sd, p):
    if x.dtype == torch.float64 or not _ideal(p):
        return _lin(x, sd, p)
    return F.linear(x.double(), sd[p + ".weight"].double(), sd[p + ".bias"].double()).float()


class _T:
    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def matmul(a, b):      # attention products: stage of the last conv seen (FIT / SET)
        if a.dtype == torch.float32 and STATE["on"] and STATE["cur"] not in STATE["fp32"]:
            return _mm(a.double(), b.double()).float()
        return _mm(a, b)


HO._conv, HO._lin, HO.torch = _conv_x, _lin_x, _T()


def forward(sd, ar, al, ci, cm, ideal=True, fp32_stages=(), **kw):
    STATE["on"], STATE["fp32"] = bool(ideal), frozenset(fp32_stages)
    try:
        return HO.handoccnet_forward(sd, ar, al, ci, cm, **kw)
    finally:
        STATE["on"] = False
