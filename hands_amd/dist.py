"""Data-parallel inference helpers: one process per GPU, batch sharded on dim 0, one all-gather of
the predictions per forward (RCCL over xGMI on the GPU box, gloo in the CPU tests).

The reference has no inference-side collective (Lightning DDP is training
only); samples are independent (eval BatchNorm, no cross-sample op), so ranks exchange nothing
until the end, where the 22 prediction tensors are packed into ONE (bz_local, D) fp32 buffer and
gathered with a single collective -- 20.5 KB per hand, so a direct all-gather keeps all 7 xGMI
links of a GPU busy instead of 22 small ring steps.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .xdict import xdict


def shard_range(bz: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank ``rank``; remainders go to the first ranks."""
    q, r = divmod(bz, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(inputs: dict, meta_info: dict, rank: int, world: int):
    bz = inputs["img"].shape[0]
    lo, hi = shard_range(bz, rank, world)

    def cut(v):
        if isinstance(v, torch.Tensor) and v.ndim >= 1 and v.shape[0] == bz:
            return v[lo:hi].contiguous()
        if isinstance(v, (list, tuple)) and len(v) == bz:
            return type(v)(v[lo:hi])
        return v

    return {k: cut(v) for k, v in inputs.items()}, {k: cut(v) for k, v in meta_info.items()}


def pack_predictions(out: dict):
    """(bz, D) fp32 buffer + the layout needed to unpack it."""
    keys = list(out.keys())
    bz = out[keys[0]].shape[0]
    layout = [(k, tuple(out[k].shape[1:])) for k in keys]
    flat = torch.cat([out[k].reshape(bz, -1).to(torch.float32) for k in keys], dim=1).contiguous()
    return flat, layout


def unpack_predictions(flat: torch.Tensor, layout) -> xdict:
    res = xdict()
    n = flat.shape[0]
    col = 0
    for k, shp in layout:
        w = 1
        for s in shp:
            w *= s
        res[k] = flat[:, col:col + w].reshape((n,) + shp)
        col += w
    return res


def gather_predictions(out: dict, group=None) -> xdict:
    """All-gather every rank's prediction dict (equal local batch sizes) into the global one."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return xdict(out)
    world = dist.get_world_size(group)
    flat, layout = pack_predictions(out)
    full = torch.empty((world * flat.shape[0], flat.shape[1]), dtype=flat.dtype, device=flat.device)
    try:
        dist.all_gather_into_tensor(full, flat, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat, group=group)
        full = torch.cat(parts, 0)
    return unpack_predictions(full, layout)


def data_parallel_forward(model, inputs, meta_info, group=None) -> xdict:
    """Shard the global batch over the ranks, run the local forward, gather the predictions."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    li, lm = shard_batch(inputs, meta_info, rank, world)
    return gather_predictions(model(li, lm), group)
