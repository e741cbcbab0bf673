"""Data-parallel inference helpers: one process per GPU, batch sharded on dim 0, one all-gather of
the predictions per forward (RCCL over xGMI on the GPU box, gloo in the CPU tests).

The reference has no inference-side collective (Lightning DDP is training
only); samples are independent (eval BatchNorm, no cross-sample op), so ranks exchange nothing
until the end, where the 22 prediction tensors are packed into ONE (rows, D) fp32 buffer and
gathered with a single collective -- 20.5 KB per hand, so a direct all-gather keeps all 7 xGMI
links of a GPU busy instead of 22 small ring steps.

Uneven shards (``bz % world != 0``): every rank pads its packed buffer to ``ceil(bz / world)`` rows,
so the collective always moves equal counts (RCCL hangs or corrupts rows on mismatched counts), and
the padding rows are trimmed after the gather with the same ``shard_range`` arithmetic.  A rank whose
shard is empty (``bz < world``) runs the forward on sample 0 and contributes zero valid rows.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .xdict import stream_xdict, xdict

_gather_streams = {}   # device -> side stream the asynchronous gather runs on
_single_buffer_gather = {}   # (backend, device type) -> False once all_gather_into_tensor proved unavailable


def shard_range(bz: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank ``rank``; remainders go to the first ranks."""
    q, r = divmod(bz, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def max_shard(bz: int, world: int) -> int:
    return -(-bz // world)


def shard_batch(inputs: dict, meta_info: dict, rank: int, world: int):
    """This rank's slice of every per-sample entry.  An empty shard is replaced by sample 0 (its rows
    are discarded by :func:`gather_predictions`), so the model never sees a zero-sized batch."""
    bz = inputs["img"].shape[0] if "img" in inputs else next(
        v.shape[0] for v in inputs.values() if isinstance(v, torch.Tensor) and v.ndim >= 1)
    lo, hi = shard_range(bz, rank, world)
    if hi == lo:
        lo, hi = 0, 1

    def cut(v):
        if isinstance(v, torch.Tensor) and v.ndim >= 1 and v.shape[0] == bz:
            return v[lo:hi].contiguous()
        if isinstance(v, (list, tuple)) and len(v) == bz:
            return type(v)(v[lo:hi])
        return v

    return {k: cut(v) for k, v in inputs.items()}, {k: cut(v) for k, v in meta_info.items()}


def pack_predictions(out: dict, rows: int | None = None):
    """(rows, D) fp32 buffer + the layout needed to unpack it; ``rows`` > local batch pads with zeros."""
    keys = list(out.keys())
    bz = out[keys[0]].shape[0]
    layout = [(k, tuple(out[k].shape[1:])) for k in keys]
    width = lambda shp: int(torch.Size(shp).numel())       # explicit: reshape(0, -1) is ambiguous
    flat = torch.cat([out[k].reshape(bz, width(shp)).to(torch.float32) for k, shp in layout], dim=1)
    if rows is not None and rows != bz:
        assert rows > bz
        padded = flat.new_zeros((rows, flat.shape[1]))
        padded[:bz] = flat
        flat = padded
    return flat.contiguous(), layout


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


def gather_predictions(out: dict, group=None, global_bz: int | None = None) -> xdict:
    """All-gather every rank's prediction dict into the global one (rank order = sample order).

    ``global_bz=None``: every rank holds the same number of rows (weak-scaling benchmark).  Otherwise
    the ranks hold the ``shard_range(global_bz, rank, world)`` shards (possibly uneven or empty): buffers
    are padded to the largest shard before the collective and trimmed after it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return out if isinstance(out, xdict) else xdict(out)
    if isinstance(out, stream_xdict) and out.is_pending:
        # the forward's tail is still running on its own stream: pack + all-gather on a side stream ordered after
        # it and hand back another stream-ordered dict, so the caller's stream (the next forward's trunks) never
        # waits for this forward's tail or for the collective
        dev = out.__dict__["_device"]
        st = _gather_streams.get(dev)
        if st is None:
            st = _gather_streams[dev] = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            res = gather_predictions(xdict(out), group, global_bz)     # joins on `st`, collective ordered after it
            ready = torch.cuda.Event()
            ready.record(st)
        return stream_xdict(res, ready, dev)
    world = dist.get_world_size(group)
    rows = None if global_bz is None else max_shard(global_bz, world)
    flat, layout = pack_predictions(out, rows)
    full = torch.empty((world * flat.shape[0], flat.shape[1]), dtype=flat.dtype, device=flat.device)
    if _single_buffer_gather.get((dist.get_backend(group), flat.device.type), True):
        try:
            dist.all_gather_into_tensor(full, flat, group=group)
        except (NotImplementedError, RuntimeError) as e:
            # a backend / version without the single-buffer form (older gloo: "no support for _allgather_base");
            # every rank takes this branch together (same backend), the choice is cached per backend and device type.
            # Anything else (a real collective failure) is re-raised.
            # Only the backend's own "this collective does not exist" messages count -- a real failure on ONE rank that
            # happened to say "not supported" must not make that rank issue a different collective than its peers (hang).
            msg = str(e).lower()
            if not isinstance(e, NotImplementedError) and not any(
                    t in msg for t in ("allgather_base", "all_gather_into_tensor", "allgather_into_tensor")):
                raise
            _single_buffer_gather[(dist.get_backend(group), flat.device.type)] = False
    if not _single_buffer_gather.get((dist.get_backend(group), flat.device.type), True):
        parts = [torch.empty_like(flat) for _ in range(world)]      # same bytes, list form
        dist.all_gather(parts, flat, group=group)
        full = torch.cat(parts, 0)
    if global_bz is not None:
        keep = []
        for r in range(world):
            lo, hi = shard_range(global_bz, r, world)
            keep.append(full[r * rows: r * rows + (hi - lo)])
        full = torch.cat(keep, 0) if len(keep) > 1 else keep[0]
        assert full.shape[0] == global_bz
    return unpack_predictions(full, layout)


def data_parallel_forward(model, inputs, meta_info, group=None, gather_on_host=False) -> xdict:
    """Shard the global batch over the ranks, run the local forward, gather the predictions.
    ``gather_on_host``: move the local predictions to the CPU first (gloo dry runs of the GPU path)."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    bz = inputs["img"].shape[0] if "img" in inputs else next(
        v.shape[0] for v in inputs.values() if isinstance(v, torch.Tensor) and v.ndim >= 1)
    li, lm = shard_batch(inputs, meta_info, rank, world)
    out = model(li, lm)
    lo, hi = shard_range(bz, rank, world)
    if world > 1 and hi == lo:      # empty shard (decided WITHOUT touching `out`): drop the stand-in row
        out = {k: v[:0] for k, v in out.items()}
    # a non-empty shard is passed on untouched: a pending stream_xdict keeps the asynchronous tail un-joined and
    # gather_predictions packs + gathers on its side stream behind it
    if gather_on_host:
        out = {k: v.cpu() for k, v in out.items()}
    return gather_predictions(out, group, global_bz=bz if world > 1 else None)


PACKED_WIDTH = 5171      # fp32 columns of one packed sample pair (the 22 prediction tensors of a forward)


def allgather_selfcheck(device, rows: int = 32, width: int = PACKED_WIDTH, timeout_s: float = 180.0, group=None,
                        _corrupt: bool = False) -> dict:
    """First-contact check of the prediction all-gather on the layout the forward uses (one ``(rows, width)`` fp32 buffer per
    rank, ``all_gather_into_tensor`` -- RCCL on the GPU box, gloo in the CPU test): every rank fills its buffer with a pattern
    that encodes (rank, row, column) in exactly representable integers, gathers, and verifies EVERY rank's segment.

    A collective that never returns (a peer died, a link is down) cannot be cancelled from Python, so a watchdog thread ends
    this process with exit code 3 after ``timeout_s``; a mismatch raises ``RuntimeError`` (bench.py exits with code 4).  No
    process is re-executed.  The window is generous (default 180 s) because this is the job's FIRST collective: RCCL builds its
    communicator and transports inside it (seconds on a cold 8-GPU node); a small warm-up all-gather runs first, so ``us`` --
    the wall time of the checked collective in the returned ``{"ranks", "us", "first_contact_s", "rows", "width"}`` -- is a warm
    figure and ``first_contact_s`` shows what the set-up cost."""
    import os
    import sys
    import threading
    import time
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {"ranks": 1, "us": 0.0, "first_contact_s": 0.0, "rows": rows, "width": width}
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device(device)

    def pattern(r):
        i = torch.arange(rows, dtype=torch.int64, device=dev).view(-1, 1)
        j = torch.arange(width, dtype=torch.int64, device=dev).view(1, -1)
        return ((r * 7919 + i * 131 + j * 17) % 65521).to(torch.float32)       # < 2^24: exact in fp32

    done = threading.Event()

    def watchdog():
        if not done.wait(timeout_s):
            sys.stderr.write(f"hands_amd.dist.allgather_selfcheck: rank {rank} of {world}: the all-gather did not complete "
                             f"within {timeout_s:.0f} s (a peer is gone or the fabric is down) -- exiting with code 3\n")
            sys.stderr.flush()
            os._exit(3)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        mine = pattern(rank)
        if _corrupt:
            mine[rows // 2, width // 2] += 1.0
        full = torch.empty((world * rows, width), dtype=torch.float32, device=dev)
        tc = time.perf_counter()
        warm = [torch.empty(8, dtype=torch.float32, device=dev) for _ in range(world)]
        dist.all_gather(warm, torch.full((8,), float(rank), dtype=torch.float32, device=dev), group=group)   # first contact
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        first_contact_s = time.perf_counter() - tc
        t0 = time.perf_counter()
        single = _single_buffer_gather.get((dist.get_backend(group), dev.type), True)
        if single:
            try:
                dist.all_gather_into_tensor(full, mine, group=group)
            except (NotImplementedError, RuntimeError) as e:
                msg = str(e).lower()
                if not isinstance(e, NotImplementedError) and not any(
                        t in msg for t in ("allgather_base", "all_gather_into_tensor", "allgather_into_tensor")):
                    raise
                single = _single_buffer_gather[(dist.get_backend(group), dev.type)] = False
        if not single:
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            full = torch.cat(parts, 0)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        us = (time.perf_counter() - t0) * 1e6
        bad = [r for r in range(world) if not torch.equal(full[r * rows:(r + 1) * rows], pattern(r))]
    finally:
        done.set()
    if bad:
        raise RuntimeError(f"hands_amd.dist.allgather_selfcheck: rank {rank} of {world}: the segments of rank(s) {bad} arrived "
                           f"corrupted ({rows} x {width} fp32 per rank, backend {dist.get_backend(group)})")
    return {"ranks": world, "us": round(us, 1), "first_contact_s": round(first_contact_s, 3), "rows": rows, "width": width}
