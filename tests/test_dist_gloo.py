"""world_size-2 gloo test of the data-parallel path (shard -> local forward -> one all-gather)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hands_amd
from hands_amd.dist import data_parallel_forward, gather_predictions
from hands_amd.xdict import xdict


class _FakeModel:
    """Deterministic per-sample stand-in with the 22-key contract (the HIP path needs a GPU)."""

    def __call__(self, inputs, meta_info):
        s = inputs["img"].flatten(1).sum(1)
        out = xdict()
        for h in "rl":
            out[f"mano.vertices.{h}"] = s[:, None, None] * torch.ones(1, 778, 3)
            out[f"mano.joints3d.{h}"] = s[:, None, None] + torch.arange(63.0).view(1, 21, 3)
            out[f"mano.pose.{h}"] = s[:, None, None, None] * torch.eye(3).expand(1, 16, 3, 3)
            out[f"grasp.{h}"] = inputs[f"{h}_center_angle"] @ torch.ones(2, 9)
        return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    # even shards (6), uneven shards (5: 3 + 2 rows -> padded to 3 for the collective, trimmed after),
    # and an empty shard (1 sample on 2 ranks: rank 1 contributes zero rows)
    for bz in (6, 5, 1):
        inputs, meta = hands_amd.synthetic_inputs(bz, 3, img_res=8)
        meta["imgname"] = [f"{i}.jpg" for i in range(bz)]
        full = _FakeModel()(inputs, meta)
        got = data_parallel_forward(_FakeModel(), inputs, meta)
        ok = ok and all(torch.equal(got[k], full[k]) for k in full) and list(got.keys()) == list(full.keys())
        ok = ok and all(got[k].shape[0] == bz for k in got)
    single = gather_predictions({"x": torch.full((2, 3), float(rank))})
    ok = ok and torch.equal(single["x"], torch.tensor([[0.0] * 3] * 2 + [[1.0] * 3] * 2))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_allgather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: True, 1: True}
