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


def _selfcheck_worker(rank, world, port, mode, q):
    """mode: "ok" | "corrupt" (rank 1 sends one wrong element) | "absent" (rank 1 is alive but never joins the collective)."""
    import time
    from hands_amd.dist import allgather_selfcheck
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "absent" and rank == 1:
        q.put((rank, "sleeping"))
        q.close()
        q.join_thread()
        time.sleep(12)                       # a fake rank that is gone as far as the collective is concerned
        os._exit(0)
    try:
        r = allgather_selfcheck("cpu", rows=4, timeout_s=3.0, _corrupt=(mode == "corrupt" and rank == 1))
        q.put((rank, r["ranks"]))
    except RuntimeError as e:
        q.put((rank, str(e)))
        q.close()
        q.join_thread()                      # (the feeder thread must flush before the hard exit)
        os._exit(4)                          # what bench.py does on a mismatch
    dist.barrier()
    dist.destroy_process_group()


def _run_selfcheck(mode):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_selfcheck_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
    msgs = {}
    while not q.empty():
        r, m = q.get()
        msgs[r] = m
    return [p.exitcode for p in procs], msgs


def test_allgather_selfcheck_passes_detects_corruption_and_times_out():
    """the first-contact check bench.py runs at N > 1 before the timed region (hands_amd.dist.
    allgather_selfcheck, gloo here, RCCL on the node): clean run -> every rank sees both segments; one corrupted element ->
    EVERY rank names the sender and ends with code 4; a peer that never joins -> the waiting rank's watchdog ends it with
    code 3 within the timeout instead of hanging the job."""
    codes, msgs = _run_selfcheck("ok")
    assert codes == [0, 0] and msgs == {0: 2, 1: 2}
    codes, msgs = _run_selfcheck("corrupt")
    assert codes == [4, 4], (codes, msgs)
    assert all("rank(s) [1]" in msgs[r] and "corrupted" in msgs[r] for r in (0, 1)), msgs
    codes, msgs = _run_selfcheck("absent")
    assert codes[0] == 3, (codes, msgs)      # rank 0 waited 3 s for the collective, then its watchdog ended it
