"""Rank process of tests/test_gpu_dist.py (NOT a test module): one rank of a world-size-N data-parallel
forward of the REAL HandsLight on the HIP device, every rank on cuda:0, gloo collectives on host copies
(RCCL refuses two ranks on one GPU; the 1-GPU box cannot do better).  Rank 0 saves the gathered dict."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")

import torch
import torch.distributed as dist


def main():
    out_path, bz, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    on_host = len(sys.argv) <= 4 or sys.argv[4] != "device"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hands_amd
    from hands_amd.dist import data_parallel_forward
    dev = torch.device("cuda", 0)
    model = hands_amd.apply_recipe(hands_amd.HandsLight()).to(dev).eval()
    inputs, meta = hands_amd.synthetic_inputs(bz, seed, device=dev)        # every rank holds the GLOBAL batch
    meta["is_flipped"] = (torch.arange(bz, device=dev) % 3 == 1).long()
    # "device": the gather takes the stream-ordered path RCCL takes (pending stream_xdict -> pack and collective on
    # a side stream, gloo staging the device tensors itself); default: host copies
    import hands_amd.dist as dist_mod
    from hands_amd.xdict import stream_xdict
    got = data_parallel_forward(model, inputs, meta, gather_on_host=on_host)
    if not on_host:
        # the forward's pending stream_xdict must reach gather_predictions UNJOINED, so that pack +
        # collective run on the gather side stream behind the asynchronous tail; a rank with an empty shard is the
        # one exception (it slices its stand-in row away)
        from hands_amd.dist import shard_range
        lo, hi = shard_range(bz, rank, world)
        if hi > lo:
            assert dev in dist_mod._gather_streams, "the stream-ordered gather path was not taken"
            assert isinstance(got, stream_xdict) and got.is_pending
        # the gathered, still-pending dict is first READ ON A THIRD STREAM while the next forward
        # (+ its gather) is already enqueued -- the buffers `full` / `flat` of dist.gather_predictions were allocated on
        # the gather side stream and reach the consumer only through stream_xdict's wait_event + record_stream.
        # (Every rank issues the same collectives.)
        flipped = {k: (v.flip(0) if torch.is_tensor(v) and v.ndim else v) for k, v in inputs.items()}
        nxt = data_parallel_forward(model, flipped, meta, gather_on_host=False)
        third = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(third):
            snap = {k: v.clone() for k, v in got.items()}                    # joins on `third`
            junk = [torch.empty_like(v).normal_() for v in snap.values()]    # allocator churn on the consumer stream
        del got, junk, nxt
        for _ in range(3):                                                   # later forwards re-use the freed blocks
            data_parallel_forward(model, inputs, meta, gather_on_host=False)
        torch.cuda.synchronize(dev)
        got = snap
    if rank == 0:
        torch.save({k: v.detach().cpu().clone() for k, v in got.items()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
