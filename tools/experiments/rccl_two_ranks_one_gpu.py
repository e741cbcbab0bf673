"""Does RCCL accept two ranks on ONE device?  (The GPU boxes of this pool have one MI355X: the all-gather of the predictions has only
ever run as gloo or as single-rank RCCL.)  usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
--master-port 29533 tools/experiments/rccl_two_ranks_one_gpu.py"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", device_id=dev)
    x = torch.full((4, 8), float(rank + 1), device=dev)
    out = torch.empty(8, 8, device=dev)
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_gather_into_tensor over RCCL with {dist.get_world_size()} ranks on one device: rows {out[:, 0].tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as e:                      # noqa: BLE001 -- the message is the result
    print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:600]}", flush=True)
