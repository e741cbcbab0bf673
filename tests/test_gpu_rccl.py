"""RCCL self-test on the 1-GPU box (SURVEY.md section 7 'gpurun gives 1 GPU'): a single-rank NCCL(=RCCL)
process group runs the very collective the N > 1 path uses on the packed prediction buffer."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_all_gather_of_packed_predictions(recipe_model):
    import copy
    from hands_amd.dist import pack_predictions, unpack_predictions
    from hands_amd.weights import synthetic_inputs
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        model = copy.deepcopy(recipe_model).to(dev)
        inputs, meta = synthetic_inputs(2, 0, device=dev)
        out = model(inputs, meta)
        flat, layout = pack_predictions(out)
        full = torch.empty_like(flat)
        dist.all_gather_into_tensor(full, flat)          # RCCL collective, world size 1
        dist.barrier()
        torch.cuda.synchronize()
        back = unpack_predictions(full, layout)
        assert list(back.keys()) == list(out.keys())
        for k in out:
            assert torch.equal(back[k], out[k]), k
    finally:
        if created:
            dist.destroy_process_group()
