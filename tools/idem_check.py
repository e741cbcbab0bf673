import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch, hands_amd
torch.manual_seed(0)
model = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
model.overlap_trunks = os.environ.get("OVERLAP", "1") == "1"
model.async_tail = os.environ.get("ASYNC", "1") == "1"
model.engine.use_splitk = os.environ.get("SPLITK", "1") == "1"
for bz in (64,):
    inputs, meta = hands_amd.synthetic_inputs(bz, 0, device=torch.device("cuda"))
    with torch.no_grad():
        a = {k: v.clone() for k, v in model(inputs, meta).items()}
        s_in, s_meta = hands_amd.synthetic_inputs(2, 1, device=torch.device("cuda"))
        model(s_in, s_meta)["mano.v3d.cam.r"]
        b = {k: v.clone() for k, v in model(inputs, meta).items()}
    bad = {k: (a[k] - b[k]).abs().max().item() for k in a if not torch.equal(a[k], b[k])}
    print("bz", bz, "differing keys:", len(bad), sorted(bad.items(), key=lambda kv: -kv[1])[:3])
