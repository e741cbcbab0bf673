"""Ad-hoc edge cases through the HIP path against the oracle (dev tool, GPU box): bz = 1 / 3 / 33 / 64, every sample flipped, float64 and
non-contiguous inputs, per-sample intrinsics; HAMER / HandOccNet at bz = 1 / 3.  usage: python tools/edge_cases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch, hands_amd
from oracle import hands_oracle as O
dev = "cuda"
m = hands_amd.apply_recipe(hands_amd.HandsLight())
sd = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(dev).eval()
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
def check(bz, seed, mut=None, tag=""):
    ci, cm = hands_amd.synthetic_inputs(bz, seed)
    if mut: mut(ci, cm)
    ref = O.hands_light_forward(sd, ar, al, {k: v.float() for k, v in ci.items()}, cm)
    di = {k: v.to(dev) for k, v in ci.items()}; dm = {k: v.to(dev) for k, v in cm.items()}
    if "noncontig" in tag:
        di = {k: (torch.cat([v, v], -1)[..., : v.shape[-1]] if v.ndim == 4 else v) for k, v in di.items()}
    out = m(di, dm); torch.cuda.synchronize()
    e = max((out[f"mano.vertices.{h}"].cpu() - ref[f"mano.vertices.{h}"]).abs().max().item() for h in "rl")
    print(f"bz={bz} seed={seed} {tag}: max vertex err {e:.2e}", "OK" if e < 1e-6 else "FAIL")
check(1, 3)
check(3, 4)
check(7, 5, lambda ci, cm: cm.__setitem__("is_flipped", torch.ones(7, dtype=torch.long)), "all flipped")
check(2, 6, lambda ci, cm: ci.update({k: v.double() for k, v in ci.items()}), "float64 inputs")
check(2, 7, None, "noncontig views")
check(5, 8, lambda ci, cm: cm.__setitem__("intrinsics", cm["intrinsics"] * torch.tensor([[[1.1, 1, 0.9], [1, 0.95, 1.05], [1, 1, 1]]])), "other K")
check(33, 9); check(2, 10); check(64, 11)
for cls, fwd in ((hands_amd.HAMER, None), (hands_amd.HandOccNet, None)):
    mm = hands_amd.apply_recipe(cls()).to(dev).eval()
    for bz in (1, 3):
        i, mt = hands_amd.synthetic_inputs(bz, 1, device=dev)
        o = mm(i, mt); torch.cuda.synchronize()
        print(cls.__name__, bz, "finite:", all(torch.isfinite(v).all().item() for v in o.values()), len(o))
