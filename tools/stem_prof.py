#!/usr/bin/env python3
"""Phase durations of stem_pool_planar_kernel inside real hands_light forwards (dev tool; python tools/instrument.py stem).
usage: HANDS_HIP_LIB=build_ab/prof_stem.so python tools/stem_prof.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np, torch, hands_amd
raw = C.CDLL(os.environ["HANDS_HIP_LIB"])
model = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
model.overlap_trunks = False
inputs, meta = hands_amd.synthetic_inputs(256, 0, device=torch.device("cuda"))
with torch.no_grad():
    for _ in range(2): model(inputs, meta)["mano.v3d.cam.r"]
torch.cuda.synchronize()
p = np.zeros(8192 * 4, dtype=np.uint64); raw.hands_debug_sprof(C.c_void_p(p.ctypes.data))
p = p.reshape(-1, 4).astype(np.int64); p = p[p[:, 0] > 0]
d = np.diff(p, axis=1) * 0.01
print("blocks", len(p), "fill %.2f us  gemm %.2f us  pool %.2f us  total %.2f" % (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d.sum(1).mean()))
