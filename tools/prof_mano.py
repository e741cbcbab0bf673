#!/usr/bin/env python3
"""Phase profile of ONE hands_mano_heads_f32 launch (dev tool; needs `python tools/instrument.py mano`).
usage: HANDS_HIP_LIB=build_ab/prof_mano.so python tools/prof_mano.py [crops per side, default 1024]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import hands_amd
from hands_amd import _lib
from hands_amd._lib import ManoOut, ManoSide, check, ptr
from hands_amd.packing import pack_mano

L = _lib.lib()
raw = C.CDLL(os.environ["HANDS_HIP_LIB"])
bz = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
mps = [pack_mano(hands_amd.synthetic_mano_asset(s), dev) for s in (True, False)]
g = torch.Generator().manual_seed(0)
q, _ = torch.linalg.qr(torch.randn(2 * bz * 16, 3, 3, generator=g))
rot = (q * torch.linalg.det(q)[:, None, None]).contiguous().to(dev)
betas, cam = torch.randn(2 * bz, 10, device=dev), torch.tensor([1.0, 0, 0], device=dev).repeat(2 * bz, 1).contiguous()
K = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]], device=dev).repeat(bz, 1, 1).contiguous()
sides = (ManoSide * 2)()
keep = []
for s in range(2):
    mp = mps[s]
    c = _lib.ManoConsts(ptr(mp["pose_mean"]), ptr(mp["J_template"]), ptr(mp["J_shapedirs"]), ptr(mp["lbs_weights"]), ptr(mp["tip_ids"]))
    o = [torch.empty(bz, n, device=dev) for n in (2334, 63, 2334, 63, 42, 3)]
    keep.append(o)
    sides[s] = ManoSide(c, ptr(mp["blend"].w), ptr(mp["blend"].bias), ptr(rot, s * bz * 144), ptr(betas, s * bz * 10),
                        ptr(cam, s * bz * 3), ManoOut(*[ptr(t) for t in o]))
st = torch.cuda.current_stream().cuda_stream
run = lambda: check(L.hands_mano_heads_f32(sides, 2, ptr(K), 10, 224.0, 0.1, bz, 0, st))
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
prof = np.zeros(16384 * 8, dtype=np.uint64)
raw.hands_debug_mprof(C.c_void_p(prof.ctypes.data))
p = prof.reshape(-1, 8).astype(np.int64)
p = p[p[:, 5] > 0]
us = lambda v: v.astype(np.float64) * 10.0 / 1e3          # s_memrealtime: 100 MHz
life = us(p[:, 5] - p[:, 0])
print(f"{2 * bz} hands: {e0.elapsed_time(e1) * 1e3:.1f} us per launch, {len(p)} waves ({len(p) // 4} workgroups), chunks per workgroup "
      f"{p[:, 6].min()}-{p[:, 6].max()}, wave life {life.mean():.1f} us (p10 {np.percentile(life, 10):.1f}, p90 {np.percentile(life, 90):.1f}), "
      f"launch span {(p[:, 5].max() - p[:, 0].min()) * 10.0 / 1e3:.1f} us, first entry -> last entry {(p[:, 0].max() - p[:, 0].min()) * 10.0 / 1e3:.1f} us")
for nm, v in (("pose / FK (+ barriers)", us(p[:, 1] - p[:, 0])), ("blend products (+ barrier)", us(p[:, 2])), ("skinning products", us(p[:, 3])),
              ("write-back (+ barrier)", us(p[:, 4])), ("tail (tips)", life - us(p[:, 1] - p[:, 0]) - us(p[:, 2]) - us(p[:, 3]) - us(p[:, 4]))):
    print(f"   {nm:28s} {v.mean():6.2f} us  ({100 * v.mean() / life.mean():4.1f} %)   p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f}")
hw = p[:, 7]
cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf)
t0, t1 = p[:, 0].min(), p[:, 5].max()
for frac in (0.25, 0.5, 0.75):
    t = t0 + frac * (t1 - t0)
    m = (p[:, 0] <= t) & (p[:, 5] > t)
    _, cnt = np.unique(cu[m], return_counts=True)
    print(f"   at {frac:.2f} of the launch: {m.sum()} waves alive on {len(cnt)} CUs; CUs holding k waves: "
          + " ".join(f"{k}:{v}" for k, v in enumerate(np.bincount(cnt)) if v))
