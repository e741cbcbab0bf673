#!/usr/bin/env python3
"""Device time of hands_mano_heads_f32 alone: N back-to-back launches on fixed buffers between two HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
import hands_amd
from hands_amd import _lib
from hands_amd._lib import ManoOut, ManoSide, check, ptr
from hands_amd.packing import pack_mano

dev = torch.device("cuda:0")
L = _lib.lib()
SIZES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (128, 256, 1024, 4096)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200      # (PMC passes: few launches, e.g. "4096 5")
for bz in SIZES:
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
    for _ in range(min(20, N)):
        check(L.hands_mano_heads_f32(sides, 2, ptr(K), 10, 224.0, 0.1, bz, 0, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        check(L.hands_mano_heads_f32(sides, 2, ptr(K), 10, 224.0, 0.1, bz, 0, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N
    print(f"bz={bz}: {us:.1f} us per launch ({2 * bz} hands), {2 * bz / us:.1f} hands/us, "
          f"{2 * bz * 1.17e6 / us / 1e6:.1f} TFLOP/s, {2 * bz * 10.2e3 / us / 1e3:.0f} GB/s")
