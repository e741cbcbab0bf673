#!/usr/bin/env python3
"""Per-workgroup timeline of ONE conv_igemm launch (dev tool; needs a library built from a timestamp-instrumented
copy of conv_igemm.hip: python tools/instrument.py tile).
usage: HANDS_HIP_LIB=build_ab/prof_tile.so python tools/prof_tile.py B,Cin,H,Cout,k,stride,pad,res [...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
from hands_amd import _lib
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_conv

L = _lib.lib()
raw = C.CDLL(os.environ["HANDS_HIP_LIB"])
eng = ConvEngine()
eng.stream_k = False
dev = "cuda"
stream = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[1:]:
    B, Cin, H, Cout, k, st, pad, use_res = [int(v) for v in spec.split(",")]
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), st, pad, dev)
    x = torch.randn(B, H, H, Cin, device=dev)
    Ho = (H + 2 * pad - k) // st + 1
    out = torch.empty(B, Ho, Ho, Cout, device=dev)
    res = torch.randn(B, Ho, Ho, Cout, device=dev) if use_res else None
    for _ in range(3):
        eng.conv(L, pc, x, B, H, H, out, True, stream, res=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.conv(L, pc, x, B, H, H, out, True, stream, res=res)
    e1.record()
    torch.cuda.synchronize()
    prof = np.zeros(16384 * 8, dtype=np.uint64)
    steps = np.zeros(64 * 256, dtype=np.uint64)
    raw.hands_debug_prof(C.c_void_p(prof.ctypes.data), C.c_void_p(steps.ctypes.data))
    bm, bn = (256, 64) if Cout <= 64 else (128, 128)
    nt = min(-(-B * Ho * Ho // bm) * -(-Cout // bn), 16384)
    p = prof.reshape(-1, 8)[:nt].astype(np.int64)
    t0 = p[:, 0].min()
    tick = 0.01  # us per s_memrealtime tick (100 MHz)
    start = (p[:, 0] - t0) * tick
    end = (p[:, 4] - t0) * tick
    setup = (p[:, 1] - p[:, 0]) * tick
    pro = (p[:, 2] - p[:, 1]) * tick
    loop = (p[:, 3] - p[:, 2]) * tick
    epi = (p[:, 4] - p[:, 3]) * tick
    nk = pc.Kpad // 16
    if p[:, 6].min() > 0:    # finer epilogue stamps (variant prof2): barrier wait, first half, second half
        ebar = (p[:, 6] - p[:, 3]) * tick
        eh0 = (p[:, 7] - p[:, 6]) * tick
        eh1 = (p[:, 4] - p[:, 7]) * tick
        print(f"   epilogue split (us): barrier wait {ebar.mean():.2f}  j=0 half {eh0.mean():.2f}  j=1 half {eh1.mean():.2f}")
    print(f"== {spec}: event {e0.elapsed_time(e1) * 1e3:.1f} us, tiles {nt}, k-steps {nk}; span {end.max():.1f} us")
    print(f"   per tile (us)  setup {setup.mean():.2f}  prologue {pro.mean():.2f}  loop {loop.mean():.2f} ({loop.mean() / nk:.3f}/step)"
          f"  epilogue {epi.mean():.2f}  total {(end - start).mean():.2f}")
    hw = p[:, 5] & 0xffffffff
    xcc = (p[:, 5] >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    print("   wave_id of wave 0:", np.bincount(hw & 0xf), " simd_id:", np.bincount((hw >> 4) & 3), " distinct CUs:", len(np.unique(cu)))
    # concurrency: tiles resident per CU over time (sampled)
    order = np.argsort(start)
    q = np.linspace(0, end.max(), 200)
    resident = [(np.sum((start <= t) & (end > t))) for t in q]
    print("   resident tiles over the launch (min/mean/max):", min(resident[5:-20]), int(np.mean(resident[5:-20])), max(resident))
    # by dispatch round: tile duration and phases for early vs late tiles
    for lo, hi in ((0, 1024), (1024, 2048), (2048, 3072), (nt - 1024, nt)):
        if hi <= nt and lo >= 0:
            sel = order[lo:hi]
            print(f"   tiles #{lo}-{hi} by start time: start {start[sel].min():.1f}..{start[sel].max():.1f} us  pro {pro[sel].mean():.2f}"
                  f"  loop {loop[sel].mean():.2f}  epi {epi[sel].mean():.2f}")
    # one CU: its tiles in time order
    c0 = cu[order[0]]
    mine = [i for i in order if cu[i] == c0][:12]
    print("   first CU timeline (start, +pro, +loop, +epi, wave_id):",
          " | ".join(f"{start[i]:.1f} {pro[i]:.1f} {loop[i]:.1f} {epi[i]:.1f} w{hw[i] & 0xf}" for i in mine))
    s = steps.reshape(64, 256).astype(np.int64)
    nrec = min(nk - 1, 255)
    if nrec > 2:
        d = np.diff(s[:, :nrec], axis=1) * tick
        ok = (s[:, :nrec] > 0).all(axis=1)
        d = d[ok]
        if len(d):
            m = d.mean(axis=0)
            chunks = [m[i:i + 8].mean() for i in range(0, len(m), 8)]
            print("   us per k-step (mean over sampled tiles, groups of 8 steps):", " ".join(f"{c:.2f}" for c in chunks))
