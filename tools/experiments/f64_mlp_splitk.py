import sys, os
sys.path.insert(0, "/root/repo")
import torch
from hands_amd import _lib
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_linear
L = _lib.lib()
g = torch.Generator().manual_seed(0)
for (N, K) in ((1024, 1024), (512, 1024), (112, 512)):
    pc = pack_linear(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), "cuda")
    pc.acc64 = True
    x = torch.randn(64, K, generator=g).cuda()
    out = torch.empty(64, pc.Cout, device="cuda")
    for S in (1, 2, 4, 8, 16):
        eng = ConvEngine()
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            eng.conv(L, pc, x, 64, 1, 1, out, 3, st, splitk_n=S)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            eng.conv(L, pc, x, 64, 1, 1, out, 3, st, splitk_n=S)
        e1.record(); torch.cuda.synchronize()
        print(f"{K}->{N} fp64, S={S}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)
