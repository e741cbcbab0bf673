#!/usr/bin/env python3
"""Find the first launch whose output differs between two identical forwards separated by a forward at another batch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch, hands_amd
from hands_amd.engine import ConvEngine
model = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
model.overlap_trunks = False; model.async_tail = False
log = []
def region(t, off, rows, ps, width):
    return torch.as_strided(t.view(-1), (rows, width), (ps, 1), off).double()
orig_conv = ConvEngine.conv
def conv(self, L, pc, x, B, H, W, out, relu, stream, res=None, in_ps=None, out_ps=None, res_ps=None, x_off=0, out_off=0, res_off=0, **kw):
    r = orig_conv(self, L, pc, x, B, H, W, out, relu, stream, res=res, in_ps=in_ps, out_ps=out_ps, res_ps=res_ps, x_off=x_off, out_off=out_off, res_off=res_off, **kw)
    torch.cuda.synchronize()
    Ho, Wo = r
    o = region(out, out_off, B * Ho * Wo, out_ps or pc.Cout, pc.Cout)
    xi = region(x, x_off, B * H * W, in_ps or pc.Cin, pc.Cin)
    log.append(("conv", pc.Cin, pc.Cout, pc.KH, B, H, res is not None, relu, o.sum().item(), o.abs().sum().item(), xi.sum().item(), xi.abs().sum().item()))
    return r
ConvEngine.conv = conv
orig_dual = ConvEngine.conv_dual
def conv_dual(self, L, pc, split, x, x2, B, Ho, Wo, H2, W2, out, stream, act=1, out_off=0):
    r = orig_dual(self, L, pc, split, x, x2, B, Ho, Wo, H2, W2, out, stream, act=act, out_off=out_off)
    torch.cuda.synchronize()
    o = region(out, out_off, B * Ho * Wo, pc.Cout, pc.Cout)
    log.append(("dual", pc.Cin, pc.Cout, 1, B, Ho, False, act, o.sum().item(), o.abs().sum().item(), 0, 0))
    return r
ConvEngine.conv_dual = conv_dual
bz = 64
inputs, meta = hands_amd.synthetic_inputs(bz, 0, device=torch.device("cuda"))
s_in, s_meta = hands_amd.synthetic_inputs(2, 1, device=torch.device("cuda"))
with torch.no_grad():
    model(inputs, meta)["mano.v3d.cam.r"]; log.clear()
    model(inputs, meta)["mano.v3d.cam.r"]; A = list(log); log.clear()
    model(s_in, s_meta)["mano.v3d.cam.r"]; log.clear()
    model(inputs, meta)["mano.v3d.cam.r"]; B = list(log)
print(len(A), len(B))
for i, (a, b) in enumerate(zip(A, B)):
    if a != b:
        print("first difference at launch", i, "of", len(A))
        print("  A", a)
        print("  B", b)
        print("  previous", A[i - 1] if i else None)
        break
else:
    print("no per-launch difference")
