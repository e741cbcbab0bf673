"""GPU tests of hands_conv3x3_winograd4_f32 (csrc/conv_wino4.hip): Winograd F(4x4, 3x3) on the fp32 matrix cores.

Same layer as hands_conv3x3_winograd_f32 / the direct kernel (conv2 / bn2 / relu of a stride-1 Bottleneck,
src/nets/backbone/resnet.py:140-142) with 2.25 multiplications per output.  F(4x4)'s transforms carry constants up to 8, so its
per-layer rounding error is ~10-20x that of F(2x2) (measured 1e-6 ... 1e-5 of the output scale against 2e-7 ... 1e-6): checked
(i) EVERY output against an fp64 convolution at 5e-5 of the output scale, on every block geometry (D = 7 linear with and without
virtual rows, D = 4 and D = 2 rectangular, partial tiles, blocks that straddle images, pixel strides); (ii) bit-for-bit
batch-size invariance and run-to-run determinism; (iii) the engine's routing and its F(2x2) fallback; (iv) end to end in
tests/test_gpu_parity.py / test_gpu_parity_sweep.py through HandsLight (the stages listed in HandsLight.winograd4_stages)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from hands_amd import _lib
from hands_amd._lib import ConvDesc, check, ptr
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_conv

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 5e-5          # of the output scale (F(2x2) / direct: 3e-5 in tests/test_gpu_winograd.py)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _wino4(x_nhwc, pc, act, in_ps=None, out_ps=None):
    L = _lib.lib()
    B, H, W, _ = x_nhwc.shape
    ips, ops = in_ps or pc.Cin, out_ps or pc.Cout
    xin = torch.full((B, H, W, ips), float("nan"), device=DEV)
    xin[..., : pc.Cin] = x_nhwc.to(DEV)
    out = torch.full((B, H, W, ops), float("nan"), device=DEV)
    d = ConvDesc(B, H, W, pc.Cin, H, W, pc.Cout, 3, 3, 1, 1, ips, ops, 0, pc.Kpad, int(act))
    assert L.hands_conv3x3_winograd4_supported(C.byref(d)) == 1
    check(L.hands_conv3x3_winograd4_f32(C.byref(d), ptr(xin), ptr(pc.wino4), ptr(pc.bias), ptr(out), _stream()), "wino4")
    torch.cuda.synchronize()
    return out.cpu()


def _ref(x_nhwc, w, bias, act):
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1)
    if act == 1:
        y = F.relu(y)
    elif act == 3:
        y = F.leaky_relu(y, 0.01)
    return y.permute(0, 2, 3, 1)


def _case(B, Cin, H, W, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    return x, w, bias


CASES = [
    # B, Cin, H, W, Cout, act
    (3, 64, 56, 56, 64, 1),      # layer1: 14 tiles per row -> linear order, two virtual rows of 7 per tile row
    (5, 128, 28, 28, 128, 1),    # layer2: 7 tiles per row -> linear order, blocks straddle rows and images
    (7, 256, 14, 14, 256, 1),    # layer3: 4 tiles per row (16 x 16 cover), D = 4 blocks of 8 tile rows
    (9, 512, 7, 7, 512, 1),      # layer4: 2 tiles per row (8 x 8 cover), D = 2 blocks of 16 tile rows
    (2, 16, 1, 1, 32, 0),        # one pixel
    (3, 32, 2, 3, 32, 3),        # tiny, LeakyReLU
    (2, 16, 5, 5, 64, 1),        # 2 x 2 tiles, partial
    (1, 48, 9, 11, 96, 0),       # odd x odd, 3 tile columns
    (2, 32, 20, 19, 32, 3),      # 5 tile columns: two D = 4 segments, the second partial
    (4, 64, 13, 14, 32, 1),      # 4 columns, odd height
    (1, 32, 64, 64, 64, 3),      # handoccnet-sized map: 16 tile columns
    (33, 64, 8, 8, 64, 1),       # tile rows not a multiple of the block height
    (2, 24, 28, 28, 32, 1),      # Cin % 16 != 0 (8-channel stages)
    (3, 64, 55, 57, 64, 1),      # 14 x 15 tiles: linear geometry not applicable (nw = 15) -> D = 4
    (2, 32, 26, 26, 32, 1),      # LINEAR geometry (nw = 7) with a PARTIAL last tile column (26 = 6 x 4 + 2) and a partial last tile row
    (2, 32, 54, 55, 64, 3),      # LINEAR geometry with two virtual rows (nw = 14), width 55 = 13 x 4 + 3, height 54 = 13 x 4 + 2
    (3, 64, 25, 27, 32, 0),      # nw = 7, width 27, height 25: the 4 otx + oj < W guard on three of four columns of the last tile
]


@pytest.mark.parametrize("case", CASES)
def test_winograd4_every_output_vs_fp64(case):
    B, Cin, H, W, Cout, act = case
    x, w, bias = _case(B, Cin, H, W, Cout, hash(case) % (2 ** 31))
    pc = pack_conv(w, bias, 1, 1, DEV, winograd4=True)
    assert pc.wino4 is not None and pc.wino4.numel() == 36 * Cout * Cin
    got = _wino4(x, pc, act)
    ref = _ref(x, w, bias, act)
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item()
    assert err <= TOL * scale, (case, err, scale)


@pytest.mark.parametrize("case", [(2, 32, 26, 27, 32), (2, 32, 54, 53, 64), (5, 64, 28, 28, 64)])
def test_winograd4_linear_geometries_with_strided_layouts(case):
    """The linear tile orders (7 and 14 tiles per row: the swizzled LDS-DMA fill and the per-column output guard) on maps whose
    width is not a multiple of 4, through in / out pixel strides larger than the channel counts."""
    B, Cin, H, W, Cout = case
    x, w, bias = _case(B, Cin, H, W, Cout, sum(case))
    pc = pack_conv(w, bias, 1, 1, DEV, winograd4=True)
    got = _wino4(x, pc, 1, in_ps=Cin + 12, out_ps=Cout + 4)
    assert torch.isnan(got[..., Cout:]).all()
    ref = _ref(x, w, bias, 1)
    assert (got[..., :Cout].double() - ref).abs().max().item() <= TOL * ref.abs().max().item()


def test_winograd4_honours_pixel_strides_and_the_engine_routes_to_it():
    B, Cin, H, W, Cout = 4, 64, 14, 14, 96
    x, w, bias = _case(B, Cin, H, W, Cout, 7)
    pc = pack_conv(w, bias, 1, 1, DEV, winograd4=True)
    got = _wino4(x, pc, 1, in_ps=Cin + 8, out_ps=Cout + 12)
    assert torch.isnan(got[..., Cout:]).all()           # nothing written beyond the layer's channels
    ref = _ref(x, w, bias, 1)
    assert (got[..., :Cout].double() - ref).abs().max().item() <= TOL * ref.abs().max().item()
    L = _lib.lib()
    seen = []
    eng = ConvEngine()
    eng.winograd4 = True
    eng.hook = lambda phase, pc_, npix, st, has_res, kernel: seen.append(kernel)
    out = torch.empty(B, H, W, Cout, device=DEV)
    eng.conv(L, pc, x.to(DEV), B, H, W, out, True, _stream())
    torch.cuda.synchronize()
    assert seen == ["conv_wino4_f32_kernel"] * 2 and eng.last_wino_macs == L.hands_conv3x3_winograd4_executed_macs(
        C.byref(ConvDesc(B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, Cin, Cout, 0, pc.Kpad, 1)))
    assert torch.equal(out.cpu(), got[..., :Cout])
    # engine.winograd4 off, or a layer packed without the F(4x4) weights: F(2x2)
    for e2, p2 in ((ConvEngine(), pc), (eng, pack_conv(w, bias, 1, 1, DEV))):
        seen.clear()
        e2.hook = lambda phase, pc_, npix, st, has_res, kernel: seen.append(kernel)
        e2.conv(L, p2, x.to(DEV), B, H, W, out, True, _stream())
        torch.cuda.synchronize()
        assert seen == ["conv_wino_f32_kernel"] * 2


@pytest.mark.parametrize("geom", [(64, 56), (128, 28), (256, 14), (512, 7)])
def test_winograd4_is_batch_size_invariant_and_deterministic(geom):
    Cch, H = geom
    x, w, bias = _case(37, Cch, H, H, Cch, 11)
    pc = pack_conv(w, bias, 1, 1, DEV, winograd4=True)
    big = _wino4(x, pc, 1)
    again = _wino4(x, pc, 1)
    small = _wino4(x[:2], pc, 1)
    assert torch.equal(big, again)
    assert torch.equal(big[:2], small)


def test_winograd4_many_workgroups_every_output():
    """More workgroups than the chip holds at once (layer1 at 48 images: 294 tile blocks x 2 channel blocks, one per CU at a time)."""
    B, Cch, H = 48, 64, 56
    x, w, bias = _case(B, Cch, H, H, Cch, 3)
    pc = pack_conv(w, bias, 1, 1, DEV, winograd4=True)
    got = _wino4(x, pc, 1)
    ref = _ref(x, w, bias, 1)
    assert (got.double() - ref).abs().max().item() <= TOL * ref.abs().max().item()


def test_winograd4_rejects_what_it_cannot_take():
    L = _lib.lib()
    ok = ConvDesc(2, 8, 8, 32, 8, 8, 32, 3, 3, 1, 1, 32, 32, 0, 288, 1)
    assert L.hands_conv3x3_winograd4_supported(C.byref(ok)) == 1
    for field, val in (("stride", 2), ("pad", 0), ("KH", 1), ("Cin", 12), ("Cout", 48), ("act", 2), ("H", 0), ("B", 0)):
        d = ConvDesc(2, 8, 8, 32, 8, 8, 32, 3, 3, 1, 1, 32, 32, 0, 288, 1)
        setattr(d, field, val)
        assert L.hands_conv3x3_winograd4_supported(C.byref(d)) == 0, field
        assert L.hands_conv3x3_winograd4_executed_macs(C.byref(d)) == 0
