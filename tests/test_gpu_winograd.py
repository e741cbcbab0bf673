"""GPU tests of hands_conv3x3_winograd_f32 (csrc/conv_wino.hip): Winograd F(2x2, 3x3) on the fp32 matrix cores.

The kernel re-associates the 3x3 sum, so it is checked (i) against an fp64 convolution at the same per-op tolerance as
the direct kernel (2e-5 of the output scale), EVERY output; (ii) against the direct implicit-GEMM kernel; (iii) for the
properties the direct kernel has: batch-size invariance bit for bit, run-to-run determinism, pixel strides, odd maps,
tiles that straddle images, maps of every block geometry (D = 4 / 8 rectangular blocks, D = 7 linear order)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hands_amd import _lib
from hands_amd._lib import ConvDesc, check, ptr
from hands_amd.engine import ConvEngine
from hands_amd.packing import pack_conv

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _wino(x_nhwc, pc, act, in_ps=None, out_ps=None):
    L = _lib.lib()
    B, H, W, _ = x_nhwc.shape
    ips, ops = in_ps or pc.Cin, out_ps or pc.Cout
    xin = torch.full((B, H, W, ips), float("nan"), device=DEV)
    xin[..., : pc.Cin] = x_nhwc.to(DEV)
    out = torch.full((B, H, W, ops), float("nan"), device=DEV)
    d = ConvDesc(B, H, W, pc.Cin, H, W, pc.Cout, 3, 3, 1, 1, ips, ops, 0, pc.Kpad, int(act))
    assert L.hands_conv3x3_winograd_supported(C.byref(d)) == 1
    check(L.hands_conv3x3_winograd_f32(C.byref(d), ptr(xin), ptr(pc.wino), ptr(pc.bias), ptr(out), _stream()), "wino")
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


WINO_CASES = [
    # B, Cin, H, W, Cout, act
    (3, 64, 56, 56, 64, 1),      # layer1: nw = 28 -> D = 4 blocks (8 tile rows x 4 columns), 7 column segments
    (5, 128, 28, 28, 128, 1),    # layer2: nw = 14 -> D = 8 blocks, second segment partial
    (7, 256, 14, 14, 256, 1),    # layer3: nw = 7 -> linear order, blocks straddle images
    (9, 512, 7, 7, 512, 1),      # layer4: odd map, nw = 4 (8x8 cover), last tile row / column half outside
    (2, 16, 1, 1, 32, 0),        # one pixel
    (3, 32, 2, 3, 32, 3),        # tiny, LeakyReLU
    (2, 16, 5, 5, 64, 1),        # nw = 3
    (1, 48, 9, 11, 96, 0),       # odd x odd, nw = 6 -> D = 4 with a partial segment
    (2, 32, 20, 19, 32, 3),      # nw = 10 -> D = 8
    (4, 64, 13, 14, 32, 1),      # nw = 7 linear with an odd height
    (1, 32, 64, 64, 64, 3),      # handoccnet-sized map: nw = 32
    (33, 64, 8, 8, 64, 1),       # rows not a multiple of the block height
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_every_output_vs_fp64(case):
    B, Cin, H, W, Cout, act = case
    x, w, bias = _case(B, Cin, H, W, Cout, hash(case) % (2 ** 31))
    pc = pack_conv(w, bias, 1, 1, DEV)
    assert pc.wino is not None and pc.wino.numel() == 16 * Cout * Cin
    got = _wino(x, pc, act)
    ref = _ref(x, w, bias, act)
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item()
    assert err <= 3e-5 * scale, (case, err, scale)


def test_winograd_matches_the_direct_kernel_and_honours_pixel_strides():
    B, Cin, H, W, Cout = 4, 64, 14, 14, 96
    x, w, bias = _case(B, Cin, H, W, Cout, 7)
    pc = pack_conv(w, bias, 1, 1, DEV)
    got = _wino(x, pc, 1, in_ps=Cin + 8, out_ps=Cout + 12)
    assert torch.isnan(got[..., Cout:]).all()           # nothing written beyond the layer's channels
    L = _lib.lib()
    eng = ConvEngine()
    eng.winograd = False
    out = torch.empty(B, H, W, Cout, device=DEV)
    eng.conv(L, pc, x.to(DEV), B, H, W, out, True, _stream())
    torch.cuda.synchronize()
    d = (got[..., :Cout] - out.cpu()).abs().max().item()
    assert d <= 2e-5 * out.abs().max().item(), d
    # and the engine routes this layer to the Winograd kernel by default
    seen = []
    eng2 = ConvEngine()
    eng2.hook = lambda phase, pc_, npix, st, has_res, kernel: seen.append(kernel)
    out2 = torch.empty(B, H, W, Cout, device=DEV)
    eng2.conv(L, pc, x.to(DEV), B, H, W, out2, True, _stream())
    torch.cuda.synchronize()
    assert seen == ["conv_wino_f32_kernel"] * 2
    assert torch.equal(out2.cpu(), got[..., :Cout])


@pytest.mark.parametrize("geom", [(64, 56), (256, 14), (512, 7), (128, 28)])
def test_winograd_is_batch_size_invariant_and_deterministic(geom):
    Cch, H = geom
    x, w, bias = _case(37, Cch, H, H, Cch, 11)
    pc = pack_conv(w, bias, 1, 1, DEV)
    big = _wino(x, pc, 1)
    again = _wino(x, pc, 1)
    small = _wino(x[:2], pc, 1)
    assert torch.equal(big, again)
    assert torch.equal(big[:2], small)


def test_winograd_many_tiles_every_output():
    """More workgroups than the chip holds at once (layer1 at 48 images: 9408 workgroups x 2 channel blocks)."""
    B, Cch, H = 48, 64, 56
    x, w, bias = _case(B, Cch, H, H, Cch, 3)
    pc = pack_conv(w, bias, 1, 1, DEV)
    got = _wino(x, pc, 1)
    ref = _ref(x, w, bias, 1)
    err = (got.double() - ref).abs().max().item()
    assert err <= 3e-5 * ref.abs().max().item(), err


def test_winograd_misaligned_pointers_fall_back_to_the_direct_kernel():
    """16-byte accesses: the C entry refuses a pointer that is not 16-byte aligned, and the engine routes such a call
    (a view at an odd float offset) to the direct kernel instead."""
    L = _lib.lib()
    x, w, bias = _case(2, 32, 6, 6, 32, 5)
    pc = pack_conv(w, bias, 1, 1, DEV)
    flat = torch.zeros(2 * 6 * 6 * 32 + 8, device=DEV)
    flat[1:1 + x.numel()] = x.reshape(-1).to(DEV)
    out = torch.empty(2, 6, 6, 32, device=DEV)
    d = ConvDesc(2, 6, 6, 32, 6, 6, 32, 3, 3, 1, 1, 32, 32, 0, pc.Kpad, 1)
    assert L.hands_conv3x3_winograd_f32(C.byref(d), ptr(flat, 1), ptr(pc.wino), ptr(pc.bias), ptr(out), _stream()) == 10001
    seen = []
    eng = ConvEngine()
    eng.hook = lambda phase, pc_, npix, st, has_res, kernel: seen.append(kernel)
    eng.conv(L, pc, flat, 2, 6, 6, out, True, _stream(), x_off=1)
    torch.cuda.synchronize()
    assert seen == ["conv_igemm_f32_kernel"] * 2
    ref = _ref(x, w, bias, 1)
    assert (out.cpu().double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()


def test_winograd_rejects_what_it_cannot_run():
    L = _lib.lib()
    ok = ConvDesc(2, 14, 14, 64, 14, 14, 64, 3, 3, 1, 1, 64, 64, 0, 576, 1)
    assert L.hands_conv3x3_winograd_supported(C.byref(ok)) == 1
    for field, val in (("stride", 2), ("pad", 0), ("KH", 1), ("Cin", 24), ("Cout", 48), ("act", 2), ("act", 1 | 0x100)):
        d = ConvDesc(2, 14, 14, 64, 14, 14, 64, 3, 3, 1, 1, 64, 64, 0, 576, 1)
        setattr(d, field, val)
        assert L.hands_conv3x3_winograd_supported(C.byref(d)) == 0, field


@pytest.mark.parametrize("shape", [(512, 64, 56), (512, 128, 28), (512, 256, 14), (256, 512, 7)])
def test_winograd_full_size_against_the_direct_kernel(shape):
    """BASELINE configs[1] sizes (the launches of a bz = 256 forward: 512 hand crops / 256 images per trunk job): EVERY
    output of the Winograd launch against the direct implicit-GEMM kernel on the same device tensors (the direct kernel is
    pinned to fp64 convolutions by tests/test_gpu_parity.py), plus linearity in the input -- a size-independent property."""
    B, Cch, H = shape
    L = _lib.lib()
    g = torch.Generator().manual_seed(B + Cch)
    w = torch.randn(Cch, Cch, 3, 3, generator=g) / (Cch * 9) ** 0.5
    bias = torch.randn(Cch, generator=g)
    pc = pack_conv(w, bias, 1, 1, DEV)
    gd = torch.Generator(device=DEV).manual_seed(1)
    x1 = torch.randn(B, H, H, Cch, device=DEV, generator=gd)
    x2 = torch.randn(B, H, H, Cch, device=DEV, generator=gd)
    wino, direct = ConvEngine(), ConvEngine()
    direct.winograd = False
    run = lambda eng, x, act: (lambda o: (eng.conv(L, pc, x, B, H, H, o, act, _stream()), o)[1])(torch.empty(B, H, H, Cch, device=DEV))
    yw, yd = run(wino, x1, 1), run(direct, x1, 1)
    torch.cuda.synchronize()
    scale = yd.abs().max().item()
    assert (yw - yd).abs().max().item() <= 2e-5 * scale
    # linearity (activation off): conv(x1 + x2) + bias = conv(x1) + conv(x2)  [each side carries the bias once / twice]
    y1, y2, y12 = run(wino, x1, 0), run(wino, x2, 0), run(wino, x1 + x2, 0)
    torch.cuda.synchronize()
    b = pc.bias[:Cch].view(1, 1, 1, Cch)
    assert (y12 + b - y1 - y2).abs().max().item() <= 3e-5 * y12.abs().max().item()
