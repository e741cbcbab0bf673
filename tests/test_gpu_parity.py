"""GPU parity tests: the HIP path (through the C ABI) against the oracle / golden fixtures.

Tolerances (fp32 path):
  * per-op: 2e-5 relative to the output scale (different summation order than ATen's CPU kernels);
  * end-to-end canonical vertices / joints: 1e-6 m absolute (= the north star's 1e-3 mm);
  * camera-space keys: relative 2e-6 (they sit ~1-10 m from the camera, fp32 ulp ~1e-6 m).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import hands_amd
from hands_amd import _lib
from hands_amd._lib import check, ptr
from hands_amd.engine import DEFAULT_ENGINE, ConvEngine
from hands_amd.hands_light import HandsLight
from hands_amd.mano import synthetic_mano_asset
from hands_amd.packing import fold_bn, pack_conv, pack_linear, pack_mano
from hands_amd.weights import synthetic_inputs
from oracle import hands_oracle as O
from switch_cases import SWITCH_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _run_conv(x_nchw, w, bias, stride, pad, relu, res_nchw=None, cin_pad_to=None, engine=None):
    L = _lib.lib()
    B, Cin, H, W = x_nchw.shape
    pc = pack_conv(w, bias, stride, pad, DEV, cin_pad_to=cin_pad_to)
    x = _nhwc(x_nchw)
    if cin_pad_to and cin_pad_to != Cin:
        x = F.pad(x, (0, cin_pad_to - Cin))
    x = x.to(DEV).contiguous()
    Ho = (H + 2 * pad - w.shape[2]) // stride + 1
    Wo = (W + 2 * pad - w.shape[3]) // stride + 1
    out = torch.full((B, Ho, Wo, pc.Cout), float("nan"), device=DEV)
    res = _nhwc(res_nchw).to(DEV) if res_nchw is not None else None
    (engine or DEFAULT_ENGINE).conv(L, pc, x, B, H, W, out, relu, _stream(), res=res)
    torch.cuda.synchronize()
    return out.cpu().permute(0, 3, 1, 2)[:, : w.shape[0]]


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, relu, residual
    (2, 64, 56, 56, 64, 1, 1, 0, True, False),     # 1x1, Cout=64 tile (256x64)
    (2, 64, 56, 56, 256, 1, 1, 0, True, True),     # 1x1 expand + residual + relu
    (3, 64, 28, 28, 64, 3, 1, 1, True, False),     # 3x3 pad 1
    (2, 128, 28, 28, 128, 3, 2, 1, True, False),   # 3x3 stride 2
    (2, 256, 28, 28, 512, 1, 2, 0, False, False),  # downsample 1x1 stride 2, no relu
    (1, 512, 7, 7, 2048, 1, 1, 0, True, True),     # layer4 expand, M=49 (tail rows)
    (2, 1024, 7, 7, 512, 3, 1, 0, True, False),    # feature_conv 3x3 pad 0 (7->5)
    (5, 16, 9, 11, 12, 3, 1, 1, False, False),     # odd sizes, Cout=12 (n tail)
    (3, 2304, 1, 1, 2048, 1, 1, 0, True, False),   # linear layer
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm_vs_torch(case):
    B, Cin, H, W, Cout, k, stride, pad, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=pad)
    if res is not None:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    got = _run_conv(x, w, bias, stride, pad, relu, res)
    assert got.shape == ref.shape
    err = (got.double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def _random_conv_cases(n, seed):
    import random
    rnd = random.Random(seed)
    cases = []
    while len(cases) < n:
        k = rnd.choice([1, 1, 3, 3, 5, 7])
        stride = rnd.choice([1, 1, 2, 3])
        pad = rnd.choice([0, k // 2, k // 2, 1])
        Cin = 16 * rnd.randint(1, 6)
        Cout = rnd.choice([4, 12, 20, 64, 68, 100, 128, 132, 200])
        H, W = rnd.randint(k, 23), rnd.randint(k, 23)
        if (H + 2 * pad - k) < 0 or (W + 2 * pad - k) < 0:
            continue
        cases.append((rnd.randint(1, 5), Cin, H, W, Cout, k, stride, pad, rnd.random() < 0.5, rnd.random() < 0.4))
    return cases


@pytest.mark.parametrize("case", _random_conv_cases(24, 1234))
def test_conv_igemm_random_shapes(case):
    """Ragged everything: odd maps, strides 1-3, any padding, Cout not a multiple of the 64/128 tile."""
    test_conv_igemm_vs_torch(case)


MANY_TILE_CASES = [
    # B, Cin, H, Cout, k, stride, pad, residual -- more output tiles than CUs (several workgroups resident per CU),
    # every output checked: a variant whose epilogue used raw-buffer STORES passed all small cases and corrupted a
    # few lanes of a few tiles only from 257 tiles up (found by the model-level idempotence test)
    (48, 128, 28, 128, 3, 1, 1, False),
    (64, 128, 56, 128, 3, 2, 1, False),
    (64, 64, 28, 64, 3, 1, 1, False),
    (64, 128, 28, 128, 3, 1, 1, True),
    (64, 256, 56, 128, 1, 1, 0, False),
    (40, 256, 14, 1024, 1, 1, 0, True),
    (48, 128, 28, 132, 3, 1, 1, True),      # channel tail: the general epilogue on every second tile
    (47, 64, 27, 192, 3, 1, 1, False),      # odd map, pixel tail
    (300, 1024, 7, 260, 1, 1, 0, True),     # many images per tile, channel tail
]


@pytest.mark.parametrize("case", MANY_TILE_CASES)
def test_conv_igemm_many_tiles_every_output(case):
    B, Cin, H, Cout, k, stride, pad, use_res = case
    L = _lib.lib()
    g = torch.Generator().manual_seed(B + Cin + k)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pc = pack_conv(w, bias, stride, pad, DEV)
    x = torch.randn(B, H, H, Cin, device=DEV)
    Ho = (H + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, device=DEV) if use_res else None
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double().to(DEV), bias.double().to(DEV), stride=stride, padding=pad)
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2).double()
    ref = F.relu(ref).permute(0, 2, 3, 1)
    outs = []
    for fill in (float("nan"), 7.0):          # twice: the result may not depend on what the output buffer held
        out = torch.full((B, Ho, Ho, Cout), fill, device=DEV)
        DEFAULT_ENGINE.conv(L, pc, x, B, H, H, out, True, _stream(), res=res)
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


ADDRESSING_CASES = [
    # the k-loop addresses a tile's inputs with 32-bit byte offsets from the first image a wave touches and masks
    # taps with one validity word per output pixel: tiles spanning many tiny images, 1x1 maps with padding, a 1x1
    # kernel with padding, a strided pointwise layer on 3x3 maps, a 5x5 kernel on maps smaller than the kernel
    (300, 32, 2, 2, 64, 3, 1, 1, True, False),
    (700, 16, 1, 1, 128, 3, 1, 1, False, False),
    (2, 16, 5, 5, 20, 1, 1, 1, True, False),
    (130, 48, 3, 3, 132, 1, 2, 0, True, False),
    (67, 16, 3, 4, 68, 5, 1, 2, True, True),
    (9, 32, 13, 7, 64, 3, 3, 1, False, False),
]


@pytest.mark.parametrize("case", ADDRESSING_CASES)
def test_conv_igemm_addressing_edges(case):
    test_conv_igemm_vs_torch(case)


def test_conv_geometry_limits_are_rejected():
    """Padded convolutions take KH, KW <= 15 (include/hands_hip.h); beyond that the entry point refuses."""
    from hands_amd._lib import ConvDesc, ptr
    import ctypes as C
    L = _lib.lib()
    x = torch.zeros(1, 20, 20, 16, device=DEV)
    w = torch.zeros(128, 16 * 16 * 16, device=DEV)
    b = torch.zeros(128, device=DEV)
    out = torch.zeros(1, 7, 7, 4, device=DEV)
    d = ConvDesc(1, 20, 20, 16, 7, 7, 4, 16, 16, 1, 1, 16, 4, 0, 16 * 16 * 16, 0)
    assert L.hands_conv2d_nhwc_f32(C.byref(d), ptr(x), ptr(w), ptr(b), None, ptr(out), _stream()) != 0
    d = ConvDesc(1, 20, 20, 16, 8, 8, 4, 15, 15, 1, 1, 16, 4, 0, 15 * 15 * 16, 0)
    w = torch.zeros(128, 15 * 15 * 16, device=DEV)
    out = torch.zeros(1, 8, 8, 4, device=DEV)
    assert L.hands_conv2d_nhwc_f32(C.byref(d), ptr(x), ptr(w), ptr(b), None, ptr(out), _stream()) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("k", [1, 3])
def test_conv_igemm_input_larger_than_2gib(k):
    """Offsets are relative to the first image a wave stages, so a 2.6 GB input (absolute byte offsets past 2^31)
    is fine: the last images of the batch are checked against torch."""
    L = _lib.lib()
    B, Cin, H, Cout = 800, 256, 56, 64
    g = torch.Generator().manual_seed(k)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pc = pack_conv(w, bias, 1, k // 2, DEV)
    x = torch.randn(B, H, H, Cin, device=DEV)
    assert x.numel() * 4 > 2 ** 31
    out = torch.full((B, H, H, Cout), float("nan"), device=DEV)
    DEFAULT_ENGINE.conv(L, pc, x, B, H, H, out, True, _stream())
    torch.cuda.synchronize()
    for b0 in (0, 417, B - 2):
        ref = F.relu(F.conv2d(x[b0:b0 + 2].permute(0, 3, 1, 2).double().cpu(), w.double(), bias.double(), padding=k // 2))
        got = out[b0:b0 + 2].permute(0, 3, 1, 2).double().cpu()
        assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    del x, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0] * c[2] * c[3] <= 3000])
def test_conv_igemm_latency_mode_split_k(case):
    """Caller-chosen split-K on convolutions (small-batch serving mode): same result as the unsplit
    launch up to the fp32 re-association of S partial sums."""
    B, Cin, H, W, Cout, k, stride, pad, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=pad)
    if res is not None:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    eng = ConvEngine()
    eng.latency_mode = True
    got = _run_conv(x, w, bias, stride, pad, relu, res, engine=eng)
    again = _run_conv(x, w, bias, stride, pad, relu, res, engine=eng)
    assert not DEFAULT_ENGINE.latency_mode                        # per-engine state, nothing global was touched
    assert torch.equal(got, again)                                 # deterministic
    err = (got.double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


SK_CASES = [
    # B, Cin, H, Cout, k, stride, pad, residual        (tile counts chosen inside the stream-K window: 512 .. 16384)
    (256, 256, 14, 256, 3, 1, 1, False),     # layer3 conv2 at bz=256: 392 x 2 = 784 tiles, 144 k-steps
    (64, 512, 7, 512, 3, 1, 1, False),       # few tiles (below the window): plain launch, must still agree
    (200, 1024, 14, 256, 1, 1, 0, False),    # pointwise, 64 k-steps
    (150, 256, 14, 1024, 1, 1, 0, True),     # expand + residual + relu
    (37, 128, 28, 128, 3, 2, 1, False),      # stride 2, ragged M
    (300, 64, 14, 64, 3, 1, 1, False),       # narrow 256 x 64 tile instantiation
    (256, 512, 7, 512, 3, 1, 1, False),      # 392 tiles (1.5 per CU), 288 k-steps: 512 workgroups, ranges INSIDE a tile
    (512, 1024, 7, 512, 3, 1, 0, False),     # feature_conv: 400 tiles, 576 k-steps
    (256, 2048, 7, 512, 1, 1, 0, True),      # pointwise in the same regime, with residual
]


@pytest.mark.parametrize("case", SK_CASES)
def test_stream_k_is_bit_identical_to_the_plain_launch(case):
    """hands_conv2d_nhwc_streamk_f32: persistent workgroups with equal (tile, k-step) shares; a tile cut between
    two workgroups continues the same fp32 FMA chain, so the output must equal hands_conv2d_nhwc_f32 BIT FOR BIT
    (repeated launches on one workspace with a running epoch included)."""
    B, Cin, H, Cout, k, stride, pad, use_res = case
    L = _lib.lib()
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, H, Cin, generator=g).to(DEV)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), stride, pad, DEV)
    Ho = (H + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g).to(DEV) if use_res else None
    plain, sk = ConvEngine(), ConvEngine()
    plain.winograd = sk.winograd = False        # this test is about the direct kernel's two launch forms
    plain.stream_k, sk.stream_k = False, True
    ref = torch.full((B, Ho, Ho, pc.Cout), float("nan"), device=DEV)
    plain.conv(L, pc, x, B, H, H, ref, True, _stream(), res=res)
    d = _lib.ConvDesc(B, H, H, pc.Cin, Ho, Ho, pc.Cout, k, k, stride, pad, pc.Cin, pc.Cout, pc.Cout if use_res else 0, pc.Kpad, 1)
    G = L.hands_conv2d_streamk_grid(C.byref(d))
    for rep in range(3):
        got = torch.full((B, Ho, Ho, pc.Cout), float("nan"), device=DEV)
        sk.conv(L, pc, x, B, H, H, got, True, _stream(), res=res)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), (case, G, rep, (got - ref).abs().max().item())
    if case is SK_CASES[0]:
        assert G > 0 and G % 256 == 0, G
    if case is SK_CASES[1]:
        assert G == 0
    if case is SK_CASES[6]:
        assert G == 512, G


@pytest.mark.parametrize("case", [SK_CASES[0], SK_CASES[6], SK_CASES[8]])
def test_stream_k_fallback_recomputes_the_same_bits(case):
    """No workgroup publishes its accumulator hand-off (negative epoch = the test hook): every consumer -- tail parts
    and, with 512 workgroups on 392 tiles, the parts in the MIDDLE of a tile -- runs into its bounded wait and
    recomputes the missing k-steps itself.  Same bits as the plain launch."""
    import ctypes as C
    from hands_amd._lib import ptr
    B, Cin, H, Cout, k, stride, pad, use_res = case
    L = _lib.lib()
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(B, H, H, Cin, generator=g).to(DEV)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), stride, pad, DEV)
    Ho = (H + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g).to(DEV) if use_res else None
    plain = ConvEngine()
    plain.stream_k = False
    plain.winograd = False                      # the reference is the direct kernel's plain launch
    ref = torch.full((B, Ho, Ho, pc.Cout), float("nan"), device=DEV)
    plain.conv(L, pc, x, B, H, H, ref, True, _stream(), res=res)
    d = _lib.ConvDesc(B, H, H, pc.Cin, Ho, Ho, pc.Cout, k, k, stride, pad, pc.Cin, pc.Cout, pc.Cout if use_res else 0, pc.Kpad, 1)
    assert L.hands_conv2d_streamk_grid(C.byref(d)) > 0
    ws = torch.zeros(L.hands_conv2d_streamk_workspace_bytes() // 4, dtype=torch.int32, device=DEV)
    got = torch.full((B, Ho, Ho, pc.Cout), float("nan"), device=DEV)
    torch.cuda.synchronize()
    rc = L.hands_conv2d_nhwc_streamk_f32(C.byref(d), ptr(x), ptr(pc.w), ptr(pc.bias), ptr(res) if use_res else None, ptr(got),
                                         ptr(ws), ws.numel() * 4, -7, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("case", [(2, 64, 64, 256, 56, 1), (3, 128, 256, 512, 28, 2), (1, 512, 1024, 2048, 14, 2), (2, 16, 32, 12, 9, 2)])
def test_conv1x1_dual_vs_torch(case):
    """relu(conv1x1(x) + conv1x1_stride(x2) + bias): conv3 + downsample of a bottleneck as one GEMM."""
    from hands_amd.packing import pack_conv
    B, K0, K1, Cout, Ho, s2 = case
    g = torch.Generator().manual_seed(sum(case))
    H2 = Ho * s2 - (s2 - 1) * (case[0] % 2)            # odd input sizes too: (Ho-1)*s2 < H2
    x = torch.randn(B, K0, Ho, Ho, generator=g)
    x2 = torch.randn(B, K1, H2, H2, generator=g)
    w0 = torch.randn(Cout, K0, 1, 1, generator=g) / K0 ** 0.5
    w1 = torch.randn(Cout, K1, 1, 1, generator=g) / K1 ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), w0.double()) + F.conv2d(x2.double(), w1.double(), stride=s2) + bias.double().view(1, -1, 1, 1))
    pc = pack_conv(torch.cat([w0, w1], 1), bias, 1, 0, DEV)
    xd, x2d = _nhwc(x).to(DEV).contiguous(), _nhwc(x2).to(DEV).contiguous()
    out = torch.full((B, Ho, Ho, pc.Cout), float("nan"), device=DEV)
    HandsLight._conv_dual(_lib.lib(), pc, (K0, K1, s2), xd, x2d, B, Ho, Ho, H2, H2, out, _stream())
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2)[:, :Cout]
    err = (got.double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def test_stem_conv_bn_fold_vs_torch():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 3, 224, 224, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * (2 / 147) ** 0.5
    bn = [1 + 0.1 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g),
          0.1 * torch.randn(64, generator=g), 1 + 0.1 * torch.rand(64, generator=g)]
    ref = F.relu(F.batch_norm(F.conv2d(x, w, stride=2, padding=3), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5))
    wf, bf = fold_bn(w, *bn)
    got = _run_conv(x, wf, bf, 2, 3, True, cin_pad_to=4)
    assert (got - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 64, 80), (1, 57, 43), (2, 256, 256)])
@pytest.mark.parametrize("act", [1, 3])
def test_fused_stem_maxpool_is_bit_identical(shape, act):
    """hands_stem_conv_maxpool_nhwc_f32 == stem conv (conv_igemm) followed by the max-pool kernel, bit for
    bit, for full, ragged and odd image sizes (partial pooled tiles, image borders)."""
    from hands_amd._lib import check, ptr
    B, H, W = shape
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * H + W + act)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    bias = torch.randn(64, generator=g)
    pc = pack_conv(w, bias, 2, 3, DEV, cin_pad_to=4)
    x4 = F.pad(_nhwc(torch.randn(B, 3, H, W, generator=g)), (0, 1)).to(DEV).contiguous()
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    a = torch.empty(B, Hc, Wc, 64, device=DEV)
    HandsLight._conv(L, pc, x4, B, H, W, a, act, _stream())
    ref = torch.empty(B, Hp, Wp, 64, device=DEV)
    check(L.hands_maxpool3x3s2_nhwc_f32(ptr(a), ptr(ref), B, Hc, Wc, 64, _stream()), "maxpool")
    got = torch.full((B, Hp, Wp, 64), float("nan"), device=DEV)
    check(L.hands_stem_conv_maxpool_nhwc_f32(ptr(x4), ptr(pc.w), ptr(pc.bias), ptr(got), B, H, W, act, _stream()), "fused stem")
    torch.cuda.synchronize()
    assert torch.equal(got, ref), (got - ref).abs().max().item()
    # and against torch (fp64) for good measure
    t = F.conv2d(F.pad(x4[..., :3].permute(0, 3, 1, 2).double().cpu(), (0, 0)), w.double(), bias.double(), stride=2, padding=3)
    t = F.relu(t) if act == 1 else F.leaky_relu(t, 0.01)
    t = F.max_pool2d(t, 3, 2, 1)
    assert (got.cpu().permute(0, 3, 1, 2).double() - t).abs().max().item() < 2e-5


@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 64, 80), (1, 57, 43), (2, 256, 256)])
@pytest.mark.parametrize("act", [1, 3])
def test_planar_stem_reads_nchw_and_matches(shape, act):
    """hands_stem_conv_maxpool_nchw_f32: the stem straight from the NCHW image, contraction ordered (plane, kh, kw)
    (K = 160 instead of 208).  Against torch fp64 and against the NHWC4 stem (same math, other summation order)."""
    from hands_amd.packing import pack_linear
    B, H, W = shape
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * H + W + act)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    bias = torch.randn(64, generator=g)
    x = torch.randn(B + 1, 3, H, W, generator=g)
    col = [c * 52 + t for c in range(3) for t in range(49)]
    pp = pack_linear(w.reshape(64, 147), bias, DEV, col_index=col, k_total=160)
    assert pp.w.shape == (128, 160) and torch.all(pp.w[:, 49:52] == 0) and torch.all(pp.w[:, 156:] == 0)
    xd = x.to(DEV).contiguous()
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    got = torch.full((B, Hp, Wp, 64), float("nan"), device=DEV)
    check(L.hands_stem_conv_maxpool_nchw_f32(ptr(xd, 3 * H * W), ptr(pp.w), ptr(pp.bias), ptr(got), B, H, W, act, _stream()),
          "planar stem")                                    # image offset 1: the segment form the trunk uses
    pc = pack_conv(w, bias, 2, 3, DEV, cin_pad_to=4)
    x4 = F.pad(_nhwc(x[1:]), (0, 1)).to(DEV).contiguous()
    old = torch.full((B, Hp, Wp, 64), float("nan"), device=DEV)
    check(L.hands_stem_conv_maxpool_nhwc_f32(ptr(x4), ptr(pc.w), ptr(pc.bias), ptr(old), B, H, W, act, _stream()), "nhwc stem")
    torch.cuda.synchronize()
    t = F.conv2d(x[1:].double(), w.double(), bias.double(), stride=2, padding=3)
    t = F.relu(t) if act == 1 else F.leaky_relu(t, 0.01)
    t = F.max_pool2d(t, 3, 2, 1)
    assert (got.cpu().permute(0, 3, 1, 2).double() - t).abs().max().item() < 2e-5
    assert (got - old).abs().max().item() < 2e-5
    assert L.hands_stem_conv_maxpool_nchw_f32(None, ptr(pp.w), ptr(pp.bias), ptr(got), B, H, W, act, _stream()) == 10001


def test_layout_pool_kernels():
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 32, 40, generator=g)
    out = torch.empty(3, 32, 40, 4, device=DEV)
    xd = x.to(DEV)
    check(L.hands_nchw3_to_nhwc4_f32(ptr(xd), ptr(out), 3, 32, 40, _stream()))
    o = out.cpu()
    assert torch.equal(o[..., :3], x.permute(0, 2, 3, 1)) and torch.all(o[..., 3] == 0)

    y = torch.randn(2, 64, 17, 14, generator=g)
    ref = F.max_pool2d(y, 3, 2, 1)
    out = torch.empty(2, ref.shape[2], ref.shape[3], 64, device=DEV)
    yd = _nhwc(y).to(DEV)
    check(L.hands_maxpool3x3s2_nhwc_f32(ptr(yd), ptr(out), 2, 17, 14, 64, _stream()))
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), ref)           # max is exact

    z = torch.randn(3, 2048, 7, 7, generator=g)
    out = torch.empty(3, 2048, device=DEV)
    zd = _nhwc(z).to(DEV)
    check(L.hands_sumpool_nhwc_f32(ptr(zd), ptr(out), 3, 49, 2048, 2048, _stream()))
    ref = z.double().view(3, 2048, -1).sum(2)
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5


def test_kpe_concat_vs_oracle():
    L = _lib.lib()
    g = torch.Generator().manual_seed(6)
    Bg = 3
    crop = torch.randn(2 * Bg, 2048, 7, 7, generator=g)
    glb = torch.randn(Bg, 2048, 7, 7, generator=g)
    ce = 0.5 * torch.randn(2 * Bg, 2, generator=g)
    co = 0.5 * torch.randn(2 * Bg, 8, generator=g)
    ref = O.assemble_features(crop, glb.repeat(2, 1, 1, 1), ce, co)
    out = torch.empty(2 * Bg, 49, 2128, device=DEV)
    dv = [_nhwc(crop).to(DEV), _nhwc(glb).to(DEV), ce.to(DEV), co.to(DEV)]   # keep the buffers alive
    check(L.hands_kpe_concat_f32(ptr(dv[0]), ptr(dv[1]), ptr(dv[2]), ptr(dv[3]), ptr(out), 2 * Bg, Bg, 49,
                                 2048, 4, _stream()))
    got = out.cpu().view(2 * Bg, 7, 7, 2128).permute(0, 3, 1, 2)
    assert torch.equal(got[:, :2048], ref[:, :2048])
    assert (got[:, 2048:] - ref[:, 2048:]).abs().max().item() < 1e-6


def test_rot6d_and_flip_vs_oracle():
    L = _lib.lib()
    g = torch.Generator().manual_seed(8)
    Bg = 5
    d6 = torch.randn(2 * Bg, 96, generator=g)
    rot = torch.empty(2 * Bg, 16, 3, 3, device=DEV)
    d6d = d6.to(DEV)
    check(L.hands_rot6d_to_matrix_f32(ptr(d6d), 96, ptr(rot), 2 * Bg, _stream()))
    ref = O.rotation_6d_to_matrix(d6.view(-1, 6)).view(2 * Bg, 16, 3, 3)
    assert (rot.cpu() - ref).abs().max().item() < 5e-6     # Gram-Schmidt amplifies fp32 rounding

    shape = torch.randn(2 * Bg, 10, generator=g)
    cam = torch.randn(2 * Bg, 3, generator=g)
    cami = torch.randn(2 * Bg, 3, generator=g)
    fl = torch.tensor([0, 1, 1, 0, 1])
    outs = [torch.empty_like(t, device=DEV) for t in (ref, shape, cam, cami)]
    dv = [t.to(DEV) for t in (fl, ref, shape, cam, cami)]                     # keep the buffers alive
    check(L.hands_flip_swap_f32(*[ptr(t) for t in dv], *[ptr(o) for o in outs], Bg, _stream()))
    sgn = torch.tensor([[1.0, -1.0, 1.0]])

    def mirror(p):
        aa = O.matrix_to_axis_angle(p).view(Bg, -1).clone()
        aa[:, 1::3] *= -1
        aa[:, 2::3] *= -1
        return O.axis_angle_to_matrix(aa.view(Bg, 16, 3))

    f1, f3 = fl.bool()[:, None], fl.bool()[:, None, None, None]
    r, l = slice(0, Bg), slice(Bg, 2 * Bg)
    exp_rot = torch.cat([torch.where(f3, mirror(ref[l]), ref[r]), torch.where(f3, mirror(ref[r]), ref[l])])
    exp_shape = torch.cat([torch.where(f1, shape[l], shape[r]), torch.where(f1, shape[r], shape[l])])
    exp_cam = torch.cat([torch.where(f1, cam[l] * sgn, cam[r]), torch.where(f1, cam[r] * sgn, cam[l])])
    exp_cami = torch.cat([torch.where(f1, cami[l] * sgn, cami[r]), torch.where(f1, cami[r] * sgn, cami[l])])
    assert (outs[0].cpu() - exp_rot).abs().max().item() < 2e-6
    assert torch.equal(outs[1].cpu(), exp_shape) and torch.equal(outs[2].cpu(), exp_cam)
    assert torch.equal(outs[3].cpu(), exp_cami)


def test_device_rot_conversions_vs_reference_fixture(golden_dir):
    """The device matrix_to_axis_angle / axis_angle_to_matrix (csrc/rot_device.h, the functions
    mano_pose_kernel and flip_swap_kernel inline) on every rotation of rot_conversions.npz -- 500 random
    + 12 adversarial (theta in {0, 1e-7, ..., pi-1e-4, pi}) + identity -- against the outputs of the
    reference's own common/rot.py:118-193 stored in the fixture.  The quaternion-candidate argmax
    (first maximum wins) is where a port diverges first."""
    L = _lib.lib()
    d = np.load(os.path.join(golden_dir, "rot_conversions.npz"))
    R = torch.from_numpy(d["R"]).to(DEV).contiguous()
    n = R.shape[0]
    aa = torch.full((n, 3), float("nan"), device=DEV)
    check(L.hands_matrix_to_axis_angle_f32(ptr(R), ptr(aa), n, _stream()), "matrix_to_axis_angle")
    got = aa.cpu().numpy()
    ref = d["aa"]
    assert np.isfinite(got).all()
    # theta ~ pi: the axis sign is decided by rounding in both implementations -> compare the ROTATION there
    th = np.linalg.norm(ref, axis=1)
    near_pi = th > np.pi - 1e-3
    assert near_pi.sum() >= 2
    np.testing.assert_allclose(got[~near_pi], ref[~near_pi], rtol=0, atol=2e-6)
    Rg = O.axis_angle_to_matrix(torch.from_numpy(got[near_pi]))
    np.testing.assert_allclose(Rg.numpy(), d["R"][near_pi], rtol=0, atol=2e-6)
    # same candidate as the reference wherever the choice is not a rounding-level tie
    assert (np.abs(got - ref).max(axis=1) < 2e-6).sum() >= n - near_pi.sum()
    # the inverse, on the fixture's own axis-angle inputs
    aa_in = torch.from_numpy(d["aa_in"]).to(DEV).contiguous()
    Ro = torch.empty(aa_in.shape[0], 3, 3, device=DEV)
    check(L.hands_axis_angle_to_matrix_f32(ptr(aa_in), ptr(Ro), aa_in.shape[0], _stream()), "axis_angle_to_matrix")
    np.testing.assert_allclose(Ro.cpu().numpy(), d["R_from_aa"], rtol=0, atol=1e-6)
    assert L.hands_matrix_to_axis_angle_f32(None, ptr(aa), n, _stream()) == 10001


def _run_mano(asset, rotmat, betas, cam, K):
    L = _lib.lib()
    B = rotmat.shape[0]
    mp = pack_mano(asset, DEV)
    consts = _lib.ManoConsts(ptr(mp["pose_mean"]), ptr(mp["J_template"]), ptr(mp["J_shapedirs"]),
                             ptr(mp["lbs_weights"]), ptr(mp["tip_ids"]))
    blend_in = torch.empty(B, 160, device=DEV)
    A = torch.empty(B, 16, 12, device=DEV)
    j16 = torch.empty(B, 16, 3, device=DEV)
    rot_d, betas_d, cam_d, K_d = rotmat.to(DEV), betas.to(DEV), cam.to(DEV), K.to(DEV)
    check(L.hands_mano_pose_f32(C.byref(consts), ptr(rot_d), ptr(betas_d), 10, ptr(blend_in), 160,
                                ptr(A), ptr(j16), B, _stream()))
    vposed = torch.empty(B, 2336, device=DEV)
    HandsLight._conv(L, mp["blend"], blend_in, B, 1, 1, vposed, False, _stream())
    o = {"vertices": torch.empty(B, 778, 3, device=DEV), "joints3d": torch.empty(B, 21, 3, device=DEV),
         "v3d.cam": torch.empty(B, 778, 3, device=DEV), "j3d.cam": torch.empty(B, 21, 3, device=DEV),
         "j2d.norm": torch.empty(B, 21, 2, device=DEV), "cam_t": torch.empty(B, 3, device=DEV)}
    mo = _lib.ManoOut(*[ptr(o[k]) for k in ("vertices", "joints3d", "v3d.cam", "j3d.cam", "j2d.norm", "cam_t")])
    check(L.hands_mano_skin_f32(C.byref(consts), ptr(vposed), 2336, ptr(A), ptr(j16), ptr(cam_d),
                                ptr(K_d), 224.0, 0.1, C.byref(mo), B, _stream()))
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in o.items()}


def _run_mano_fused(assets, rot2, betas2, cam2, K, aa_input=False, n_sides=2):
    """hands_mano_heads_f32: both hands (rows [0,B) right, [B,2B) left) in ONE launch."""
    L = _lib.lib()
    B = K.shape[0]
    mps = [pack_mano(a, DEV) for a in assets]
    keep = [rot2.to(DEV).contiguous(), betas2.to(DEV).contiguous(), cam2.to(DEV).contiguous(), K.to(DEV).contiguous()]
    per = rot2[0].numel()
    sides = (_lib.ManoSide * n_sides)()
    outs = []
    for s_ in range(n_sides):
        mp = mps[s_]
        consts = _lib.ManoConsts(ptr(mp["pose_mean"]), ptr(mp["J_template"]), ptr(mp["J_shapedirs"]),
                                 ptr(mp["lbs_weights"]), ptr(mp["tip_ids"]))
        o = {"vertices": torch.full((B, 778, 3), float("nan"), device=DEV), "joints3d": torch.full((B, 21, 3), float("nan"), device=DEV),
             "v3d.cam": torch.full((B, 778, 3), float("nan"), device=DEV), "j3d.cam": torch.full((B, 21, 3), float("nan"), device=DEV),
             "j2d.norm": torch.full((B, 21, 2), float("nan"), device=DEV), "cam_t": torch.full((B, 3), float("nan"), device=DEV)}
        mo = _lib.ManoOut(*[ptr(o[k]) for k in ("vertices", "joints3d", "v3d.cam", "j3d.cam", "j2d.norm", "cam_t")])
        sides[s_] = _lib.ManoSide(consts, ptr(mp["blend"].w), ptr(mp["blend"].bias), ptr(keep[0], s_ * B * per),
                                  ptr(keep[1], s_ * B * 10), ptr(keep[2], s_ * B * 3), mo)
        outs.append(o)
    check(L.hands_mano_heads_f32(sides, n_sides, ptr(keep[3]), 10, 224.0, 0.1, B, int(aa_input), _stream()), "mano_heads")
    torch.cuda.synchronize()
    return [{k: v.cpu() for k, v in o.items()} for o in outs]


@pytest.mark.parametrize("B", [37, 1, 16, 130])
def test_fused_two_hand_mano_vs_oracle_and_three_launch_chain(B):
    """One launch for both hands (BASELINE configs[4]) against the oracle, the fp64 run of the restatement and
    the three-launch chain; ragged hand counts (B % 16 != 0), identity / near-pi rotations, min_s clamp."""
    assets = [synthetic_mano_asset(True), synthetic_mano_asset(False)]
    rot, betas, cam, K = _mano_inputs(2 * B, 11 + B)
    K = K[:B]
    rot[0] = torch.eye(3)
    if B > 2:
        rot[1, :, :, :] = O.axis_angle_to_matrix(torch.tensor([[3.1415, 0.0, 0.0]]))[0]
        cam[2, 0] = -0.5
        rot[B + 1] = torch.eye(3)
    got = _run_mano_fused(assets, rot, betas, cam, K)
    for s_, asset in enumerate(assets):
        sl = slice(s_ * B, (s_ + 1) * B)
        ref = O.mano_head(rot[sl], betas[sl], cam[sl], K, asset, 224, "")
        v64, j64 = O.mano_lbs(betas[sl], *torch.split(O.matrix_to_axis_angle(rot[sl].double().view(-1, 3, 3)).view(-1, 48), [3, 45], 1),
                              asset, dtype=torch.float64)
        g = got[s_]
        assert all(torch.isfinite(v).all() for v in g.values())
        assert (g["vertices"] - ref["vertices"]).abs().max().item() < 1e-6
        assert (g["joints3d"] - ref["joints3d"]).abs().max().item() < 1e-6
        assert (g["vertices"].double() - v64).abs().max().item() < 1e-6
        assert (g["joints3d"].double() - j64).abs().max().item() < 1e-6
        assert torch.allclose(g["cam_t"], ref["cam_t"], rtol=1e-6, atol=0)
        assert torch.allclose(g["v3d.cam"], ref["v3d.cam"], rtol=2e-6, atol=2e-6)
        assert torch.allclose(g["j3d.cam"], ref["j3d.cam"], rtol=2e-6, atol=2e-6)
        assert (g["j2d.norm"] - ref["j2d.norm"]).abs().max().item() < 1e-5
        assert torch.equal(g["joints3d"][:, 16:], g["vertices"][:, list(O.TIP_IDS)])
        chain = _run_mano(asset, rot[sl], betas[sl], cam[sl], K)
        assert (g["vertices"] - chain["vertices"]).abs().max().item() < 5e-7          # same math, other k order in the blend
        assert torch.equal(g["cam_t"], chain["cam_t"])
    # one side only == the same side of the two-side launch, bit for bit (grid shape never changes a result)
    one = _run_mano_fused(assets[:1], rot[:B], betas[:B], cam[:B], K, n_sides=1)[0]
    for k in one:
        assert torch.equal(one[k], got[0][k]), k


def test_fused_mano_axis_angle_input_and_batch_invariance():
    """axis-angle input (ground-truth MANO parameters, process_arctic.py:16-21) and: hand i of a 1024-hand
    launch equals hand i of a 3-hand launch bit for bit."""
    assets = [synthetic_mano_asset(True), synthetic_mano_asset(False)]
    B = 1024
    rot, betas, cam, K = _mano_inputs(2 * B, 5)
    K = K[:B]
    big = _run_mano_fused(assets, rot, betas, cam, K)
    idx = torch.tensor([0, 1, 2])
    small = _run_mano_fused(assets, torch.cat([rot[idx], rot[B + idx]]), torch.cat([betas[idx], betas[B + idx]]),
                            torch.cat([cam[idx], cam[B + idx]]), K[:3])
    for s_ in range(2):
        for k in small[s_]:
            assert torch.equal(small[s_][k], big[s_][k][:3]), k
    aa = O.matrix_to_axis_angle(rot.view(-1, 3, 3)).view(2 * B, 48)
    via_aa = _run_mano_fused(assets, aa, betas, cam, K, aa_input=True)
    for s_ in range(2):
        assert (via_aa[s_]["vertices"] - big[s_]["vertices"]).abs().max().item() < 1e-6


def _mano_inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    rot = O.rotation_6d_to_matrix(torch.randn(B * 16, 6, generator=g)).view(B, 16, 3, 3)
    betas = torch.randn(B, 10, generator=g)
    cam = torch.tensor([1.0, 0, 0]) + 0.1 * torch.randn(B, 3, generator=g)
    K = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]])[None].repeat(B, 1, 1)
    K[:, 0, 0] += 20 * torch.randn(B, generator=g)
    return rot, betas, cam, K


@pytest.mark.parametrize("is_rhand", [True, False])
def test_mano_head_vs_oracle(is_rhand):
    asset = synthetic_mano_asset(is_rhand)
    rot, betas, cam, K = _mano_inputs(37, 11)
    # adversarial rotations: identity and near-pi
    rot[0] = torch.eye(3)
    rot[1, :, :, :] = O.axis_angle_to_matrix(torch.tensor([[3.1415, 0.0, 0.0]]))[0]
    cam[2, 0] = -0.5      # exercises the min_s clamp
    got = _run_mano(asset, rot, betas, cam, K)
    ref = O.mano_head(rot, betas, cam, K, asset, 224, "")
    v64, j64 = O.mano_lbs(betas, *torch.split(O.matrix_to_axis_angle(rot.double().view(-1, 3, 3)).view(-1, 48), [3, 45], 1),
                          asset, dtype=torch.float64)
    assert (got["vertices"] - ref["vertices"]).abs().max().item() < 1e-6
    assert (got["joints3d"] - ref["joints3d"]).abs().max().item() < 1e-6
    assert (got["vertices"].double() - v64).abs().max().item() < 1e-6      # vs the fp64 run of the restatement
    assert (got["joints3d"].double() - j64).abs().max().item() < 1e-6
    assert torch.allclose(got["cam_t"], ref["cam_t"], rtol=1e-6, atol=0)
    assert torch.allclose(got["v3d.cam"], ref["v3d.cam"], rtol=2e-6, atol=2e-6)
    assert torch.allclose(got["j3d.cam"], ref["j3d.cam"], rtol=2e-6, atol=2e-6)
    assert (got["j2d.norm"] - ref["j2d.norm"]).abs().max().item() < 1e-5


def test_mano_rigid_property_full_size():
    """Config 5 size (1024 hands per side): a global-only rotation must move every vertex rigidly
    about the wrist: verts = R (v_shaped - J0) + J0, when hands_mean = 0."""
    asset = synthetic_mano_asset(True)
    asset.hands_mean[:] = 0
    asset.posedirs[:] = asset.posedirs  # unchanged; pose feature is zero for identity finger joints
    B = 1024
    g = torch.Generator().manual_seed(3)
    rot = torch.eye(3).repeat(B, 16, 1, 1)
    Rg = O.rotation_6d_to_matrix(torch.randn(B, 6, generator=g))
    rot[:, 0] = Rg
    betas = torch.randn(B, 10, generator=g)
    cam = torch.tensor([1.0, 0, 0]).repeat(B, 1)
    K = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]])[None].repeat(B, 1, 1)
    got = _run_mano(asset, rot, betas, cam, K)
    vs = torch.from_numpy(asset.v_template).double() + torch.einsum(
        "bl,mkl->bmk", betas.double(), torch.from_numpy(asset.shapedirs).double())
    J0 = torch.einsum("bik,i->bk", vs, torch.from_numpy(asset.J_regressor[0]).double())
    # R here is the round trip matrix->axis-angle->Rodrigues of Rg; compare against Rg directly
    exp = torch.einsum("bij,bvj->bvi", Rg.double(), vs - J0[:, None]) + J0[:, None]
    assert (got["vertices"].double() - exp).abs().max().item() < 2e-6
    assert (got["joints3d"][:, 16:] - got["vertices"][:, list(O.TIP_IDS)]).abs().max().item() == 0


@pytest.fixture(scope="module")
def gpu_model(recipe_model):
    import copy
    return copy.deepcopy(recipe_model).to(DEV)


@pytest.mark.parametrize("bz,seed", [(2, 0), (2, 1), (2, 2), (1, 0)])
def test_forward_vs_golden(golden_dir, gpu_model, bz, seed):
    """HIP path vs fixtures written by the imported reference; (1, 0) is BASELINE configs[0] (bs=1)."""
    d = np.load(os.path.join(golden_dir, f"hands_light_bz{bz}_seed{seed}.npz"))
    inputs, meta_info = synthetic_inputs(bz, seed, device=DEV)
    meta_info["is_flipped"] = torch.from_numpy(d["is_flipped"]).to(DEV)
    out = gpu_model(inputs, meta_info)
    torch.cuda.synchronize()
    keys = [k[4:] for k in d.files if k.startswith("out/")]
    assert sorted(out.keys()) == sorted(keys) and len(out) == 22
    for k in keys:
        ref, got = d["out/" + k], out[k].cpu().numpy()
        assert got.shape == ref.shape and out[k].is_contiguous() and out[k].device.type == "cuda", k
        if k.startswith("grasp"):
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4, err_msg=k)
        elif ".cam." in k or k.startswith("mano.cam_t."):
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5, err_msg=k)
        elif k.startswith("depth."):                            # eight convolutions behind the trunk features
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), err_msg=k)
        else:
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5, err_msg=k)
    for hn in "rl":
        verr = np.abs(out[f"mano.vertices.{hn}"].cpu().numpy() - d[f"out/mano.vertices.{hn}"]).max()
        assert verr < 1e-6, verr      # north star: fp32 within 1e-3 mm
        mp = O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"].cpu(), torch.from_numpy(d[f"out/mano.joints3d.{hn}"]))
        assert mp < 1e-3, mp          # "MPJPE vs ref" of the north star, mm


def test_forward_with_the_direct_3x3_kernel_vs_golden(golden_dir, gpu_model):
    """engine.winograd = False: the 3x3 / stride-1 layers on the direct implicit GEMM again (the route of rounds 1-2) --
    same golden bar, and within fp32 re-association noise of the default (Winograd) forward."""
    d = np.load(os.path.join(golden_dir, "hands_light_bz2_seed1.npz"))
    inputs, meta = synthetic_inputs(2, 1, device=DEV)
    meta["is_flipped"] = torch.from_numpy(d["is_flipped"]).to(DEV)
    seen = []
    gpu_model.conv_hook = lambda phase, pc, npix, st, has_res, kernel: seen.append(kernel)
    gpu_model.overlap_trunks = False
    try:
        w = {k: v.clone() for k, v in gpu_model(inputs, meta).items()}
        n_wino = seen.count("conv_wino_f32_kernel") + seen.count("conv_wino4_f32_kernel")     # F(2x2) and F(4x4) (winograd4_stages)
        gpu_model.engine.winograd = False
        seen.clear()
        o = {k: v.clone() for k, v in gpu_model(inputs, meta).items()}
        assert n_wino > 0 and n_wino % (2 * 13) == 0 and not any(k.startswith("conv_wino") for k in seen)   # 13 stride-1 3x3 layers per trunk job
    finally:
        gpu_model.engine.winograd = True
        gpu_model.conv_hook = None
        gpu_model.overlap_trunks = True
    torch.cuda.synchronize()
    for hn in "rl":
        for out in (o, w):
            verr = np.abs(out[f"mano.vertices.{hn}"].cpu().numpy() - d[f"out/mano.vertices.{hn}"]).max()
            assert verr < 1e-6, verr
        assert (o[f"mano.vertices.{hn}"] - w[f"mano.vertices.{hn}"]).abs().max().item() < 1e-6


@pytest.mark.parametrize("seed", [0, 2])
def test_forward_latency_mode_vs_golden(golden_dir, gpu_model, seed):
    """Small-batch serving mode (split-K on every layer with few output tiles): same parity bar."""
    d = np.load(os.path.join(golden_dir, f"hands_light_bz2_seed{seed}.npz"))
    inputs, meta_info = synthetic_inputs(2, seed, device=DEV)
    meta_info["is_flipped"] = torch.from_numpy(d["is_flipped"]).to(DEV)
    gpu_model.latency_mode = True
    try:
        out = gpu_model(inputs, meta_info)
        torch.cuda.synchronize()
    finally:
        gpu_model.latency_mode = False
    for hn in "rl":
        verr = np.abs(out[f"mano.vertices.{hn}"].cpu().numpy() - d[f"out/mano.vertices.{hn}"]).max()
        assert verr < 1e-6, verr
        mp = O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"].cpu(), torch.from_numpy(d[f"out/mano.joints3d.{hn}"]))
        assert mp < 1e-3, mp


def test_graphed_forward_is_bit_identical(gpu_model):
    """hipGraph capture of the whole forward (5 streams, fork/join events): replay == eager, bit for bit,
    also on new inputs copied into the captured buffers."""
    from hands_amd import GraphedForward
    inputs, meta_info = synthetic_inputs(2, 0, device=DEV)
    gf = GraphedForward(gpu_model, inputs, meta_info)
    for seed in (0, 7):
        inputs, meta_info = synthetic_inputs(2, seed, device=DEV)
        eager = {k: v.clone() for k, v in gpu_model(inputs, meta_info).items()}
        got = gf(inputs, meta_info)
        torch.cuda.synchronize()
        for k in eager:
            assert torch.equal(got[k], eager[k]), (seed, k)
    bad, meta3 = synthetic_inputs(3, 0, device=DEV)
    with pytest.raises(ValueError):
        gf(bad, meta3)
    # one-stream mode (stream-K would want a zeroed workspace: not under capture) captures and replays too
    import copy
    serial = hands_amd.apply_recipe(hands_amd.HandsLight()).to(DEV).eval()
    serial.overlap_trunks = False
    inputs, meta_info = synthetic_inputs(2, 0, device=DEV)
    eager = {k: v.clone() for k, v in serial(inputs, meta_info).items()}
    gs = GraphedForward(serial, inputs, meta_info)
    got = gs(inputs, meta_info)
    torch.cuda.synchronize()
    for k in eager:
        assert torch.equal(got[k], eager[k]), k


def test_replica_on_a_second_stream(gpu_model):
    """Two forwards in flight on two streams (model + replica sharing the packed weights): same bits as
    the plain call."""
    inputs, meta_info = synthetic_inputs(4, 3, device=DEV)
    ref = {k: v.clone() for k, v in gpu_model(inputs, meta_info).items()}
    rep = gpu_model.replica()
    assert rep._packed is gpu_model._packed and rep._ws is not gpu_model._ws and rep.engine is not gpu_model.engine
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for i in range(4):
        m, st = (gpu_model, s0) if i % 2 == 0 else (rep, s1)
        with torch.cuda.stream(st):
            outs.append(m(inputs, meta_info))
    torch.cuda.synchronize()
    for o in outs:
        for k in ref:
            assert torch.equal(o[k], ref[k]), k


@pytest.mark.parametrize("case", CONV_CASES + [(2, 64, 28, 28, 256, 1, 1, 0, True, True)])
def test_conv_bf16x3_mode_is_fp32_grade(case):
    """HANDS_MATH_BF16X3 (separately reported mode): operands split exactly into three bf16 planes, six bf16 MFMAs per
    k-16 step, fp32 accumulation.  Same error bar against an fp64 convolution as the exact fp32 path."""
    B, Cin, H, W, Cout, k, stride, pad, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, H, W, generator=g) * torch.logspace(-3, 3, Cin).view(1, Cin, 1, 1)   # 6 decades of scale
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5 / torch.logspace(-3, 3, Cin).view(1, Cin, 1, 1)
    bias = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=pad)
    if res is not None:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    eng = ConvEngine()
    eng.math = "bf16x3"
    got = _run_conv(x, w, bias, stride, pad, relu, res, engine=eng)
    exact = _run_conv(x, w, bias, stride, pad, relu, res)
    scale = max(1.0, ref.abs().max().item())
    err, err32 = (got.double() - ref).abs().max().item(), (exact.double() - ref).abs().max().item()
    assert err < 2e-5 * scale and err < 4 * err32 + 1e-7 * scale, (err, err32)


def test_forward_bf16x3_mode_vs_golden(golden_dir, recipe_model):
    """The whole forward in the bf16x3 mode against the reference-generated fixture: the same 1e-6 m / 1e-3 mm bar."""
    import copy
    model = copy.deepcopy(recipe_model).to(DEV)
    model.engine.math = "bf16x3"
    d = np.load(os.path.join(golden_dir, "hands_light_bz2_seed0.npz"))
    inputs, meta_info = synthetic_inputs(2, 0, device=DEV)
    out = model(inputs, meta_info)
    for hn in "rl":
        verr = np.abs(out[f"mano.vertices.{hn}"].cpu().numpy() - d[f"out/mano.vertices.{hn}"]).max()
        assert verr < 1e-6, verr
        assert O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"].cpu(), torch.from_numpy(d[f"out/mano.joints3d.{hn}"])) < 1e-3


def test_async_tail_keeps_stream_semantics(gpu_model):
    """The forward's tail runs on its own stream and the result joins at first use: results are bit-identical
    to the synchronous path; inputs may be overwritten right after forward() returns; several un-consumed
    forwards may be in flight (double-buffered trunk outputs)."""
    from hands_amd.xdict import stream_xdict
    inputs, meta_info = synthetic_inputs(4, 21, device=DEV)
    meta_info["is_flipped"] = torch.tensor([0, 1, 0, 1], device=DEV)
    gpu_model.async_tail = False
    ref = {k: v.clone() for k, v in gpu_model(inputs, meta_info).items()}
    other_in, other_meta = synthetic_inputs(4, 22, device=DEV)
    ref2 = {k: v.clone() for k, v in gpu_model(other_in, other_meta).items()}
    gpu_model.async_tail = True
    torch.cuda.synchronize()
    outs = []
    for i in range(5):                                    # five forwards enqueued back to back, none consumed
        src_in, src_meta = (inputs, meta_info) if i % 2 == 0 else (other_in, other_meta)
        mine = {k: v.clone() for k, v in src_in.items()}
        mm = {k: v.clone() for k, v in src_meta.items()}
        o = gpu_model(mine, mm)
        assert isinstance(o, stream_xdict) and o.is_pending
        for v in list(mine.values()) + list(mm.values()):  # the caller reuses its input buffers immediately
            v.zero_()
        outs.append(o)
    for i, o in enumerate(outs):
        exp = ref if i % 2 == 0 else ref2
        assert len(o) == 22 and not o.is_pending
        for k in exp:
            assert torch.equal(o[k], exp[k]), (i, k)
    # a LARGER batch right after an un-consumed forward makes every workspace grow: the old blocks must not be
    # recycled under the tail that is still reading them
    fresh = hands_amd.apply_recipe(hands_amd.HandsLight()).to(DEV).eval()      # no workspace exists yet
    small = fresh(inputs, meta_info)
    big_in, big_meta = synthetic_inputs(24, 23, device=DEV)
    big = fresh(big_in, big_meta)
    assert small.is_pending and big.is_pending
    for k in ref:
        assert torch.equal(small[k], ref[k]), k
    assert torch.isfinite(big["mano.vertices.r"]).all()


def test_forward_vs_oracle_with_flips(recipe_sd, gpu_model):
    inputs, meta_info = synthetic_inputs(3, 5)
    meta_info["is_flipped"] = torch.tensor([1, 0, 1])
    ref = O.hands_light_forward(recipe_sd, synthetic_mano_asset(True), synthetic_mano_asset(False), inputs, meta_info)
    out = gpu_model({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in meta_info.items()})
    for hn in "rl":
        assert (out[f"mano.vertices.{hn}"].cpu() - ref[f"mano.vertices.{hn}"]).abs().max().item() < 1e-6
        assert (out[f"mano.pose.{hn}"].cpu() - ref[f"mano.pose.{hn}"]).abs().max().item() < 5e-5
        assert O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"].cpu(), ref[f"mano.joints3d.{hn}"]) < 1e-3


def test_full_size_batch_independence(gpu_model):
    """BASELINE config 2 size (bz=256): samples are independent and every kernel's summation order is
    batch-size invariant, so the first two samples of the bz=256 forward must equal the bz=2 forward
    BIT FOR BIT; the forward is also idempotent."""
    inputs, meta_info = synthetic_inputs(256, 0, device=DEV)
    big = gpu_model(inputs, meta_info)
    big = {k: v.clone() for k, v in big.items()}
    small = gpu_model({k: v[:2].contiguous() for k, v in inputs.items()},
                      {k: v[:2].contiguous() for k, v in meta_info.items()})
    for k in small:
        assert torch.equal(big[k][:2], small[k]), k
    again = gpu_model(inputs, meta_info)
    for k in big:
        assert torch.equal(big[k], again[k]), k
        assert torch.isfinite(big[k]).all(), k


def test_wrapper_inference_contract(gpu_model):
    from hands_amd.wrapper import HandsWrapper
    w = HandsWrapper(model=gpu_model)
    inputs, meta_info = synthetic_inputs(2, 1, device=DEV)
    meta_info["imgname"] = ["a.jpg", "b.jpg"]
    out = w.inference(inputs, meta_info)
    assert all(k.startswith(("inputs.", "pred.", "meta_info.")) for k in out)
    assert out["pred.mano.vertices.r"].device.type == "cpu" and out["inputs.img"].device.type == "cpu"
    assert out["meta_info.imgname"] == ["a.jpg", "b.jpg"]
    assert sum(k.startswith("pred.") for k in out) == 22


SPLITK_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, act, residual, S
    (64, 128, 8, 8, 128, 3, 1, 1, 3, False, 4),      # handoccnet hourglass level (LeakyReLU), 32 tiles x 4 slices
    (64, 256, 4, 4, 128, 1, 1, 0, 3, True, 2),       # pointwise + residual
    (256, 2304, 1, 1, 2048, 1, 1, 0, 1, False, 8),   # feature_conv's Linear: 2 x 16 tiles x 8 slices
    (700, 1024, 1, 1, 112, 1, 1, 0, 0, True, 4),     # HMR decoders (N tail: 112 of 128), ragged M, in-place style residual
    (37, 64, 9, 11, 20, 3, 2, 1, 2, False, 3),       # narrow tile, GELU, ragged everything
    (512, 512, 7, 7, 512, 3, 1, 1, 1, False, 2),     # 784 tiles x 2 slices: several workgroups resident per CU
]


@pytest.mark.parametrize("case", SPLITK_CASES)
def test_splitk_n_is_deterministic_and_right(case):
    """hands_conv2d_nhwc_splitk_n_f32 (call-site slice count S; partial sums reduced in slice order by splitk_reduce_kernel):
    reproducible bit for bit over repeated launches and fresh engines, and right against an fp64 convolution."""
    B, Cin, H, W, Cout, k, stride, pad, act, use_res, S = case
    L = _lib.lib()
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), stride, pad, DEV)
    x = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Wo, pc.Cout, generator=g).to(DEV) if use_res else None
    outs = {}
    for rep in (False, True):
        eng = ConvEngine()
        out = torch.full((B, Ho, Wo, pc.Cout), float("nan"), device=DEV)
        for _ in range(3 if rep else 1):
            eng.conv(L, pc, x, B, H, W, out, act, _stream(), res=res, splitk_n=S)
        torch.cuda.synchronize()
        if rep:
            assert torch.equal(out, outs[False])
        outs[rep] = out.clone()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double(), pc.bias[:Cout].double().cpu(), stride=stride, padding=pad)
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2)[:, :Cout].double().cpu()
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref), 3: F.leaky_relu(ref, 0.01)}[act]
    got = outs[True].permute(0, 3, 1, 2)[:, :Cout].double().cpu()
    assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


SUM_BLOCK_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, act, residual, pre, splitk_n
    (8, 512, 7, 7, 512, 3, 1, 1, 1, False, False, 0),      # K = 4608, the 128 x 128 tile, general route (36 / 72 blocks)
    (8, 256, 8, 8, 64, 3, 2, 1, 3, False, False, 0),       # the 256 x 64 tile, stride 2, K = 2304
    (300, 2048, 1, 1, 512, 1, 1, 0, 0, True, False, 0),    # pointwise route + residual, ragged M
    (16, 1024, 4, 4, 112, 1, 1, 0, 3, False, True, 0),     # BatchNorm -> LeakyReLU folded into the operand (the pre entry), N tail
    (64, 2304, 1, 1, 2048, 1, 1, 0, 1, False, False, 3),   # a launch that is split for another reason: every slice is blocked
    (5, 80, 9, 9, 40, 3, 1, 1, 2, False, False, 0),        # K = 720: a partial last block
    (5, 16, 9, 9, 40, 3, 1, 1, 1, False, False, 0),        # K = 144: below two blocks, the single chain stays
]


@pytest.mark.parametrize("case", SUM_BLOCK_CASES)
@pytest.mark.parametrize("limit", [64, 128])
def test_in_kernel_blocked_summation(case, limit):
    """engine.chain_in_kernel (desc.act | HANDS_SUM_BLOCK128 / 64): the k-ordered chain of every output is cut into blocks of
    `limit` floats inside the launch.  Right against an fp64 convolution, no farther from it than the single chain on average,
    bit-reproducible, independent of the batch size, one launch (no split-K workspace pass), and -- where the split-K form of
    the same blocking exists (K / limit <= 32 equal slices) -- within rounding of it."""
    B, Cin, H, W, Cout, k, stride, pad, act, use_res, use_pre, S = case
    L = _lib.lib()
    g = torch.Generator().manual_seed(sum(case) + limit)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), stride, pad, DEV)
    x = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Wo, pc.Cout, generator=g).to(DEV) if use_res else None
    pre = (torch.rand(Cin, generator=g).to(DEV) + 0.5, torch.randn(Cin, generator=g).to(DEV) * 0.1) if use_pre else None
    kernels = []

    def run(blocked, xs=x, rs=res, in_kernel=True):
        eng = ConvEngine()
        eng.winograd = False
        if blocked:
            eng.chain_limit, eng.chain_in_kernel = limit, in_kernel
        eng.hook = lambda phase, pc_, npix, st, has_res, kernel: kernels.append(kernel) if phase == "begin" else None
        out = torch.full((xs.shape[0], Ho, Wo, pc.Cout), float("nan"), device=DEV)
        eng.conv(L, pc, xs, xs.shape[0], H, W, out, act, _stream(), res=rs, splitk_n=S, pre=pre)
        torch.cuda.synchronize()
        return out

    plain, blocked = run(False), run(True)
    applies = pc.Kpad >= 2 * limit
    assert kernels[-1] == ("conv_igemm_splitk_f32_kernel" if S > 1 else "conv_igemm_f32_kernel")     # no extra split
    assert torch.equal(run(True), blocked)
    if not applies:
        assert torch.equal(blocked, plain)
    nb = max(1, B // 3)
    assert torch.equal(run(True, x[:nb].contiguous(), res[:nb].contiguous() if use_res else None), blocked[:nb])
    xin = x.permute(0, 3, 1, 2).double().cpu()
    if use_pre:
        xin = F.leaky_relu(xin * pre[0].double().cpu().view(1, -1, 1, 1) + pre[1].double().cpu().view(1, -1, 1, 1), 0.01)
    ref = F.conv2d(xin, w.double(), pc.bias[:Cout].double().cpu(), stride=stride, padding=pad)
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2)[:, :Cout].double().cpu()
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref), 3: F.leaky_relu(ref, 0.01)}[act]
    err = lambda o: (o.permute(0, 3, 1, 2)[:, :Cout].double().cpu() - ref).abs()
    eb, ep = err(blocked), err(plain)
    assert eb.max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    if applies:
        assert eb.mean().item() <= ep.mean().item() * 1.02, (eb.mean().item(), ep.mean().item())
        if pc.Kpad >= 1024:       # long chains: blocking visibly helps
            assert eb.mean().item() < 0.9 * ep.mean().item(), (eb.mean().item(), ep.mean().item())
    if applies and S <= 1 and pc.Kpad % limit == 0 and pc.Kpad // limit <= 32:
        via_splitk = run(True, in_kernel=False)
        assert kernels[-1] == "conv_igemm_splitk_f32_kernel"
        assert (via_splitk - blocked).abs().max().item() <= 4e-6 * max(1.0, ref.abs().max().item())


FUSED_SPLITK_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, act, residual, S, acc64
    (256, 2160, 1, 1, 1024, 1, 1, 0, 1, False, 8, False),     # an HMR refine layer at bz = 256 (2 x 8 tiles x 8 slices)
    (37, 1024, 1, 1, 112, 1, 1, 0, 0, True, 4, False),        # the stacked decoders: partial m- and n-tile, residual
    (5, 512, 9, 7, 72, 3, 1, 1, 3, True, 5, False),           # a 3x3 layer, odd sizes, S not a power of two
    (300, 256, 1, 1, 64, 1, 1, 0, 2, False, 2, False),        # the narrow (256 x 64) tile, GELU
    (70, 1024, 1, 1, 1024, 1, 1, 0, 3, False, 4, True),       # fp64 accumulation: fp64 partial sums
    (3, 512, 6, 6, 136, 1, 1, 0, 1, True, 3, True),
]


@pytest.mark.parametrize("case", FUSED_SPLITK_CASES)
def test_fused_splitk_reduction_equals_the_reduce_kernel(case):
    """engine.fuse_splitk_reduce (hands_conv2d_nhwc_splitk_fused_f32): the slice of a tile that arrives last adds the partial sums
    in ascending slice order and applies bias / residual / activation.  The same bits as split-K + splitk_reduce_kernel, for every
    arrival order (repeated launches), with the counters left zero -- and an in-place residual (the HMR state row) works."""
    B, Cin, H, W, Cout, k, stride, pad, act, use_res, S, acc64 = case
    L = _lib.lib()
    g = torch.Generator().manual_seed(sum(int(v) for v in case))
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), stride, pad, DEV)
    pc.acc64 = acc64
    x = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Wo, pc.Cout, generator=g).to(DEV) if use_res else None

    def run(fused, eng=None, out=None, rs=res):
        eng = eng or ConvEngine()
        eng.winograd = False
        eng.fuse_splitk_reduce = fused
        out = out if out is not None else torch.full((B, Ho, Wo, pc.Cout), float("nan"), device=DEV)
        eng.conv(L, pc, x, B, H, W, out, act, _stream(), res=rs, splitk_n=S)
        return out, eng

    two, _ = run(False)
    one, eng = run(True)
    torch.cuda.synchronize()
    assert torch.isfinite(one).all() and torch.equal(one, two)
    ctr = eng._counters(L, x.device, _stream())
    assert int(ctr.abs().sum().item()) == 0                       # every tile's counter is back at zero
    for _ in range(5):                                            # other arrival orders, same engine (same counters): same bits
        again, _ = run(True, eng)
        assert torch.equal(again, two)
    assert int(ctr.abs().sum().item()) == 0
    if use_res:                                                   # in place: out aliases the residual
        buf = res.clone()
        inplace, _ = run(True, eng, out=buf, rs=buf)
        torch.cuda.synchronize()
        assert torch.equal(inplace, two)


def test_summation_flags_are_exclusive_and_chain_in_kernel_rejects_other_lengths():
    """desc.act carries at most ONE summation form: both block flags, or a block flag with HANDS_MATH_BF16X3, are an invalid
    descriptor (HANDS_EINVAL = 10001, nothing launched); engine.chain_in_kernel with a block length the kernel has no
    instantiation for raises before anything is launched."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    pc = pack_conv(torch.randn(128, 256, 1, 1, generator=g) / 16, torch.randn(128, generator=g), 1, 0, DEV)
    x = torch.randn(4, 8, 8, 256, generator=g).to(DEV)
    out = torch.full((4, 8, 8, 128), float("nan"), device=DEV)
    for flags in (0x200 | 0x400, 0x200 | 0x100, 0x400 | 0x100, 0x800 | 0x100, 0x800 | 0x200, 0x800 | 0x400, 0x1000):
        d = _lib.ConvDesc(4, 8, 8, 256, 8, 8, 128, 1, 1, 1, 0, 256, 128, 0, pc.Kpad, 1 | flags)
        assert L.hands_conv2d_nhwc_f32(C.byref(d), ptr(x), ptr(pc.w), ptr(pc.bias), None, ptr(out), _stream()) == 10001, hex(flags)
    torch.cuda.synchronize()
    assert torch.isnan(out).all()
    eng = ConvEngine()
    eng.chain_limit, eng.chain_in_kernel = 96, True
    with pytest.raises(ValueError, match="chain_in_kernel"):
        eng.conv(L, pc, x, 4, 8, 8, out, 1, _stream())
    eng.chain_in_kernel = False                    # the split-K form takes any block length
    eng.chain_limit = 128
    eng.conv(L, pc, x, 4, 8, 8, out, 1, _stream())
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and eng.last_sum_block == 128


ACC64_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, act, residual, pre
    (3, 256, 32, 32, 128, 1, 1, 0, 3, False, True),      # a pre-activation unit's conv1 (hand_head.py:170-172)
    (3, 128, 32, 32, 128, 3, 1, 1, 3, False, False),     # its 3x3 (direct, never Winograd)
    (3, 128, 32, 32, 256, 1, 1, 0, 0, True, False),      # its conv3 + skip
    (2, 256, 32, 32, 24, 1, 1, 0, 0, False, False),      # the score layer: 21 + 3 channels, one partial n-tile
    (5, 16, 9, 11, 12, 3, 1, 1, 1, True, False),         # odd sizes, partial m- and n-tiles
    (2, 64, 14, 14, 128, 3, 2, 1, 1, False, False),      # strided
    (70, 1024, 1, 1, 1024, 1, 1, 0, 3, False, False),    # a per-sample MLP row block (mano_head.py:190-207), split-K asked for
    (2, 1024, 7, 7, 512, 3, 1, 0, 1, False, False),      # K = 9216 without padding
]


@pytest.mark.parametrize("case", ACC64_CASES)
def test_fp64_accumulation_is_correctly_rounded(case):
    """PackedConv.acc64 (desc.act | HANDS_ACC_F64): products accumulated by v_mfma_f64_16x16x4_f64, bias added in fp64, ONE rounding
    to fp32, then residual and activation in fp32.  Every output equals the fp32 rounding of an fp64 convolution (the k order only
    matters at 1e-16), it is a direct launch whatever Winograd / blocking the engine would otherwise choose (split-K keeps fp64
    partial sums), and it is bit-reproducible and batch-size invariant."""
    B, Cin, H, W, Cout, k, stride, pad, act, use_res, use_pre = case
    L = _lib.lib()
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    pc = pack_conv(w, torch.randn(Cout, generator=g), stride, pad, DEV)
    pc.acc64 = True
    x = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Wo, pc.Cout, generator=g).to(DEV) if use_res else None
    pre = (torch.rand(Cin, generator=g).to(DEV) + 0.5, torch.randn(Cin, generator=g).to(DEV) * 0.1) if use_pre else None
    kernels = []

    def run(xs=x, rs=res, acc64=True):
        eng = ConvEngine()
        eng.acc64 = acc64
        eng.chain_limit, eng.chain_in_kernel = 64, True              # HandOccNet's engine settings: none may apply to a marked layer
        eng.hook = lambda phase, pc_, npix, st, has_res, kernel: kernels.append((kernel, eng.last_acc64)) if phase == "begin" else None
        out = torch.full((xs.shape[0], Ho, Wo, pc.Cout), float("nan"), device=DEV)
        eng.conv(L, pc, xs, xs.shape[0], H, W, out, act, _stream(), res=rs, splitk=(H * W == 1), pre=pre)
        torch.cuda.synchronize()
        return out

    out = run()
    # a per-sample row block is cut by the library's layer-only split-K policy: its partial sums stay fp64
    assert kernels[-1] == ("conv_igemm_splitk_f32_kernel" if H * W == 1 else "conv_igemm_f32_kernel", True)
    assert torch.equal(run(), out)
    nb = max(1, B // 3)
    assert torch.equal(run(x[:nb].contiguous(), res[:nb].contiguous() if use_res else None), out[:nb])
    xin = x.cpu()
    if use_pre:
        xin = F.leaky_relu(xin * pre[0].cpu() + pre[1].cpu(), 0.01)          # fp32, as the staging path computes it
    ref = F.conv2d(xin.permute(0, 3, 1, 2).double(), w.double(), pc.bias[:Cout].double().cpu(), stride=stride, padding=pad).float()
    if res is not None:
        ref = ref + res.cpu().permute(0, 3, 1, 2)[:, :Cout]
    ref = {0: ref, 1: F.relu(ref), 3: F.leaky_relu(ref, 0.01)}[act]
    got = out.cpu().permute(0, 3, 1, 2)[:, :Cout]
    assert torch.isfinite(got).all()
    mism = (got != ref)
    # an fp64 sum in another order differs at 1e-16 relative: a different fp32 rounding needs a tie within that distance
    assert mism.float().mean().item() < 1e-5, mism.float().mean().item()
    assert (got - ref).abs().max().item() <= 2.4e-7 * max(1.0, ref.abs().max().item())
    plain = run(acc64=False).cpu().permute(0, 3, 1, 2)[:, :Cout]            # the engine switch turns the marks off
    assert kernels[-1][1] is False
    e64 = F.conv2d(xin.permute(0, 3, 1, 2).double(), w.double(), pc.bias[:Cout].double().cpu(), stride=stride, padding=pad)
    if res is not None:
        e64 = e64 + res.cpu().permute(0, 3, 1, 2)[:, :Cout].double()
    e64 = {0: e64, 1: F.relu(e64), 3: F.leaky_relu(e64, 0.01)}[act]
    assert (got.double() - e64).abs().mean().item() <= (plain.double() - e64).abs().mean().item()


@pytest.mark.parametrize("name", SWITCH_CASES)
def test_switch_configurations_vs_reference_fixtures(golden_dir, name):
    """Non-default HandsLight switches through the HIP path against what the REFERENCE produced for them
    (tests/golden/make_golden_switches.py): `no_crops` (arctic_light: hands_avgpool_nhwc_f32 -> both heads), the image-level
    encodings (hands_image_posenc_nhwc_f32 -> widened conv1 on the general implicit-GEMM route + max-pool), pos_enc None,
    'sinusoidal_cc', the grasp head without the global feature vector / absent, the per-pixel encodings 'dense' / 'dense_latent' /
    'cam_conv' (hands_dense_posenc_f32, hands_concat_nhwc_f32), the rotation corrections 'pcl' / 'perspective_correction' (with and
    without a flipped sample) and the depth head (hands_upsample_bilinear_ac_f32).  Same bar as the default configuration."""
    from switch_cases import load_case
    d, cfg, args, inputs, meta_info = load_case(golden_dir, name)
    model = hands_amd.apply_recipe(hands_amd.HandsLight(args=args)).eval().to(DEV)
    dev = lambda t: {k: v.to(DEV) for k, v in t.items()}
    out = model(dev(inputs), dev(meta_info))
    torch.cuda.synchronize()
    keys = [k[4:] for k in d.files if k.startswith("out/")]
    assert sorted(out.keys()) == sorted(keys)
    for k in keys:
        ref, got = d["out/" + k], out[k].cpu().numpy()
        assert got.shape == ref.shape, k
        if k.startswith(("grasp", "center.", "corner.")):       # MLP read-outs of O(1-10) activations
            np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4, err_msg=k)
        elif ".cam." in k or k.startswith("mano.cam_t."):
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5, err_msg=k)
        elif k.startswith("depth."):                            # eight convolutions behind the trunk features
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), err_msg=k)
        else:
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5, err_msg=k)
    for hn in "rl":
        verr = np.abs(out[f"mano.vertices.{hn}"].cpu().numpy() - d[f"out/mano.vertices.{hn}"]).max()
        mp = O.mpjpe_ra_mm(out[f"mano.joints3d.{hn}"].cpu(), torch.from_numpy(d[f"out/mano.joints3d.{hn}"]))
        assert verr < 1e-6 and mp < 1e-3, (name, hn, verr, mp)
    # batch independence holds on these routes too: sample 1 alone == sample 1 of the pair
    one = model({k: v[1:].contiguous() for k, v in dev(inputs).items()}, {k: v[1:].contiguous() for k, v in dev(meta_info).items()})
    for k in keys:
        assert torch.equal(one[k], out[k][1:]), k


def test_depth_head_passes_are_batch_independent():
    """use_depth_loss at 130 samples: the depth head walks its 260 crops in passes of 256 (the 224 x 224 x 32 map of one pass is
    1.6 GB); samples of the first and of the second pass equal the same samples run as a pair, bit for bit."""
    args = type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS, use_depth_loss=True))
    model = hands_amd.apply_recipe(hands_amd.HandsLight(args=args)).eval().to(DEV)
    inputs, meta = synthetic_inputs(130, 9)
    dev = lambda t: {k: v.to(DEV) for k, v in t.items()}
    inputs, meta = dev(inputs), dev(meta)
    big = model(inputs, meta)
    big = {k: v.clone() for k, v in big.items()}
    assert big["depth.r"].shape == (130, 224, 224) and torch.isfinite(big["depth.l"]).all()
    for lo in (0, 128):
        cut = lambda t: {k: v[lo:lo + 2].contiguous() for k, v in t.items()}
        small = model(cut(inputs), cut(meta))
        for k in ("depth.r", "depth.l", "mano.vertices.r", "mano.vertices.l", "grasp.l"):
            assert torch.equal(small[k], big[k][lo:lo + 2]), (k, lo)


@pytest.mark.parametrize("over", [dict(pos_enc="dense_latent"), dict(pos_enc="pcl"), dict(use_depth_loss=True)])
def test_graphed_forward_of_non_default_configurations(over):
    """hipGraph replay of configurations with extra inputs (per-pixel maps, the pcl rotations) and extra kernels (the depth head):
    the replay on new inputs equals the eager forward bit for bit."""
    from hands_amd import GraphedForward
    from hands_amd.weights import synthetic_dense_inputs
    args = type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS, **over))
    model = hands_amd.apply_recipe(hands_amd.HandsLight(args=args)).eval().to(DEV)

    def batch(seed):
        inputs, meta = synthetic_inputs(2, seed)
        if over.get("pos_enc") in ("dense_latent", "pcl"):
            inputs.update(synthetic_dense_inputs(2, seed, over["pos_enc"]))
        meta["is_flipped"] = torch.tensor([seed & 1, 0])
        return {k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in meta.items()}

    gf = GraphedForward(model, *batch(0))
    for seed in (1, 2):
        inputs, meta = batch(seed)
        want = {k: v.clone() for k, v in model(inputs, meta).items()}
        got = gf(inputs, meta)
        torch.cuda.synchronize()
        assert sorted(got.keys()) == sorted(want.keys())
        for k in want:
            assert torch.equal(got[k], want[k]), (over, seed, k)
