"""Per-model launch state of the GEMM / convolution kernel (``hands_conv2d_nhwc_f32`` and friends).

Every model (``HandsLight``, ``HAMER``, ``HandOccNet``) owns one :class:`ConvEngine`: the switches that
used to be class attributes (split-K policy, small-batch ``latency_mode``, multi-stream overlap, the
stem / downsample fusions, the profiling hook) are per-instance state, so two models in one process can
differ and the C library underneath keeps its "no global mutable state" promise all the way up.
``DEFAULT_ENGINE`` serves the few call sites that have no model (tests and tools driving a single layer,
the wrapper's ground-truth MANO pass).
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import ConvDesc, ConvJob, check, ptr

MATH_BF16X3 = 0x100   # HANDS_MATH_BF16X3 (include/hands_hip.h)
_SUM_BLOCK = {128: 0x200, 64: 0x400}   # HANDS_SUM_BLOCK128 / HANDS_SUM_BLOCK64
ACC_F64 = 0x800                        # HANDS_ACC_F64


class ConvEngine:
    def __init__(self):
        self.use_splitk = True        # deterministic split-K for the latency-bound per-sample head GEMMs
        self.latency_mode = False     # opt-in small-batch serving: split-K on every launch with <= 128 output
                                      # tiles (results then depend on the batch size at the 1e-7 level; the
                                      # default keeps every output bit independent of the batch size)
        self.overlap = True           # run independent jobs (trunks, heads, crop chunks) on side HIP streams
        self.fuse_stem_pool = True    # stem conv + BN + act + max-pool as one kernel (csrc/stem_pool.hip)
        self.fuse_downsample = True   # first block of a stage: conv3 + downsample + add + ReLU as one two-source GEMM
        self.stream_k = "auto"        # persistent stream-K launches where the tile count quantises badly (bit-identical
                                      # to the plain launch).  "auto": only when the launch has the chip to itself
                                      # (overlap off: +5 % measured); with several streams in flight the other
                                      # streams' workgroups already fill the tail and persistent workgroups would
                                      # only hold their slots (-1.7 % measured).  True / False force it.
        self.math = "fp32"            # "fp32" = exact fp32 MFMA (the parity path, the default).  "bf16x3" = separately
                                      # reported mode: operands split into three exact bf16 planes, six bf16 MFMAs per
                                      # k-16 step with fp32 accumulation (HANDS_MATH_BF16X3); never the headline
        self.winograd = True          # 3x3 / stride 1 / pad 1 layers as Winograd F(2x2,3x3) on the fp32 matrix cores
                                      # (hands_conv3x3_winograd_f32: 2.25x fewer multiplications; fp32 throughout, results
                                      # differ from the direct kernel by fp32 rounding).  False = the direct implicit GEMM
        self.winograd4 = False        # the same layers as Winograd F(4x4,3x3) (hands_conv3x3_winograd4_f32: 2.25 multiplications per
                                      # output instead of 4) where the layer was packed for it (PackedConv.wino4) and the library
                                      # takes the shape; HandsLight turns it on, HandOccNet keeps F(2x2) (DESIGN.md "Conditioning note")
        self.fuse_mano = True         # MANOHead.forward of both hands as one launch (hands_mano_heads_f32)
        self.fuse_pre = True          # handoccnet_light's pre-activation units: BatchNorm -> LeakyReLU folded into the operand
                                      # staging of the unit's first (pointwise) convolution (hands_conv2d_nhwc_pre_f32)
        self.chain_limit = 0          # > 0: blocked fp32 summation.  In a direct (non-Winograd) launch whose contraction is at least
                                      # chain_min_k floats long no fp32 FMA chain is longer than chain_limit floats: block sums are
                                      # added in block order -- inside the launch (chain_in_kernel, below) or by cutting K into
                                      # S = Kpad / chain_limit slices of the deterministic split-K form (a workspace round trip per
                                      # such launch).  Fewer roundings per output; a fixed function of the layer -- batch-size
                                      # independent.  HandOccNet sets it (DESIGN.md "Conditioning note"); 0 elsewhere
        self.chain_min_k = 0          # launches with Kpad below max(chain_min_k, 2 * chain_limit) keep their single chain
        self.chain_skip_tokens = False  # True: token GEMMs (H = W = 1 with >= 4096 rows: the transformer MLPs) keep their single chain
        self.chain_max_pix = 0        # > 0: only launches on maps of at most this many output pixels per image are blocked (the
                                      # workspace round trip is per output element: large maps pay most for the same chain)
        self.chain_in_kernel = False  # True (chain_limit 128 or 64 only): the blocks are summed INSIDE the launch (desc.act |=
                                      # HANDS_SUM_BLOCK128 / 64: a second accumulator set, no workspace, no reduce launch) instead
                                      # of through split-K; launches that are split for another reason block each slice
        self.acc64 = True             # honour PackedConv.acc64: layers a model marked run with fp64 accumulation (HANDS_ACC_F64:
                                      # v_mfma_f64_16x16x4_f64, correctly rounded fp32 outputs, half the fp32 matrix rate) -- ONE plain
                                      # direct launch, never Winograd / split-K / stream-K / blocked.  HandOccNet marks its heat-map head,
                                      # encoder and MLPs (DESIGN.md "Conditioning note"); False = those layers take the fp32 routes
        self.fuse_splitk_reduce = True  # split-K launches reduce inside the launch (the last slice of a tile to arrive adds the partial
                                      # sums in slice order: hands_conv2d_nhwc_splitk_fused_f32, same bits); False = the reduce kernel
        self.group_launches = True    # conv_group(): independent pointwise layers of one kernel instantiation as ONE launch
                                      # (hands_conv2d_group_f32, bit-identical to the separate launches); False = one launch each
        self.last_acc64 = False       # the launch the hook is being called for accumulates in fp64
        self.hook = None              # callable(phase, pc, npix, stream_handle, has_res, kernel): bench.py brackets
                                      # every MFMA launch (conv_igemm and the fused stem) with events; `kernel` is the
                                      # kernel the launch really runs (conv_igemm_f32_kernel / conv_igemm_sk_f32_kernel =
                                      # stream-K / conv_igemm_splitk_f32_kernel = split-K + reduce / stem_pool_*)
        self.last_wino_macs = 0       # executed (not algorithmic) MACs of the Winograd launch the hook is being called for
        self.last_sum_block = 0       # summation block (floats) of the direct launch the hook is being called for: chain_limit when
                                      # the launch is blocked (either form), 0 for a single chain
        self._splitk_ws = {}          # (device, stream handle) -> workspace tensor
        self._sk_ws = {}              # (device, stream handle) -> [zeroed stream-K workspace, epoch counter]
        self._capture_ws = {}         # split-K workspaces of launches recorded into the hipGraph being captured (graph memory
                                      # pool): the table of the CURRENT capture, see begin_capture()
        self._capture_tables = {}     # capture token -> its table (kept alive as long as the captured graph may replay)
        self._capture_seq = 0

    def clone_settings(self) -> "ConvEngine":
        e = ConvEngine()
        for k in ("use_splitk", "latency_mode", "overlap", "fuse_stem_pool", "fuse_downsample", "winograd", "winograd4", "fuse_mano", "fuse_pre", "stream_k", "math", "chain_limit",
                  "chain_min_k", "chain_max_pix", "chain_skip_tokens", "chain_in_kernel", "acc64", "group_launches", "fuse_splitk_reduce"):
            setattr(e, k, getattr(self, k))
        return e

    def begin_capture(self) -> int:
        """A hipGraph capture is about to record launches of this engine: give it a workspace table of its own and return
        its token.  Other live captures of the same (possibly shared) engine keep theirs."""
        self._capture_seq += 1
        self._capture_ws = self._capture_tables[self._capture_seq] = {}
        return self._capture_seq

    def drop_capture(self, token: int):
        """The graph captured under ``token`` will not be replayed again."""
        t = self._capture_tables.pop(token, None)
        if t is not None and t is self._capture_ws:
            self._capture_ws = {}

    def release_workspaces(self, dev=None):
        """Drop the per-stream split-K / stream-K workspaces (all devices, or one).  They are keyed by the raw
        stream handle the launch went to: a model calls this when it drops its side streams (``invalidate_packed``,
        ``.to()``), so a later stream that happens to get a recycled handle starts from a fresh, zero-filled
        workspace and no destroyed stream keeps 64 MB pinned."""
        for table in (self._splitk_ws, self._sk_ws, self._capture_ws, *self._capture_tables.values()):
            for key in [k for k in table if dev is None or k[0] == dev]:
                del table[key]

    @staticmethod
    def _capturing(L, stream):
        """hipGraph capture status of the stream the launch GOES TO (not of torch's current stream)."""
        st = L.hands_stream_is_capturing(stream)
        if st < 0:
            raise RuntimeError(f"hands_amd: hipStreamIsCapturing failed: {L.hands_error_string(-st).decode()}")
        return bool(st)

    def _workspace(self, L, dev, stream, need):
        key = (dev, stream)
        capturing = self._capturing(L, stream)
        # a launch recorded into a hipGraph gets a workspace from the graph's own memory pool (torch allocates from it
        # during capture) and never shares one with eager launches: the capture stream's handle can be recycled
        table = self._capture_ws if capturing else self._splitk_ws
        ws = table.get(key)
        if ws is None or ws.numel() < need:
            if ws is not None and not capturing:
                # a kernel on a raw side-stream handle may still read the old block, and the caching
                # allocator only knows torch's current stream: drain before dropping it (growth is rare --
                # the first forward at a new batch size).  (Under capture a larger block is simply recorded as a new
                # allocation of the graph pool; synchronising would be illegal there.)
                torch.cuda.synchronize(dev)
            ws = table[key] = torch.empty(max(need, 1 << 22), dtype=torch.float32, device=dev)
        return ws

    def _counters(self, L, dev, stream):
        """The zeroed per-tile arrival counters of the fused split-K reduction: one array per (device, launch stream) -- launches
        on different streams must not share it; every launch leaves it zero.  None under hipGraph capture (the zero fill of a new
        array would have to be ordered before a launch on another stream inside the capture: those launches keep the reduce kernel)."""
        if self._capturing(L, stream):
            return None
        key = (dev, stream, "ctr")
        c = self._splitk_ws.get(key)
        if c is None:
            c = self._splitk_ws[key] = torch.zeros(16384, dtype=torch.int32, device=dev)
            torch.cuda.current_stream(dev).synchronize()     # zero fill done before a side stream uses it
        return c

    def conv(self, L, pc, x, B, H, W, out, relu, stream, res=None, in_ps=None, out_ps=None, res_ps=None,
             x_off=0, out_off=0, res_off=0, splitk=False, splitk_n=0, pre=None):
        """One convolution / linear layer.  ``splitk=True`` marks rows that are per-SAMPLE (head MLPs): only
        there may the library cut K by its own (layer-only) policy -- token / pixel GEMMs would cross the
        library's row threshold between batch sizes and lose bit-reproducibility.  ``splitk_n`` is a
        call-site constant slice count (summation order independent of the batch size)."""
        Ho = (H + 2 * pc.pad - pc.KH) // pc.stride + 1
        Wo = (W + 2 * pc.pad - pc.KW) // pc.stride + 1
        d = ConvDesc(B, H, W, pc.Cin, Ho, Wo, pc.Cout, pc.KH, pc.KW, pc.stride, pc.pad,
                     in_ps or pc.Cin, out_ps or pc.Cout,
                     (pc.Cout if res_ps is None else res_ps) if res is not None else 0,
                     pc.Kpad, int(relu) | (MATH_BF16X3 if self.math == "bf16x3" else 0))   # relu: bool or a HANDS_ACT_* code
        hook = self.hook
        self.last_sum_block = 0
        acc64 = self.last_acc64 = bool(self.acc64 and pc.acc64 and self.math == "fp32" and pc.Cin != 4)
        S = L.hands_conv2d_splitk_factor(C.byref(d)) if (splitk and self.use_splitk) else 1
        if splitk_n > 1 and self.use_splitk:
            S = splitk_n
        if self.latency_mode:
            # small-batch serving: a layer with a handful of output tiles walks a K of 2304-4608 serially
            # on a few CUs; cut K so that ~256 workgroups exist, at least 8 k-steps (128 floats) per slice
            bm, bn = (256, 64) if pc.Cout <= 64 else (128, 128)
            tiles = -(-(B * Ho * Wo) // bm) * -(-pc.Cout // bn)
            S = max(S, min(256 // tiles, pc.Kpad // 128, 32)) if tiles <= 128 else S
        if acc64:               # direct launch(es); a split keeps fp64 partial sums (per-sample head GEMMs, call-site constants)
            d.act |= ACC_F64
            S = (L.hands_conv2d_splitk_factor(C.byref(d)) if (splitk and self.use_splitk) else 1)
            if splitk_n > 1 and self.use_splitk:
                S = splitk_n
        rp = ptr(res, res_off) if res is not None else None
        if (not acc64 and self.winograd and self.winograd4 and pc.wino4 is not None and res is None and pre is None and S <= 1
                and self.math == "fp32" and (ptr(x, x_off) | ptr(out, out_off)) % 16 == 0
                and L.hands_conv3x3_winograd4_supported(C.byref(d))):
            if hook is not None:
                self.last_wino_macs = L.hands_conv3x3_winograd4_executed_macs(C.byref(d))
                hook("begin", pc, B * Ho * Wo, stream, False, "conv_wino4_f32_kernel")
            check(L.hands_conv3x3_winograd4_f32(C.byref(d), ptr(x, x_off), ptr(pc.wino4), ptr(pc.bias), ptr(out, out_off), stream),
                  "hands_conv3x3_winograd4_f32")
            if hook is not None:
                hook("end", pc, B * Ho * Wo, stream, False, "conv_wino4_f32_kernel")
            return Ho, Wo
        if (not acc64 and self.winograd and pc.wino is not None and res is None and pre is None and S <= 1 and self.math == "fp32"
                and (ptr(x, x_off) | ptr(out, out_off)) % 16 == 0        # its accesses are 16 bytes wide
                and L.hands_conv3x3_winograd_supported(C.byref(d))):
            if hook is not None:
                self.last_wino_macs = L.hands_conv3x3_winograd_executed_macs(C.byref(d))   # what the matrix cores execute
                hook("begin", pc, B * Ho * Wo, stream, False, "conv_wino_f32_kernel")
            check(L.hands_conv3x3_winograd_f32(C.byref(d), ptr(x, x_off), ptr(pc.wino), ptr(pc.bias), ptr(out, out_off), stream),
                  "hands_conv3x3_winograd_f32")
            if hook is not None:
                hook("end", pc, B * Ho * Wo, stream, False, "conv_wino_f32_kernel")
            return Ho, Wo
        limit = self.chain_limit if pc.sum_block < 0 else (pc.sum_block if self.chain_limit else 0)   # per-layer override of the block
        if (limit and not acc64 and self.math == "fp32" and (self.chain_in_kernel or self.use_splitk)
                and pc.Kpad >= max(2 * limit, self.chain_min_k)
                and (not self.chain_max_pix or Ho * Wo <= self.chain_max_pix)
                and not (self.chain_skip_tokens and H * W == 1 and B >= 4096)):
            # blocked summation (not the Winograd launches above: their chains are Cin long)
            self.last_sum_block = limit
            if self.chain_in_kernel:
                if limit not in _SUM_BLOCK:
                    raise ValueError(f"hands_amd: chain_in_kernel takes chain_limit 64 or 128, not {limit}")
                d.act |= _SUM_BLOCK[limit]
            else:
                S = max(S, min(pc.Kpad // limit, 32))
        if pre is not None:
            # pointwise layer behind an eval BatchNorm -> LeakyReLU (pre = (scale, shift) device vectors): the affine +
            # activation is applied to the operand on its way into LDS (hands_conv2d_nhwc_pre_f32)
            kname = "conv_igemm_splitk_f32_kernel" if S > 1 else "conv_igemm_f32_kernel"
            if hook is not None:
                hook("begin", pc, B * Ho * Wo, stream, res is not None, kname)
            ws = self._workspace(L, x.device, stream, L.hands_conv2d_workspace_floats(C.byref(d), S)) if S > 1 else None
            check(L.hands_conv2d_nhwc_pre_f32(C.byref(d), ptr(x, x_off), ptr(pre[0]), ptr(pre[1]), ptr(pc.w), ptr(pc.bias), rp,
                                              ptr(out, out_off), S, ptr(ws), ws.numel() if ws is not None else 0, stream),
                  "hands_conv2d_nhwc_pre_f32")
            if hook is not None:
                hook("end", pc, B * Ho * Wo, stream, res is not None, kname)
            return Ho, Wo
        use_sk = S <= 1 and not acc64 and self.math == "fp32" and ((not self.overlap) if self.stream_k == "auto" else self.stream_k) and \
            not (d.act & (_SUM_BLOCK[64] | _SUM_BLOCK[128])) and \
            L.hands_conv2d_streamk_grid(C.byref(d)) > 0 and not self._capturing(L, stream)
        # (a blocked launch keeps the plain kernel: stream-K continues ONE chain across workgroups)
        # (not under hipGraph capture: the zero-filled workspace of a new stream cannot be set up inside one)
        kname = "conv_igemm_splitk_f32_kernel" if S > 1 else ("conv_igemm_sk_f32_kernel" if use_sk else "conv_igemm_f32_kernel")
        if hook is not None:
            hook("begin", pc, B * Ho * Wo, stream, res is not None, kname)
        if S > 1:     # latency-bound GEMM: deterministic split-K with a per-stream workspace
            ws = self._workspace(L, x.device, stream, L.hands_conv2d_workspace_floats(C.byref(d), S))
            ctr = self._counters(L, x.device, stream) if self.fuse_splitk_reduce else None
            if ctr is not None:       # the last slice of a tile reduces it: no second launch (same bits)
                check(L.hands_conv2d_nhwc_splitk_fused_f32(C.byref(d), ptr(x, x_off), ptr(pc.w), ptr(pc.bias), rp, ptr(out, out_off),
                                                           S, ptr(ws), ws.numel(), ptr(ctr), ctr.numel(), stream),
                      "hands_conv2d_nhwc_splitk_fused_f32")
            else:
                check(L.hands_conv2d_nhwc_splitk_n_f32(C.byref(d), ptr(x, x_off), ptr(pc.w), ptr(pc.bias), rp, ptr(out, out_off),
                                                       S, ptr(ws), ws.numel(), stream), "hands_conv2d_nhwc_splitk_n_f32")
        elif use_sk:
            key = (x.device, stream)
            sk = self._sk_ws.get(key)
            if sk is None:
                nbytes = L.hands_conv2d_streamk_workspace_bytes()
                sk = self._sk_ws[key] = [torch.zeros(nbytes // 4, dtype=torch.int32, device=x.device), 0]
                torch.cuda.current_stream(x.device).synchronize()     # zero-fill done before a side stream uses it
            sk[1] = sk[1] % 0x7FFFFFF0 + 1
            check(L.hands_conv2d_nhwc_streamk_f32(C.byref(d), ptr(x, x_off), ptr(pc.w), ptr(pc.bias), rp, ptr(out, out_off),
                                                  ptr(sk[0]), sk[0].numel() * 4, sk[1], stream),
                  "hands_conv2d_nhwc_streamk_f32")
        else:
            check(L.hands_conv2d_nhwc_f32(C.byref(d), ptr(x, x_off), ptr(pc.w), ptr(pc.bias), rp, ptr(out, out_off),
                                          stream), "hands_conv2d_nhwc_f32")
        if hook is not None:
            hook("end", pc, B * Ho * Wo, stream, res is not None, kname)
        return Ho, Wo

    def conv_group(self, L, jobs, stream):
        """Independent layers that may run concurrently: ``jobs`` is a list of dicts with the arguments of :meth:`conv` (``pc, x, B,
        H, W, out, relu`` + optional ``res, pre, in_ps, out_ps, res_ps, x_off, out_off, res_off``).  Pointwise jobs that select the
        same kernel instantiation (hands_conv2d_group_class) go out as ONE launch of up to 8 (hands_conv2d_group_f32: every tile is
        computed exactly as its own launch would, same bits); everything else -- 3x3 layers, fp64 or split-K layers, a class with a
        single member -- takes :meth:`conv`.  The outputs must not alias each other or any input of the group."""
        single = lambda j: self.conv(L, j["pc"], j["x"], j["B"], j["H"], j["W"], j["out"], j.get("relu", 0), stream,
                                     **{k: j[k] for k in ("res", "pre", "in_ps", "out_ps", "res_ps", "x_off", "out_off", "res_off") if k in j})
        if len(jobs) < 2 or not self.group_launches or self.latency_mode or self.math != "fp32":
            for j in jobs:
                single(j)
            return
        classes = {}
        for j in jobs:
            pc, B, H, W = j["pc"], j["B"], j["H"], j["W"]
            res, pre = j.get("res"), j.get("pre")
            Ho = (H + 2 * pc.pad - pc.KH) // pc.stride + 1
            Wo = (W + 2 * pc.pad - pc.KW) // pc.stride + 1
            d = ConvDesc(B, H, W, pc.Cin, Ho, Wo, pc.Cout, pc.KH, pc.KW, pc.stride, pc.pad, j.get("in_ps") or pc.Cin,
                         j.get("out_ps") or pc.Cout, (pc.Cout if j.get("res_ps") is None else j["res_ps"]) if res is not None else 0,
                         pc.Kpad, int(j.get("relu", 0)))
            cls = -1
            if not (self.acc64 and pc.acc64 and pc.Cin != 4):
                limit = self.chain_limit if pc.sum_block < 0 else (pc.sum_block if self.chain_limit else 0)
                blocked = bool(limit and self.chain_in_kernel and pc.Kpad >= max(2 * limit, self.chain_min_k)
                               and (not self.chain_max_pix or Ho * Wo <= self.chain_max_pix)
                               and not (self.chain_skip_tokens and H * W == 1 and B >= 4096))
                if limit and not self.chain_in_kernel:
                    blocked = None                     # the split-K form of the blocks: not a grouped launch
                if blocked is not None:
                    if blocked:
                        if limit not in _SUM_BLOCK:
                            raise ValueError(f"hands_amd: chain_in_kernel takes chain_limit 64 or 128, not {limit}")
                        d.act |= _SUM_BLOCK[limit]
                    cls = L.hands_conv2d_group_class(C.byref(d), 1 if pre is not None else 0)
                    j["_blocked"] = limit if blocked else 0
            j["_d"], j["_npix"] = d, B * Ho * Wo
            classes.setdefault(cls, []).append(j)
        hook = self.hook
        for cls, members in classes.items():
            if cls < 0 or len(members) < 2:
                for j in members:
                    single(j)
                continue
            for i in range(0, len(members), 8):
                chunk = members[i:i + 8]
                if len(chunk) == 1:
                    single(chunk[0])
                    continue
                arr = (ConvJob * len(chunk))()
                for k, j in enumerate(chunk):
                    pc, res, pre = j["pc"], j.get("res"), j.get("pre")
                    arr[k] = ConvJob(C.pointer(j["_d"]), ptr(j["x"], j.get("x_off", 0)), ptr(pc.w), ptr(pc.bias),
                                     ptr(res, j.get("res_off", 0)) if res is not None else None, ptr(j["out"], j.get("out_off", 0)),
                                     ptr(pre[0]) if pre is not None else None, ptr(pre[1]) if pre is not None else None)
                self.last_sum_block, self.last_acc64 = chunk[0]["_blocked"], False
                if hook is not None:
                    gp = _GroupPC(chunk)
                    hook("begin", gp, gp.npix, stream, any(j.get("res") is not None for j in chunk), "conv_igemm_group_f32_kernel")
                check(L.hands_conv2d_group_f32(arr, len(chunk), stream), "hands_conv2d_group_f32")
                if hook is not None:
                    hook("end", gp, gp.npix, stream, False, "conv_igemm_group_f32_kernel")

    def conv_dual(self, L, pc, split, x, x2, B, Ho, Wo, H2, W2, out, stream, act=1, out_off=0):
        """act(conv3(x) + downsample(x2)) (resnet.py:146-154) with the identity never materialised."""
        K0, K1, stride2 = split
        d = ConvDesc(B, Ho, Wo, K0, Ho, Wo, pc.Cout, 1, 1, 1, 0, K0, pc.Cout, 0, pc.Kpad,
                     int(act) | (MATH_BF16X3 if self.math == "bf16x3" else 0))
        hook = self.hook
        if hook is not None:
            hook("begin", pc, B * Ho * Wo, stream, False, "conv_igemm_f32_kernel")
        check(L.hands_conv1x1_dual_nhwc_f32(C.byref(d), ptr(x), ptr(x2), K1, H2, W2, stride2, K1, ptr(pc.w), ptr(pc.bias),
                                            ptr(out, out_off), stream), "hands_conv1x1_dual_nhwc_f32")
        if hook is not None:
            hook("end", pc, B * Ho * Wo, stream, False, "conv_igemm_f32_kernel")

    def stem_pool(self, L, pc, x4, x_off, out, B, H, W, act, stream):
        """conv 7x7/2 + folded BN + act + max-pool 3x3/2 in one kernel (resnet.py:264-268); the conv map
        (B, Hc, Wc, 64) never reaches HBM.  Returns the conv map size (Hc, Wc)."""
        Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        hook = self.hook
        if hook is not None:
            hook("begin", pc, B * Hc * Wc, stream, False, "stem_pool_kernel")
        check(L.hands_stem_conv_maxpool_nhwc_f32(ptr(x4, x_off), ptr(pc.w), ptr(pc.bias), ptr(out), B, H, W, int(act), stream),
              "hands_stem_conv_maxpool_nhwc_f32")
        if hook is not None:
            hook("end", pc, B * Hc * Wc, stream, False, "stem_pool_kernel")
        return Hc, Wc

    def stem_pool_nchw(self, L, pc, x, x_off, out, out_off, B, H, W, act, stream):
        """The same stem reading the NCHW image in place (hands_stem_conv_maxpool_nchw_f32): K = 3 planes x 52 =
        160 instead of 208, and no NHWC4 copy of the input.  ``pc`` is the planar packing (k = plane * 52 + tap)."""
        Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        hook = self.hook
        if hook is not None:
            hook("begin", pc, B * Hc * Wc, stream, False, "stem_pool_planar_kernel")
        check(L.hands_stem_conv_maxpool_nchw_f32(ptr(x, x_off), ptr(pc.w), ptr(pc.bias), ptr(out, out_off), B, H, W, int(act),
                                                 stream), "hands_stem_conv_maxpool_nchw_f32")
        if hook is not None:
            hook("end", pc, B * Hc * Wc, stream, False, "stem_pool_planar_kernel")
        return Hc, Wc


class _GroupPC:
    """What the profiling hook sees for a grouped launch: the summed work of its members (bench.py prices a launch from
    ``group_macs`` / ``group_bytes`` when they exist)."""

    def __init__(self, chunk):
        self.npix = sum(j["_npix"] for j in chunk)
        self.group_macs = sum(j["pc"].macs_per_pixel * j["_npix"] for j in chunk)
        self.group_bytes = sum(4.0 * (j["_npix"] * j["pc"].Cout * (2 if j.get("res") is not None else 1)
                                      + j["_npix"] * j["pc"].stride ** 2 * j["pc"].Cin + j["pc"].w.numel()) for j in chunk)
        pc0 = chunk[0]["pc"]
        self.Cin, self.Cout, self.KH, self.stride, self.w, self.Kpad = pc0.Cin, pc0.Cout, 1, pc0.stride, pc0.w, pc0.Kpad
        self.macs_per_pixel = self.group_macs / max(self.npix, 1)
        self.members = len(chunk)


DEFAULT_ENGINE = ConvEngine()


class EngineSwitches:
    """Mixin: per-instance views of the engine switches under the names the models always used."""

    @property
    def latency_mode(self):
        return self.engine.latency_mode

    @latency_mode.setter
    def latency_mode(self, v):
        self.engine.latency_mode = bool(v)

    @property
    def overlap_trunks(self):
        return self.engine.overlap

    @overlap_trunks.setter
    def overlap_trunks(self, v):
        self.engine.overlap = bool(v)

    @property
    def conv_hook(self):
        return self.engine.hook

    @conv_hook.setter
    def conv_hook(self, fn):
        self.engine.hook = fn
