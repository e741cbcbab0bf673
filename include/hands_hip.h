/*
 * hands_hip.h -- C ABI of libhands_hip.so: the MI355X (gfx950) forward path of the WildHands
 * hand-mesh regressor (hands_light), hand-written HIP.
 *
 * The reference (ap229997/hands) is 100 % Python and has no FFI: the boundary of its hot path is
 * `HandsLight.forward(inputs, meta_info)` (src/models/hands_light/model.py:187-437).  Every entry
 * point below replaces the ATen ops that one statement of that function dispatches; the statement
 * is cited per function.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - All pointers are DEVICE pointers to fp32 unless stated.  The caller owns every buffer; the
 *     library never allocates, frees or synchronises.  Work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream).
 *   - Activations are NHWC ("pixel-major"): element (b,h,w,c) at ((b*H+h)*W+w)*pix_stride + c.
 *   - Return value: 0 on success, otherwise a hipError_t (or HANDS_EINVAL for a bad descriptor);
 *     hands_error_string() gives text.  No global mutable state; re-entrant across streams.
 */
#ifndef HANDS_HIP_H
#define HANDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HANDS_EINVAL 10001
#define HANDS_ABI_VERSION 5

typedef void* hands_stream_t;

int hands_abi_version(void);
const char* hands_error_string(int code);
/* 1 if `stream` is being captured into a hipGraph, 0 if not, < 0 = -(hipError_t).  The host side asks this about the
 * stream it is about to LAUNCH on (torch's "current stream" is not necessarily that stream) before it takes a path that
 * is illegal under capture: growing / zero-filling a workspace (hands_conv2d_nhwc_streamk_f32's epoch flags). */
int hands_stream_is_capturing(hands_stream_t stream);
/* 16 hex digits: sha256 over the kernel sources (every .hip / .h / .cpp of csrc/ and the headers of include/) THIS library was built from.  A stored
 * counter summary (profiles/rNN_pmc_*.json) describes one binary: bench.py uses it only when the hashes agree. */
const char* hands_csrc_sha16(void);

/* ---------------------------------------------------------------------------------------------
 * The chip's own ceilings, measured beside the kernels they bound (csrc/ceilings.hip).  No reference
 * counterpart.  hands_ceiling_mfma_f32: 512 workgroups x 4 waves run `iters` x 32 v_mfma_f32_32x32x2_f32 and nothing else
 * (random_operands != 0: full-mantissa lane-dependent operands instead of constants -- the matrix pipe clocks ~7 % lower on real
 * data); returns the FLOPs the launch executes (time it with events on `stream`), < 0 = -(error).  `out` takes 512 * 256 floats.
 * hands_ceiling_hbm_read_f32: a streaming read of n_floats (use >= 1 GiB: beyond the 256 MB Infinity Cache), 16-byte loads. */
long long hands_ceiling_mfma_f32(float* out, long long out_floats, int iters, int random_operands, hands_stream_t stream);
int hands_ceiling_hbm_read_f32(const float* in, long long n_floats, float* out, hands_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution / linear layer as implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   out[m, n] = act( bias[n] + sum_k patch(m, k) * w[n, k]  (+ residual[m, n]) )
 *   m = (b, ho, wo) flattened, k = (kh, kw, cin) flattened, act selected by desc.act.
 * Replaces nn.Conv2d + eval-mode BatchNorm2d (folded into w/bias by the host) + ReLU + the
 * bottleneck's residual add: src/nets/backbone/resnet.py:134-154,264-280; feature_conv
 * model.py:91-101; and every nn.Linear of the heads (H=W=KH=KW=1): hand_hmr.py:34-40,
 * hmr_layer.py:47-62, model.py:117-125.
 *
 * Weight layout (produced once by hands_amd.packing): [Cout_pad][Kpad] row-major, k ordered
 * (kh, kw, cin), Kpad = round_up(KH*KW*Cin, 16) zero-filled, Cout_pad = round_up(Cout, 128)
 * zero-filled.  bias has Cout_pad entries.  Cin % 4 == 0 and Cout % 4 == 0 are required;
 * Cin % 16 == 0 unless Cin == 4 (the RGB0 stem).
 * Limits (HANDS_EINVAL beyond them; every layer of the three models is far inside): for Cin != 4, padded
 * convolutions take KH, KW <= 15 (tap validity is one 15 + 15 bit word per output pixel), Kpad < 2^21, and
 * (256 / (Ho*Wo) + 2) * H * W * in_pix_stride * 4 < 2^31 (the kernels address a tile's inputs with 32-bit
 * byte offsets from the first image the tile touches).
 * --------------------------------------------------------------------------------------------- */
typedef struct hands_conv_desc {
  int32_t B, H, W, Cin;      /* input  (B,H,W,Cin) */
  int32_t Ho, Wo, Cout;      /* output (B,Ho,Wo,Cout) */
  int32_t KH, KW, stride, pad;
  int32_t in_pix_stride;     /* floats between consecutive input pixels  (>= Cin)  */
  int32_t out_pix_stride;    /* floats between consecutive output pixels (>= Cout) */
  int32_t res_pix_stride;    /* floats between consecutive residual pixels; ignored if residual==NULL */
  int32_t Kpad;              /* packed weight row length */
  int32_t act;               /* HANDS_ACT_*: epilogue activation after bias (+ residual) */
} hands_conv_desc;

#define HANDS_ACT_NONE 0
#define HANDS_ACT_RELU 1
#define HANDS_ACT_GELU 2        /* nn.GELU(): x*0.5*(1+erf(x/sqrt(2)))  (vit.py:76, pose_transformer.py:44) */
#define HANDS_ACT_LEAKY_RELU 3  /* negative slope 0.01 (handoccnet_light/backbone.py) */
#define HANDS_ACT_MASK 0xff
/* Optional arithmetic flag OR'ed into desc.act (default 0 = exact fp32 MFMA, the parity path): both operands are
 * split on the fly into three bf16 planes whose sum is the fp32 value exactly, and the products b_i * b_j with
 * i + j <= 2 run on the bf16 matrix pipe with fp32 accumulation (dropped terms <= 2^-24 of a product).  A separately
 * reported mode: results are fp32-grade but NOT bit-identical to the fp32 chain.  Honoured by
 * hands_conv2d_nhwc_f32 and hands_conv1x1_dual_nhwc_f32 (not the RGB0 stem); the split-K / stream-K entries
 * ignore it (fp32). */
#define HANDS_MATH_BF16X3 0x100
/* Blocked fp32 summation, OR'ed into desc.act (round 5): the k-ordered FMA chain of every output is cut into blocks of 128 /
 * 64 floats (8 / 4 k-steps); a block's sum is added to a second accumulator set in block order -- no fp32 accumulation chain
 * longer than the block, inside the launch (no workspace, no second pass).  Same products, another association: results differ
 * from the single chain by fp32 rounding and are closer to an fp64 evaluation (and to ATen's blocked CPU sums).  A fixed function
 * of the layer: batch-size invariant and deterministic.  Honoured by hands_conv2d_nhwc_f32, hands_conv2d_nhwc_pre_f32 and the
 * split-K entries (each K slice is blocked); ignored by the stem and the dual pointwise entry, the stream-K entry falls back to
 * the plain launch.  At most one of the two, and not together with HANDS_MATH_BF16X3 (HANDS_EINVAL). */
#define HANDS_SUM_BLOCK128 0x200
#define HANDS_SUM_BLOCK64 0x400
/* fp64 accumulation, OR'ed into desc.act (round 6): the same fp32 operands, products accumulated by v_mfma_f64_16x16x4_f64 and the
 * bias added in fp64 -- the stored value is the correctly rounded fp32 of (sum + bias), then residual and activation in fp32 as the
 * reference's separate statements do.  Half the fp32 matrix rate: for the few layers whose rounding a network amplifies
 * (handoccnet_light's heat-map head, hand_head.py:75-94,266-280), not for throughput.  Batch-size invariant and deterministic.
 * Honoured by hands_conv2d_nhwc_f32, hands_conv2d_nhwc_pre_f32 and the split-K entries (the partial sums are fp64 there:
 * hands_conv2d_workspace_floats doubles, the workspace must be 8-byte aligned); the stream-K entry runs it as ONE plain launch;
 * not for the RGB0 stem (Cin == 4) and not together with the flags above (HANDS_EINVAL). */
#define HANDS_ACC_F64 0x800

int hands_conv2d_nhwc_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                          const float* bias, const float* residual, float* out,
                          hands_stream_t stream);

/* Grouped launch (round 6): up to 8 INDEPENDENT pointwise layers (1x1, no padding, any stride) in ONE launch -- the q / k / v
 * projections of an attention block (handoccnet_light/transformer.py:117-135), the FPN laterals (backbone.py:54-57), the two
 * branches of an hourglass level (hand_head.py:217-235), conv1 + downsample of a stage's first bottleneck (resnet.py:134-154).
 * Every job computes exactly what hands_conv2d_nhwc_f32 (pre_scale == NULL) / hands_conv2d_nhwc_pre_f32 (S = 1) would, bit for
 * bit; all jobs of a launch must run the same kernel instantiation: hands_conv2d_group_class(desc, pre) >= 0 and equal for all
 * (-1: not a pointwise layer, or a flag the grouped kernel has no form for -- bf16x3, 128-float blocks, fp64 accumulation),
 * otherwise HANDS_EINVAL and nothing is launched.  The outputs of a launch must not overlap each other or any input of it. */
typedef struct hands_conv_job {
  const hands_conv_desc* desc;
  const float* in;
  const float* w_packed;
  const float* bias;
  const float* residual;      /* or NULL */
  float* out;
  const float* pre_scale;     /* both NULL, or the operand affine of hands_conv2d_nhwc_pre_f32 */
  const float* pre_shift;
} hands_conv_job;
int hands_conv2d_group_class(const hands_conv_desc* d, int pre);
int hands_conv2d_group_f32(const hands_conv_job* jobs, int n, hands_stream_t stream);

/* 3x3 / stride 1 / pad 1 convolution (+ folded BatchNorm bias + activation) as Winograd F(2x2, 3x3) on the fp32 matrix
 * cores (csrc/conv_wino.hip): 16 instead of 36 multiplications per 2x2 output pixels and (cin, cout) pair.  Same layer
 * semantics as hands_conv2d_nhwc_f32 with desc.KH = KW = 3, stride 1, pad 1 and no residual
 * (conv2 / bn2 / relu of a stride-1 Bottleneck: src/nets/backbone/resnet.py:140-142); `u_packed` is the
 * hands_pack_conv3x3_winograd_f64 form of the BN-folded weight, `bias` the same vector hands_pack_conv_f64 produced.
 * All arithmetic is fp32 in a fixed, batch-size-independent order; the result differs from hands_conv2d_nhwc_f32 by
 * fp32 rounding only (Winograd re-associates the 3x3 sum).  Requires Cin % 16 == 0, Cout % 32 == 0, act in
 * {NONE, RELU, LEAKY_RELU}; hands_conv3x3_winograd_supported() returns 1 when `d` can take this route. */
int hands_conv3x3_winograd_supported(const hands_conv_desc* d);
/* multiply-accumulates the matrix cores execute for the layer on this route (16 per 2x2 output tile and (cin, cout) pair,
 * idle tile lanes of partial blocks included); 0 if unsupported.  The algorithmic count is 9 * B * H * W * Cin * Cout. */
long long hands_conv3x3_winograd_executed_macs(const hands_conv_desc* d);
int hands_conv3x3_winograd_f32(const hands_conv_desc* d, const float* in, const float* u_packed, const float* bias,
                               float* out, hands_stream_t stream);

/* The same layer as Winograd F(4x4, 3x3) (csrc/conv_wino4.hip, round 5): 36 multiplications per 4x4 output pixels and
 * (cin, cout) pair -- 2.25 per output against 4 for F(2x2, 3x3) and 9 for the direct form.  Same semantics and restrictions as
 * hands_conv3x3_winograd_f32 (Cin % 8 == 0, Cin >= 16, Cout % 32 == 0, no residual; conv2 / bn2 / relu of a stride-1
 * Bottleneck, src/nets/backbone/resnet.py:140-142); `u_packed` is the hands_pack_conv3x3_winograd4_f64 form of the BN-folded
 * weight.  fp32 throughout in a fixed, batch-size-independent order; the larger transform constants of F(4x4) make the
 * per-layer rounding error ~20x that of F(2x2) (still fp32 noise: DESIGN.md), so a model opts in per layer family
 * (HandsLight: on, HandOccNet: off).  _executed_macs counts 36 per 4x4 tile and (cin, cout) pair, idle tile lanes included. */
int hands_conv3x3_winograd4_supported(const hands_conv_desc* d);
long long hands_conv3x3_winograd4_executed_macs(const hands_conv_desc* d);
int hands_conv3x3_winograd4_f32(const hands_conv_desc* d, const float* in, const float* u_packed, const float* bias,
                                float* out, hands_stream_t stream);

/* Deterministic split-K form of the same layer for latency-bound GEMMs (1x1 / linear layers with few
 * rows and a long K: the HMR / decoder / regressor heads).  hands_conv2d_splitk_factor() returns the
 * number of K slices S the library would use (1 = no split): S = min(8, Kpad/256) for linear layers
 * (H = W = 1) with B <= 2048 rows and Kpad >= 512 -- a function of the layer only, so every output bit
 * is independent of the batch size up to 2048 rows.  The caller provides
 * `workspace` of >= S * M * Cout floats (M = B*Ho*Wo); slices write raw partial sums, a second launch
 * adds them in ascending slice order and applies bias / residual / activation.  With S == 1 or a
 * too-small workspace the call is identical to hands_conv2d_nhwc_f32. */
int hands_conv2d_splitk_factor(const hands_conv_desc* d);
int hands_conv2d_nhwc_splitk_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                 const float* bias, const float* residual, float* out, float* workspace,
                                 long long workspace_floats, hands_stream_t stream);
/* Same with a caller-chosen slice count S (clamped to Kpad/16; any convolution, not only linear
 * layers): the small-batch serving mode, where a layer has a handful of output tiles and a K of
 * 2304-4608 walked serially.  The summation order depends on S, so results are reproducible for a
 * given (layer, S) but not bit-identical to the S = 1 call. */
int hands_conv2d_nhwc_splitk_n_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                   const float* bias, const float* residual, float* out, int S,
                                   float* workspace, long long workspace_floats, hands_stream_t stream);
/* The same with the reduction FUSED into the launch (round 6): `counters` = n_counters ints, zero before the first call and left
 * zero by every call (one array per stream: concurrent launches must not share it).  The slice of a tile that arrives last adds
 * the partial sums in ascending slice order and applies bias / residual / activation -- the output bits of the two-launch form,
 * one launch.  n_counters below the launch's tile count (ceil(M / 128) * ceil(Cout / 128), or ceil(M / 256) for Cout <= 64):
 * falls back to the two launches. */
int hands_conv2d_nhwc_splitk_fused_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                       const float* bias, const float* residual, float* out, int S,
                                       float* workspace, long long workspace_floats, int* counters,
                                       long long n_counters, hands_stream_t stream);

/* Pointwise layer (1x1, no padding) whose INPUT first goes through a per-channel affine + LeakyReLU(0.01):
 *     out = act( bias + W . leaky_relu(in * pre_scale[c] + pre_shift[c]) (+ residual) )
 * i.e. the eval BatchNorm -> LeakyReLU that precedes conv1 of a pre-activation residual unit
 * (handoccnet_light/hand_head.py:131-136,170-175) folded into the convolution's operand staging instead of a launch of
 * hands_bn_leaky_f32 plus a round trip of the activations.  pre_scale / pre_shift: Cin floats (hands_fold_bn_f32 of a unit
 * weight).  S > 1: the same with the deterministic S-slice split-K (workspace as for hands_conv2d_nhwc_splitk_n_f32).
 * Bit-identical to hands_bn_leaky_f32 followed by the plain / split-K launch. */
int hands_conv2d_nhwc_pre_f32(const hands_conv_desc* desc, const float* in, const float* pre_scale, const float* pre_shift,
                              const float* w_packed, const float* bias, const float* residual, float* out, int S,
                              float* workspace, long long workspace_floats, hands_stream_t stream);

/* Stream-K form of hands_conv2d_nhwc_f32 for launches whose tile count quantises badly on the chip (400-1600
 * tiles on 256 CUs leave 12-24 % of it idle): G persistent workgroups take equal shares of the (tile, k-step)
 * units; a tile cut between two workgroups is finished by the second one CONTINUING the first one's fp32 FMA chain
 * from its dumped accumulators, so every output bit equals the plain launch (and stays independent of the batch
 * size).  hands_conv2d_streamk_grid() returns G, or 0 when the library would take the plain launch (short K, few
 * or very many tiles, < 5 % to gain) -- the call is then identical to hands_conv2d_nhwc_f32.
 * workspace: hands_conv2d_streamk_workspace_bytes() bytes of device memory, ZEROED once by the caller, private to
 * one stream at a time; epoch: any nonzero value different from the one passed with the previous call on the
 * same workspace (a per-workspace counter), POSITIVE in production.  A negative epoch is a test hook: no workgroup
 * publishes its hand-off, so every consumer takes the bounded-wait fallback and recomputes the missing k-steps
 * itself (slow, same bits). */
long long hands_conv2d_streamk_workspace_bytes(void);
int hands_conv2d_streamk_grid(const hands_conv_desc* d);
int hands_conv2d_nhwc_streamk_f32(const hands_conv_desc* d, const float* in, const float* w_packed,
                                  const float* bias, const float* residual, float* out, void* workspace,
                                  long long workspace_bytes, int epoch, hands_stream_t stream);

/* Two 1x1 convolutions summed into one output: out = act(W0 * in + W1 * in2[stride2-sampled] + bias).
 * `d` describes the first one (stride 1, H = Ho, W = Wo, Cin = K0) with Kpad = K0 + Cin2; the packed
 * weight row is [W0 (K0) | W1 (Cin2)], both with their BatchNorm folded, bias = b0 + b1.  This is the
 * last convolution of a ResNet bottleneck fused with the block's downsample branch
 * (src/nets/backbone/resnet.py:134-154: `out = bn3(conv3(out)); identity = downsample(x); out += identity`):
 * the identity tensor is never written to or re-read from HBM.  in2 is (B, H2, W2, Cin2) with
 * in2_pix_stride floats per pixel; output pixel (ho, wo) reads in2 pixel (ho*stride2, wo*stride2). */
int hands_conv1x1_dual_nhwc_f32(const hands_conv_desc* d, const float* in, const float* in2, int Cin2, int H2,
                                int W2, int stride2, int in2_pix_stride, const float* w_packed,
                                const float* bias, float* out, hands_stream_t stream);

/* The ResNet stem as one kernel: conv 7x7 / stride 2 / pad 3 on the RGB0 image (B,H,W,4) with the
 * folded eval-BatchNorm, activation, and max-pool 3x3 / stride 2 / pad 1 -- conv1, bn1, relu, maxpool of
 * src/nets/backbone/resnet.py:264-268 (LeakyReLU variant: src/models/handoccnet_light/backbone.py:44-47).
 * w_packed is the stem's hands_conv2d_nhwc_f32 weight ([128][208], k = (kh, kw, rgb0)), bias [>=64];
 * out (B, Hp, Wp, 64) with Hc = (H-1)/2+1, Hp = (Hc-1)/2+1.  Bit-identical to hands_conv2d_nhwc_f32
 * followed by hands_maxpool3x3s2_nhwc_f32; the (B,Hc,Wc,64) map is never written to HBM. */
int hands_stem_conv_maxpool_nhwc_f32(const float* x4, const float* w_packed, const float* bias, float* out,
                                     int B, int H, int W, int act, hands_stream_t stream);

/* The same stem reading the reference's NCHW image (B,3,H,W) directly -- no NHWC4 conversion launch -- with the
 * contraction ordered (colour plane, kh, kw): w_planar is [128][160] row-major, k = plane * 52 + kh * 7 + kw (taps
 * 49..51 of each plane and k >= 156 zero), i.e. hands_pack_linear_f64 of the folded (64, 147) weight with
 * col_index[c * 49 + tap] = c * 52 + tap, k_total = 160.  K = 160 instead of 208: 23 % fewer MFMAs.  Same result as
 * hands_stem_conv_maxpool_nhwc_f32 up to the fp32 rounding of a different summation order. */
int hands_stem_conv_maxpool_nchw_f32(const float* x_nchw, const float* w_planar, const float* bias, float* out,
                                     int B, int H, int W, int act, hands_stream_t stream);

/* NCHW (B,3,H,W) image batch -> NHWC with C padded to 4 (4th channel = 0).
 * Replaces the implicit layout of inputs["img"|"r_img"|"l_img"] (model.py:188,238-239). */
int hands_nchw3_to_nhwc4_f32(const float* in, float* out, int B, int H, int W, hands_stream_t stream);

/* MaxPool2d(kernel 3, stride 2, padding 1) on NHWC.  resnet.py:268. C % 4 == 0. */
int hands_maxpool3x3s2_nhwc_f32(const float* in, float* out, int B, int H, int W, int C,
                                hands_stream_t stream);

/* feat_vec[b, c] = sum_p feat[b, p, c]  (a SUM over the 7x7 map, not a mean).  model.py:196. */
int hands_sumpool_nhwc_f32(const float* feat, float* out, int B, int HW, int C, int out_stride,
                           hands_stream_t stream);

/* out[b, c] = mean_p feat[b, p, c]: nn.AdaptiveAvgPool2d(1) at the top of HandHMR.forward(use_pool=True)
 * (src/nets/hand_heads/hand_hmr.py:73-78), taken when HandsLight runs with no_crops (src/models/hands_light/model.py:316-318,
 * the arctic_light configuration): both heads read the pooled GLOBAL feature map.  Same summation order as
 * hands_sumpool_nhwc_f32, then one division by HW. */
int hands_avgpool_nhwc_f32(const float* feat, float* out, int B, int HW, int C, int out_stride,
                           hands_stream_t stream);

/* Image-level positional encodings, pos_enc = 'center' (mode 1) | 'corner' (mode 2) | 'center+corner' (mode 3)
 * (src/models/hands_light/model.py:203-218: torch.cat([img, enc.view(bz,-1,1,1).repeat(1,1,w,h)], dim=1)):
 * out (B, H, W, Cpad) NHWC = [r g b | center enc 4 n_freq | corner enc 16 n_freq | zero padding], img_nchw (B,3,H,W),
 * center_angle (B,2), corner_angle (B,8); the encoding is that of hands_kpe_concat_f32 (model.py:444-460).  The result
 * is the input of the hand trunk's widened conv1 through hands_conv2d_nhwc_f32 (Cin = Cpad, a multiple of 16). */
int hands_image_posenc_nhwc_f32(const float* img_nchw, const float* center_angle, const float* corner_angle, float* out,
                                int B, int H, int W, int n_freq, int mode, int Cpad, hands_stream_t stream);


/* Per-pixel positional encodings, pos_enc = 'dense' | 'dense_latent' | 'cam_conv' (src/models/hands_light/model.py:462-481, call
 * sites :220-224, :244-256, :276-288).  angle (B, Ca, Hs, Ws) NCHW, mask (B, Hs, Ws).  n_freq > 0: the 2 n_freq Ca channels
 * sin / cos(2^k angle[ci]) laid out (k, ci, {sin, cos}); n_freq == 0: the Ca raw maps ('cam_conv'); times the mask; then
 * F.interpolate(bilinear, align_corners=True) to (R, R) (args.img_res_ds) and from there to (Ho, Wo) (Ho == R: the first only).
 * The result goes to channels [c_off, c_off + Cenc) of the NHWC map out (B, Ho, Wo, ld).  img_nchw != NULL ((B, 3, Ho, Wo),
 * c_off == 3): the whole pixel is written, [r g b | encoding | zeros] -- the input of the hand trunk's widened conv1
 * ('dense'); NULL: only the encoding channels (the 7x7 map hands_concat_nhwc_f32 appends to the features). */
int hands_dense_posenc_f32(const float* angle, const float* mask, const float* img_nchw, float* out, int B, int Ca, int Hs, int Ws,
                           int n_freq, int R, int Ho, int Wo, int ld, int c_off, hands_stream_t stream);

/* torch.cat along the channels of NHWC maps: out[b, p, :] = [a[b, p, :Ca] (+ add[b % Bg, p, :Ca]) | extra[b, p, :Cb] | zeros],
 * row strides lda / ld_add / Cb / ld, extra_batch_stride floats between samples of `extra` (0: one map for every sample).
 * model.py:252-256, 284-288 (features (+ global features) with the resized per-pixel maps), :177-185 (`broadcast`: the depth
 * head's (x, y) grid).  add / extra may be NULL (Cb = 0). */
int hands_concat_nhwc_f32(const float* a, int lda, int Ca, const float* add, int ld_add, const float* extra,
                          long long extra_batch_stride, int Cb, float* out, int ld, int B, int Bg, int HW, hands_stream_t stream);

/* out (B, H, W, C) = nn.Upsample / F.interpolate(x (B, h, w, C), mode='bilinear', align_corners=True): the three upsamplings of
 * the depth head (model.py:141, 146, 151).  C % 4 == 0. */
int hands_upsample_bilinear_ac_f32(const float* x, float* out, int B, int h, int w, int H, int W, int C, hands_stream_t stream);

/* pos_enc = 'pcl' (model.py:330-334): rotmat[b, 0] = rot[b] @ rotmat[b, 0] in place; rotmat (B, 16, 3, 3), rot (B, 3, 3). */
int hands_rot_leftmul_f32(float* rotmat, const float* rot, int B, hands_stream_t stream);

/* pos_enc = 'perspective_correction' (model.py:370-376): rot_swapped[b, 0] = Rx(-c[b,0]) Ry(-c[b,1]) @ rot_swapped[b, 0] on the
 * rotations AFTER the is_flipped swap (2 Bg rows: right then left; center_angle (2 Bg, 2)).  pytorch3d.euler_angles_to_matrix
 * ('XYZ') is absent from the reference tree: published definition, parity unpinned.  The reference assigns in place: in a batch
 * without a flipped sample that tensor is the heads' own output, which the grasp head reads afterwards -- the corrected matrix
 * is then also written to rotmat (the unswapped rotations); with a flipped sample rotmat stays as it is. */
int hands_perspective_correction_f32(float* rot_swapped, float* rotmat, const float* center_angle, const int64_t* is_flipped,
                                     int Bg, hands_stream_t stream);

/* out[b2, p, :] = cat(crop[b2,p,:] + glb[b2 % Bg, p, :], center_enc(b2), corner_enc(b2)).
 * crop holds the right-hand samples then the left-hand samples (2*Bg rows); encodings are the
 * reference's [sin(2^k a), cos(2^k a)] laid out (L, c, 2).  model.py:258-271, 444-460.
 * center_angle (2*Bg, 2), corner_angle (2*Bg, 8); out channels = C + 4*L + 16*L.
 * glb == NULL: use_glb_feat = False, the crop features are concatenated as they are (model.py:266-267). */
int hands_kpe_concat_f32(const float* crop, const float* glb, const float* center_angle,
                         const float* corner_angle, float* out, int B2, int Bg, int HW, int C,
                         int n_freq, hands_stream_t stream);

/* HMR state rows (ld = F + 112, every segment 16-byte aligned):
 *   [feat F | pose6d 96 | shape 10 | 0 0 | cam 3 | 0]
 * Writes the identity-6D / zero-shape initial vectors (hand_hmr.py:46-56) and copies the cam_init
 * MLP output (rows of 4 floats: s, tx, ty, pad) into columns F..F+111 of every row. */
int hands_hmr_init_f32(float* state, const float* cam_init, int B, int ld, int F,
                       hands_stream_t stream);

/* rotation_6d_to_matrix (pytorch3d; call site hand_hmr.py:85-87): Gram-Schmidt, rows.
 * pose6d rows have stride ld6; rotmat is (B,16,3,3) contiguous. */
int hands_rot6d_to_matrix_f32(const float* pose6d, int ld6, float* rotmat, int B,
                              hands_stream_t stream);

/* is_flipped branch of HandsLight.forward (model.py:341-368): per-sample swap of the two hands'
 * predictions with mirrored axis-angle (y,z negated) and cam ty sign flip.  Arrays hold the right
 * hand in rows [0,Bg) and the left hand in rows [Bg,2Bg).  is_flipped: int64 (Bg).
 * In/out: rotmat (2Bg,16,3,3), shape (2Bg,10), cam (2Bg,3), cam_init (2Bg,3) -> *_out. */
int hands_flip_swap_f32(const int64_t* is_flipped, const float* rotmat, const float* shape,
                        const float* cam, const float* cam_init, float* rotmat_out,
                        float* shape_out, float* cam_out, float* cam_init_out, int Bg,
                        hands_stream_t stream);

/* The rotation conversions of the MANO head on their own (the same device functions
 * hands_mano_pose_f32 and hands_flip_swap_f32 inline): matrix_to_axis_angle =
 * quaternion_to_axis_angle(matrix_to_quaternion(.)) (common/rot.py:118-193, 55-83; called at
 * src/nets/hand_heads/mano_head.py:27-31) and its inverse axis_angle_to_matrix (common/rot.py:754-782,
 * 86-115; model.py:345-353).  rotmat (n,3,3) row-major, axis_angle (n,3). */
int hands_matrix_to_axis_angle_f32(const float* rotmat, float* axis_angle, long long n, hands_stream_t stream);
int hands_axis_angle_to_matrix_f32(const float* axis_angle, float* rotmat, long long n, hands_stream_t stream);

/* grasp-head input rows [feat_vec(F) | rotmat(144) | shape(10) | 0-pad]: the reference's
 * cat([shape, pose.view(bz,-1), feat_vec]) (model.py:401-404) with the columns permuted so the
 * segments are 16-byte aligned; the packed grasp_classifier.0 weight uses the same permutation.
 * Rows [0,Bg) are the right hand, [Bg,2Bg) the left; feat_vec (Bg,F) is shared by both. */
int hands_grasp_input_f32(const float* shape, int ld_shape, const float* rotmat,
                          const float* feat_vec, float* out, int B2, int Bg, int F, int ld_out,
                          hands_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * MANO layer (MANOHead.forward, src/nets/hand_heads/mano_head.py:21-65 -> common/rot.py:118-193,
 * smplx.MANO / smplx.lbs, common/camera.py:456-474, common/transforms.py:316-329,
 * common/data_utils.py:361-365).  Three launches per hand side:
 *   1. hands_mano_pose_f32     rotmat -> axis-angle -> (+pose_mean) -> Rodrigues -> pose feature;
 *                              joints J(beta); forward kinematics -> skinning transforms A (B,16,12)
 *                              and posed joints (B,16,3); writes the blend-GEMM input rows
 *                              [beta(10) | pose_feature(135) | 0-pad] (B, ld_blend).
 *   2. hands_conv2d_nhwc_f32   v_posed = v_template + [beta|pf] @ [shapedirs; posedirs]
 *                              (the 778x3 x 145 contraction on MFMA), output (B, ld_vp).
 *   3. hands_mano_skin_f32     per-vertex blend of A, apply, fingertip joints, weak-perspective
 *                              camera, K-projection, 2x/res-1 normalisation.
 * --------------------------------------------------------------------------------------------- */
typedef struct hands_mano_consts {
  const float* pose_mean;   /* (48)      zeros(3) ++ hands_mean(45) */
  const float* J_template;  /* (16,3)    J_regressor @ v_template */
  const float* J_shapedirs; /* (48,10)   J_regressor @ shapedirs */
  const float* lbs_weights; /* (778,16) */
  const int32_t* tip_ids;   /* (5) */
} hands_mano_consts;

int hands_mano_pose_f32(const hands_mano_consts* c, const float* rotmat, const float* betas,
                        int ld_betas, float* blend_in, int ld_blend, float* A, float* joints16,
                        int B, hands_stream_t stream);

typedef struct hands_mano_out {
  float* vertices;  /* (B,778,3) */
  float* joints3d;  /* (B,21,3)  */
  float* v3d_cam;   /* (B,778,3) */
  float* j3d_cam;   /* (B,21,3)  */
  float* j2d_norm;  /* (B,21,2)  */
  float* cam_t;     /* (B,3)     */
} hands_mano_out;

int hands_mano_skin_f32(const hands_mano_consts* c, const float* v_posed, int ld_vp,
                        const float* A, const float* joints16, const float* cam_wp,
                        const float* K, float img_res, float min_s, const hands_mano_out* out,
                        int B, hands_stream_t stream);

/* MANOHead.forward for ONE or BOTH hands in a single launch: everything hands_mano_pose_f32 ->
 * hands_conv2d_nhwc_f32 (blend) -> hands_mano_skin_f32 compute, fused (pose / forward kinematics, the
 * 2334 x 145 blend contraction on the fp32 matrix cores with the 16 hands of a block as the MFMA N
 * dimension, skinning, camera, projection).  sides[s] carries that side's constants (hands_pack_mano_f32),
 * inputs (rot (B,16,3,3) rotation matrices -- or (B,48) axis-angle when axis_angle_input != 0 --, betas
 * (B, ld_betas), cam_wp (B,3)) and outputs; K (B,3,3) is shared.  Replaces mano_head.py:21-65 for
 * mano_r and mano_l (model.py:378-390) and the ground-truth MANO pass (process_arctic.py:16-40). */
typedef struct hands_mano_side {
  hands_mano_consts consts;
  const float* blend_w;     /* [2432][160] */
  const float* blend_bias;  /* [2432] */
  const float* rot;
  const float* betas;
  const float* cam_wp;
  hands_mano_out out;
} hands_mano_side;

int hands_mano_heads_f32(const hands_mano_side* sides, int n_sides, const float* K, int ld_betas,
                         float img_res, float min_s, int B, int axis_angle_input, hands_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * hamer_light (ViT-H/16 + cross-attention decoder head): src/models/hamer_light/.
 * GEMMs (patch embed, qkv, proj, MLP, to_kv, decoders ...) run on hands_conv2d_nhwc_f32.
 * --------------------------------------------------------------------------------------------- */

/* F.interpolate(bilinear, align_corners=False) (Hin,Win)->(S,S) of an NCHW (B,3,..) batch, keep
 * columns [col0, col0+Wc), write NHWC4 (B,S,Wc,4).  hamer_light/model.py:82-100. */
int hands_resize_crop_nchw3_to_nhwc4_f32(const float* in, float* out, int B, int Hin, int Win, int S,
                                         int col0, int Wc, hands_stream_t stream);

/* LayerNorm over the last dim C in {256,1024,1280}; out = LN(x)*gamma+beta (+ addvec[row/rows_per_vec]).
 * vit.py:128-151,338 (eps 1e-6), pose_transformer.py:22-33 (eps 1e-5); the optional add is the KPE
 * re-added to the final feature map (hamer_light/model.py:102-104). */
int hands_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out,
                        const float* addvec, int rows_per_vec, int M, int C, float eps,
                        hands_stream_t stream);

/* x[b,t,:] = ((x[b,t,:] + pos[1+t,:]) + pos[0,:]) + vec[b,:]   (vec may be NULL).  vit.py:326-330. */
int hands_add_pos_f32(float* x, const float* pos, const float* vec, int B, int T, int C,
                      hands_stream_t stream);

/* rows [center_enc 4L | corner_enc 16L | 0-pad] of the key-point encoding (pos_emb.py:53-67). */
int hands_kpe_encode_f32(const float* center_angle, const float* corner_angle, float* out, int B, int ld,
                         int n_freq, hands_stream_t stream);

/* softmax((scale*q) k^T) v per (batch, head) on fp32 MFMA.  qkv rows are tokens: [q | k | v], each
 * heads*head_dim wide (vit.py:110-126).  Built for T=192, head_dim=80 (ViT-H/16 at 256x192). */
int hands_attention_f32(const float* qkv, float* out, int B, int T, int heads, int head_dim, float scale,
                        hands_stream_t stream);

/* one query token per sample against T context tokens: q (B, heads*64), kv rows [k | v] (B*T,
 * 2*heads*64) -> out (B, heads*64); dots = (q.k)*scale (pose_transformer.py:113-123). */
int hands_cross_attention_1q_f32(const float* q, const float* kv, float* out, int B, int T, int heads,
                                 int head_dim, float scale, hands_stream_t stream);

/* rot6d_to_rotmat with b1,b2,b3 as COLUMNS (hamer_light/geometry.py:47-62). */
int hands_rot6d_to_matrix_cols_f32(const float* pose6d, int ld6, float* rotmat, int B,
                                   hands_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * handoccnet_light (LeakyReLU ResNet-50 + FPN + CBAM gate + FIT/SET attention + hourglass regressor):
 * src/models/handoccnet_light/.  Convolutions and linear layers run on hands_conv2d_nhwc_f32
 * (HANDS_ACT_LEAKY_RELU epilogue); all tensors NHWC, C % 4 == 0.
 * --------------------------------------------------------------------------------------------- */

/* out = F.interpolate(x (B,h,w,C) -> (H,W), bilinear, align_corners=False) + y.  backbone.py:40-42. */
int hands_upsample_bilinear_add_f32(const float* x, const float* y, float* out, int B, int h, int w,
                                    int H, int W, int C, hands_stream_t stream);

/* 2x2 stride-2 pooling: mode 0 = AvgPool2d (backbone.py:38,62), 1 = max (hand_head.py:219,276). */
int hands_pool2x2_nhwc_f32(const float* in, float* out, int B, int H, int W, int C, int mode,
                           hands_stream_t stream);

/* ChannelPool (cbam.py:66-68): out[pix] = [max_c x, mean_c x, 0, 0]; C must be 256. */
int hands_channel_pool_f32(const float* x, float* out, long long npix, int C, hands_stream_t stream);

/* SpatialGate tail (cbam.py:78-82): s = sigmoid(logit[pix*logit_stride]); primary = x*s;
 * secondary = x*(1-s). */
int hands_gate_apply_f32(const float* x, const float* logit, int logit_stride, float* primary,
                         float* secondary, long long npix, int C, hands_stream_t stream);

/* out_q = (query + q_emb[t]) + kpe[b]; out_k = (key + k_emb[t]) + kpe[b]; tokens (B,N,C), embeddings
 * (N,C), kpe (B,C).  transformer.py:119-134. */
int hands_add_embed2_f32(const float* query, const float* key, const float* q_emb, const float* k_emb,
                         const float* kpe, float* out_q, float* out_k, int B, int N, int C,
                         hands_stream_t stream);

/* out[b,t,:] = x[b,t,:] + vec[b,:]   (handoccnet_light/model.py:88-89). */
int hands_add_rowvec_f32(const float* x, const float* vec, float* out, int B, int N, int C,
                         hands_stream_t stream);

/* out[b,c] = sum_t x[b,t,c]. */
int hands_token_sum_f32(const float* x, float* out, int B, int N, int C, hands_stream_t stream);

/* out = LeakyReLU_0.01(x * scale[c] + shift[c]): eval BatchNorm (folded by the host) + activation
 * in front of a convolution (pre-activation residual units, hand_head.py:131-133,170-172). */
int hands_bn_leaky_f32(const float* x, const float* scale, const float* shift, float* out,
                       long long npix, int C, hands_stream_t stream);

/* out = up1 + nearest_upsample_2x(low (B,h,w,C))   (hand_head.py:228-230). */
int hands_upsample_nearest2x_add_f32(const float* low, const float* up1, float* out, int B, int h,
                                     int w, int C, hands_stream_t stream);

/* heatmaps[b,t,j] = softmax_t(latents[b,t,j] * betas[j]) for j < J, 0 for J <= j < ld_out
 * (hand_head.py:62-67).  Row strides ld_in / ld_out. */
int hands_spatial_softmax_f32(const float* latents, int ld_in, const float* betas, float* heatmaps,
                              int ld_out, int B, int N, int J, hands_stream_t stream);

/* softmax((q k^T) * scale) v per (batch, head) with online softmax on fp32 MFMA; q,k,v,out (B*N,
 * heads*64).  Optional FIT gate: output rows scaled by sigmoid((q2 . k2sum[b]) * scale), k2sum (B,
 * heads*64) = sum over the tokens of k2; optional residual added to the output.
 * transformer.py:71-98,146-153.  N % 128 == 0, head_dim == 64. */
int hands_flash_attention_f32(const float* q, const float* k, const float* v, const float* q2,
                              const float* k2sum, const float* resid, float* out, int B, int N,
                              int heads, int head_dim, float scale, hands_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Evaluation metrics on the (gathered) predictions, on device: src/utils/eval_modules.py:97-134
 * (mpjpe/ra/h), :136-219,320-343 (mpjpe/pa/ra/{r,l,h}, Procrustes with a 3x3 SVD per hand),
 * :386-407 (mrrpe/r/l), :410-428 (pix_err/{r,l}); common/metrics.py:23-55.  Millimetres / pixels,
 * NaN conventions as the reference (invalid hand -> NaN, except the PA error which it multiplies
 * by the validity flag).  All arrays fp32: joints (B,21,3), 2-D joints (B,21,2) in pixels,
 * per-sample flags (B), per-joint flags (B,21).
 * --------------------------------------------------------------------------------------------- */
typedef struct hands_eval_in {
  const float *pred_j3d_r, *pred_j3d_l, *gt_j3d_r, *gt_j3d_l;
  const float *pred_j2d_r, *pred_j2d_l, *gt_j2d_r, *gt_j2d_l;
  const float *is_valid, *right_valid, *left_valid, *joints_valid_r, *joints_valid_l;
} hands_eval_in;

typedef struct hands_eval_out {
  float *mpjpe_ra_h, *mpjpe_pa_ra_r, *mpjpe_pa_ra_l, *mpjpe_pa_ra_h, *mrrpe_rl; /* (B) each */
  float *pix_err_r, *pix_err_l;                                                  /* (B,21) */
} hands_eval_out;

int hands_eval_metrics_f32(const hands_eval_in* in, const hands_eval_out* out, int B,
                           hands_stream_t stream);

/* GT preprocessing of the wrapper's test mode (src/callbacks/process/process_arctic.py:4-75):
 *   hands_mano_pose_aa_f32   = hands_mano_pose_f32 for AXIS-ANGLE input (B,48) (GT MANO parameters);
 *   hands_gt_targets_f32     : Tr0 = mean_j(j3d_full - joints); v3d_cam = verts + Tr0;
 *                              cam_t = j3d_full[0] - joints[0];
 *                              cam_t_wp = [2 f / (res * cam_t.z + 1e-9), cam_t.x, cam_t.y]  (common/camera.py:10-29);
 *   hands_unnormalize_kp2d_f32: 0.5 * res * (x + 1)  (common/data_utils.py:368-373, generic/wrapper.py:118-134). */
int hands_mano_pose_aa_f32(const hands_mano_consts* c, const float* axis_angle, const float* betas,
                           int ld_betas, float* blend_in, int ld_blend, float* A, float* joints16,
                           int B, hands_stream_t stream);
int hands_gt_targets_f32(const float* joints, const float* verts, const float* j3d_full, const float* K,
                         float img_res, float* v3d_cam, float* cam_t, float* cam_t_wp, int B, int NV,
                         hands_stream_t stream);
int hands_unnormalize_kp2d_f32(const float* x, float* out, long long n, float img_res,
                               hands_stream_t stream);

/* ---- (f2) crop / KPE front-end, the step before the path -------------------------------------------
 * hands_frontend_boxes_f32: per sample and hand, from the normalised GT 2-D joints (B,21,ld>=2):
 *   box [x0,y0,w,h] = int16-truncated min/max of ((j+1)/2)*(res-1), clipped to [0,res-1]
 *                                                  (src/datasets/hands_light_dataset.py:137-152);
 *   crop window [x0,y0,x1,y1] of crop_and_pad (common/data_utils.py:495-509; an empty box -> whole image);
 *   the float32 2x3 affine gen_trans_from_patch_cv builds for it (common/data_utils.py:56-91, rot=0);
 *   center (2) / corner (8) angles atan2(p - c, f) of the window (hands_light_dataset.py:256-279); K == NULL: args.no_intrx
 *   (hands_light_dataset.py:247-253) -- the encodings use f = c = img_res / 2 (a float64 matrix: the corner angles are then
 *   evaluated in double like the center angles) instead of the camera's intrinsics; hands_frontend_dense_maps_f32 likewise.
 *   bbox_og = the box itself, or [0,0,res-1,res-1] when empty (hands_light_dataset.py:145-152).
 * hands_warp_affine_cubic_norm_f32: cv2.warpAffine(src, trans, (Wo,Ho), INTER_CUBIC) with constant-0
 *   border (generate_patch_image_clean, common/data_utils.py:423-460), then clip to [0,1] and
 *   (x - mean)/std (hands_light_dataset.py:171-178,528).  src (B,3,H,W) in [0,1], out (B,3,Ho,Wo);
 *   trans (B,6) device pointer or NULL for the identity; mean3/std3 are HOST pointers to 3 floats.
 *   OpenCV is not vendored by the reference: the kernel follows its published fixed-point cubic
 *   remap (see oracle/frontend_oracle.py, "parity unpinned"). */
/* Per-pixel maps of the crop windows for pos_enc 'dense' / 'dense_latent' (nch = 2) and 'cam_conv' (nch = 6)
 * (src/datasets/hands_light_dataset.py:281-333): angle (B, nch, img_res, img_res) = atan2(x - cx, fx), atan2(y - cy, fy)
 * (, x - cx, y - cy, 2 x / img_res - 1, 2 y / img_res - 1) for the window pixels x in [x0, x1], y in [y0, y1] -- first map index x
 * -- in the top-left corner of a zero map; mask (B, img_res, img_res) = 1 on the window.  bbox (B, 4) int32 [x0, y0, x1, y1] as
 * written by hands_frontend_boxes_f32, K (B, 3, 3).  The inputs of hands_dense_posenc_f32. */
int hands_frontend_dense_maps_f32(const int32_t* bbox, const float* K, float* angle, float* mask, int B, int img_res, int nch,
                                  hands_stream_t stream);

int hands_frontend_boxes_f32(const float* j2d_r, const float* j2d_l, int ld, const float* K, int B,
                             int img_res, int out_res, double bbox_scale,
                             int32_t* bbox_r, int32_t* bbox_l, int32_t* bbox_og_r, int32_t* bbox_og_l,
                             float* trans_r, float* trans_l, float* center_r, float* center_l,
                             float* corner_r, float* corner_l, hands_stream_t stream);
int hands_warp_affine_cubic_norm_f32(const float* src, const float* trans, float* out, int B, int H, int W,
                                     int Ho, int Wo, const float* mean3, const float* std3,
                                     hands_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * One-time HOST-side packing (csrc/pack.cpp): reference-layout parameters -> the layouts above.
 * Host pointers only, no GPU call, no allocation kept.  A host in any language packs a reference
 * checkpoint with these and uploads the results; hands_amd/packing.py is a thin ctypes wrapper.
 *   hands_pack_conv_dims   sizes of the packed buffers for a (Cout, Cin, KH, KW) convolution whose input
 *                          channels are padded to cin_pad_to (0 = Cin; 4 for the RGB0 stem).
 *   hands_fold_bn_f32      eval-mode BatchNorm2d folded into the preceding conv, in fp64
 *                          (resnet.py:137-149 applies them separately): per_out = Cin*KH*KW.
 *   hands_pack_conv_f64    (Cout,Cin,KH,KW) fp64 -> w_packed [Cout_pad][Kpad] fp32, bias [Cout_pad] (NULL = 0).
 *   hands_pack_linear_f64  nn.Linear (N,K) as a 1x1 conv with optional input-column / output-row
 *                          permutations (the HMR state row, the grasp row, the NCHW nn.Flatten order).
 *   hands_pack_conv1x1_dual_f64  weight of hands_conv1x1_dual_nhwc_f32: [W0 | W1], bias b0 + b1.
 *   hands_pack_mano_f32    MANO constants (hands_mano_consts members + the blend GEMM weight/bias).
 *   hands_pack_conv3x3_winograd_f64  U = G g G^T (fp64, rounded once) of a (Cout, Cin, 3, 3) weight in MFMA operand
 *                          order for hands_conv3x3_winograd_f32; hands_pack_conv3x3_winograd_floats = 16 * Cout * Cin
 *                          (0 if the shape is not supported: Cout % 32, Cin % 16).
 *   hands_pack_conv3x3_winograd4_f64  the F(4x4, 3x3) form: U = G g G^T with the 6x3 G of Lavin & Gray, operand order
 *                          [Cout/32][Cin/8][f = 6 xi + nu][lane 64][4] for hands_conv3x3_winograd4_f32; _floats = 36 * Cout * Cin
 *                          (0 if unsupported: Cout % 32, Cin % 8).
 * hands_conv2d_workspace_floats: floats of split-K workspace hands_conv2d_nhwc_splitk_n_f32 needs for
 *   S slices (S <= 0: the library's own hands_conv2d_splitk_factor); 0 when no split is taken.
 * --------------------------------------------------------------------------------------------- */
typedef struct hands_packed_dims {
  int32_t Cin;       /* channels the kernel sees (padded) = desc.Cin */
  int32_t Cout;      /* channels the kernel stores = round_up(Cout, 4) = desc.Cout */
  int32_t Cout_pad;  /* rows of w_packed / entries of bias = round_up(Cout, 128) */
  int32_t Kpad;      /* row length of w_packed = round_up(KH*KW*Cin, 16) = desc.Kpad */
} hands_packed_dims;

int hands_pack_conv_dims(int Cout, int Cin, int KH, int KW, int cin_pad_to, hands_packed_dims* dims);
int hands_fold_bn_f32(int Cout, long long per_out, const float* w, const float* gamma, const float* beta,
                      const float* mean, const float* var, double eps, double* w_folded, double* bias_folded);
int hands_pack_conv_f64(int Cout, int Cin, int KH, int KW, int cin_pad_to, const double* w_oihw,
                        const double* bias, float* w_packed, float* bias_packed);
int hands_pack_linear_f64(int N, int K, const double* w, const double* bias, const int32_t* col_index,
                          int k_total, const int32_t* row_index, int n_total, float* w_packed,
                          float* bias_packed);
int hands_pack_conv1x1_dual_f64(int Cout, int K0, int K1, const double* w0, const double* b0,
                                const double* w1, const double* b1, float* w_packed, float* bias_packed);
/* blend_w_packed (2432, 160): row m = [shapedirs[m, 0..9] | posedirs[0..134, m] | v_template[m] | 14 zeros], rows 2334.. zero;
 * blend_bias_packed (2432) = v_template.  hands_mano_heads_f32 multiplies column 145 with a constant 1 in its input rows (its
 * accumulators start at zero: no bias load in front of a tile); the three-launch chain (hands_mano_pose_f32 -> GEMM ->
 * hands_mano_skin_f32) writes 0 there and adds blend_bias_packed in the GEMM epilogue. */
int hands_pack_mano_f32(const float* v_template, const float* shapedirs, const float* posedirs,
                        const float* J_regressor, const float* hands_mean, float* pose_mean,
                        float* J_template, float* J_shapedirs, float* blend_w_packed,
                        float* blend_bias_packed);
long long hands_conv2d_workspace_floats(const hands_conv_desc* d, int S);
long long hands_pack_conv3x3_winograd_floats(int Cout, int Cin);
int hands_pack_conv3x3_winograd_f64(int Cout, int Cin, const double* w_oihw, float* u_packed);
long long hands_pack_conv3x3_winograd4_floats(int Cout, int Cin);
int hands_pack_conv3x3_winograd4_f64(int Cout, int Cin, const double* w_oihw, float* u_packed);

#ifdef __cplusplus
}
#endif
#endif /* HANDS_HIP_H */
