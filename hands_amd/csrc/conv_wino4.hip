// conv_wino4.hip -- 3x3 / stride 1 / pad 1 NHWC convolution as Winograd F(4x4, 3x3) on the gfx950 fp32 matrix cores.
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray; d = 6x6 input patch, Y = 4x4 output pixels)
//
// 36 multiplications per 4x4 outputs and (cin, cout) pair: 2.25 per output against 4 for F(2x2,3x3) (conv_wino.hip) and 9 for
// the direct implicit GEMM (conv_igemm.hip).  Replaces F.conv2d(3x3, s1, p1) + eval BatchNorm2d (folded) + ReLU of
// src/nets/backbone/resnet.py:140-142 (conv2 / bn2 / relu of every stride-1 Bottleneck).
//
// Per frequency f = (xi, nu) the layer is a GEMM   M_f[o, t] = sum_c U_f[o, c] * V_f[t, c]   (t = 4x4 output tile).
// One workgroup = 16 waves = 32 tiles (512 output pixels) x 32 output channels x all 36 frequencies, one per CU:
//   * waves 0-11 ("consumers") own three frequencies each (wave = (xi, nu half): 3 x 16 accumulator registers) and run a
//     plain GEMM k-loop of v_mfma_f32_32x32x2_f32: the weight fragment U (G g G^T, fp64 on the host, one rounding;
//     hands_pack_conv3x3_winograd4_f64) comes straight from L2 in MFMA-A operand order, the V fragment is ONE ds_read_b128 from
//     the transformed patch in LDS -- 2 memory instructions per 4 MFMAs and no transform arithmetic in these waves;
//   * waves 12-15 ("producers", one per SIMD) keep the pipeline fed: they issue the LDS-DMA fill of the raw 6x6 patches
//     (8 channels per stage, `buffer_load_dwordx4 ... lds`; pixels outside the image arrive as the buffer unit's zeros = the
//     convolution's padding) two stages ahead and compute V = B^T d B one stage ahead: a lane owns one (tile, channel), reads
//     its 36 patch values (ds_read_b32, conflict-free through a source-side XOR swizzle of the DMA), runs the two 6-point
//     transform passes in registers (12 FMAs per 6 values) and writes the 36 frequencies in the consumers' operand layout.
//     F(2x2)'s scheme (every MFMA wave transforms its own frequency row) would cost 4 LDS reads per V value here: LDS-bound.
//   * one s_barrier per 8-channel stage (12 MFMAs = 768 matrix-pipe cycles per consumer wave, three consumers per SIMD).
//   * epilogue: the consumers fold the part of A^T . A that lies inside their three frequencies and hand all 32 channels over
//     through the (now idle) stage buffers in ONE round; all 16 waves finish A^T . A, add bias, activate and store.
//
// Numerics: every product and sum is fp32 in a fixed order that depends on the layer only (batch-size invariant,
// run-to-run deterministic).  The transforms of F(4x4) have larger constants than F(2x2)'s: the per-layer error against an fp64
// convolution is ~20x larger (tools/winograd_f43_parity.py: 4e-5 at output scale 4) while the end-to-end vertex error of
// hands_light stays at 1e-7 m.  HandOccNet keeps F(2x2) (DESIGN.md "Conditioning note").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "hands_hip.h"
#include "common.h"

#ifdef W4_PROF     // dev build only (tools/prof_wino4.py): s_memtime stamps of wave 0 (consumer) and wave 12 (producer) of the first 512 workgroups
__device__ unsigned long long g_w4prof[512 * 2 * 64];
extern "C" int hands_debug_w4prof(void* dst, int clear) {
  hipDeviceSynchronize();
  if (dst) hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_w4prof), sizeof(unsigned long long) * 512 * 2 * 64);
  if (clear) { static unsigned long long z[512 * 2 * 64]; hipMemcpyToSymbol(HIP_SYMBOL(g_w4prof), z, sizeof(z)); }
  return 0;
}
#define W4_STAMP(ROLE, IDX)                                                                          \
  do {                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (lane == 0 && blockIdx.x < 512 && (IDX) < 64 && (wave == 0 || wave == 12))                   \
      g_w4prof[(blockIdx.x * 2 + (ROLE)) * 64 + (IDX)] = __builtin_amdgcn_s_memtime();              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#else
#define W4_STAMP(ROLE, IDX) do {} while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

struct Wino4Args {
  const float* __restrict__ in;
  const float* __restrict__ u;      // [Cout/32][Cin/8][f 36][lane 64][4]
  const float* __restrict__ bias;
  float* out;
  int B, H, W, Cin, Cout, nh, nw;
  int in_ps, out_ps, act;
  int nblk_m, nblk_n, nseg;
  int band;                         // channel blocks per band of the workgroup order (w4_launch_nob)
  int rows;                         // B * nh flattened tile rows
  uint32_t nh_mul, nh_sh;           // magic number: x / nh
};

__device__ __forceinline__ int w4_fastdiv(int x, uint32_t mul, uint32_t sh) {
  return mul == 0 ? x : (int)(__umulhi((uint32_t)x, mul) >> sh);
}

__device__ __forceinline__ int w4_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float4 w4_f4(const u32x4& v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float w4_e(const float4& v, int t) { return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }

__device__ __forceinline__ void w4_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)dst, 16, voff, soff, 0, 0);
}

// v = B^T x for the 6-point input transform of F(4,3):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void w4_bt6(const float (&x)[6], float (&v)[6]) {
  const float a = fmaf(-4.f, x[2], x[4]);
  const float b = fmaf(-4.f, x[1], x[3]);
  const float c = x[4] - x[2];
  const float d = x[3] - x[1];
  v[0] = fmaf(4.f, x[0], fmaf(-5.f, x[2], x[4]));
  v[1] = a + b;
  v[2] = a - b;
  v[3] = fmaf(2.f, d, c);
  v[4] = fmaf(-2.f, d, c);
  v[5] = fmaf(4.f, x[1], fmaf(-5.f, x[3], x[5]));
}

template <int D, bool LINEAR>
struct W4Geom {
  static constexpr int PW = 4 * D + 2;                          // patch pixels per input row
  static constexpr int PWP = (PW + 3) / 4 * 4;                  // row pitch (pixels): whole 4-pixel groups (the swizzle's unit)
  static constexpr int NR = LINEAR ? (D - 1 + 32 + D - 1) / D : 32 / D;   // (virtual) tile rows a block can touch
  static constexpr int ROW_BYTES = PWP * 32;                    // one input row of a stage: 8 channels per pixel
  static constexpr int SLOTS = NR * 6 * PWP * 2;                // 16-byte slots of a stage's patch
  static constexpr int PIECES = (SLOTS + 63) / 64;              // DMA wave instructions per stage (64 slots = 1 KB each)
  static constexpr int NJ = (PIECES + 3) / 4;                   // per producer wave
  static constexpr int PATCH_BYTES = PIECES * 1024;
};

constexpr int W4_VBUF = 36 * 1024;                              // [f 36][tile 32][8 channels] fp32
constexpr int W4_PATCH_MAX = 36 * 1024;
constexpr long long W4_BAND_BYTES = 2560 * 1024;                // weights of one band of channel blocks: inside an XCD's 4 MB L2

// Second half of A^T M A for 32 output channels whose folded blocks sit in the LDS (8 channels per 36 KB region), all 16 waves:
// thread = (8-channel group rd, output column j, tile, channel quad).  Column j of Z = M A:
//   j0 = P0 + P0',  j1 = P1 + 2 P1',  j2 = P2 + 4 P0',  j3 = P1 + 8 P1' + P2'   (' = the hh = 1 wave), folded into the four output
// rows as it arrives (fixed order):  y0 = z0 + z1 + z2 + z3 + z4;  y1 = z1 - z2 + 2 z3 - 2 z4;  y2 = z1 + z2 + 4 z3 + 4 z4;
// y3 = z1 - z2 + 8 z3 - 8 z4 + z5;  then bias, activation, 16-byte NHWC stores.
template <int D, bool LINEAR, int VSH>
__device__ __forceinline__ void w4_output_pass(const Wino4Args& a, const char* lds, int wave, int lane, int nb, int R0, int s0, int tx0) {
  constexpr int VM = (1 << VSH) - 1;
  const int rd = wave >> 2, oj = wave & 3;
  const int otl = lane >> 1, ocq = lane & 1;
  const int oqq = s0 + otl;
  const int oRl = LINEAR ? oqq / D : otl / D, ocol = LINEAR ? oqq - oRl * D : otl - oRl * D;
  const int oV = R0 + oRl, oR = oV >> VSH, otx = tx0 + (oV & VM) * D + ocol;
  const int ob = w4_fastdiv(oR, a.nh_mul, a.nh_sh), oty = oR - ob * a.nh;
  if (!(oR < a.rows && otx < a.nw && 4 * otx + oj < a.W)) return;
  const int n_ch = nb * 32 + rd * 8 + ocq * 4;
  float* const o = a.out + (((size_t)ob * a.H + 4 * oty) * a.W + 4 * otx + oj) * (size_t)a.out_ps + n_ch;
  const char* sE = lds + rd * W4_VBUF + otl * 32 + ocq * 16;
  const float kq = oj == 0 ? 1.f : (oj == 1 ? 2.f : (oj == 2 ? 4.f : 8.f));
  const int pi = (oj == 0 ? 0 : (oj == 2 ? 2 : 1)) * 1024, qi = (oj == 0 || oj == 2 ? 3 : 4) * 1024;
  float4 y[4];
#pragma unroll
  for (int x6 = 0; x6 < 6; ++x6) {
    const char* eb = sE + x6 * 6 * 1024;
    const float4 p0 = *reinterpret_cast<const float4*>(eb + pi);
    const float4 q0 = *reinterpret_cast<const float4*>(eb + qi);
    float4 z = make_float4(fmaf(kq, q0.x, p0.x), fmaf(kq, q0.y, p0.y), fmaf(kq, q0.z, p0.z), fmaf(kq, q0.w, p0.w));
    if (oj == 3) {
      const float4 m5 = *reinterpret_cast<const float4*>(eb + 5 * 1024);
      z.x += m5.x; z.y += m5.y; z.z += m5.z; z.w += m5.w;
    }
#define W4_ACC(C)                                                                                    \
    if (x6 == 0) { y[0].C = z.C; }                                                                  \
    else if (x6 == 1) { y[0].C += z.C; y[1].C = z.C; y[2].C = z.C; y[3].C = z.C; }                  \
    else if (x6 == 2) { y[0].C += z.C; y[1].C -= z.C; y[2].C += z.C; y[3].C -= z.C; }               \
    else if (x6 == 3) { y[0].C += z.C; y[1].C = fmaf(2.f, z.C, y[1].C); y[2].C = fmaf(4.f, z.C, y[2].C); y[3].C = fmaf(8.f, z.C, y[3].C); } \
    else if (x6 == 4) { y[0].C += z.C; y[1].C = fmaf(-2.f, z.C, y[1].C); y[2].C = fmaf(4.f, z.C, y[2].C); y[3].C = fmaf(-8.f, z.C, y[3].C); } \
    else { y[3].C += z.C; }
    W4_ACC(x) W4_ACC(y) W4_ACC(z) W4_ACC(w)
#undef W4_ACC
  }
  const float4 bv = *reinterpret_cast<const float4*>(a.bias + n_ch);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 v = make_float4(y[i].x + bv.x, y[i].y + bv.y, y[i].z + bv.z, y[i].w + bv.w);
    if (a.act == HANDS_ACT_RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (a.act == HANDS_ACT_LEAKY_RELU) {
      v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
      v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
    }
    if (4 * oty + i < a.H) *reinterpret_cast<float4*>(o + (size_t)i * a.W * a.out_ps) = v;
  }
}

// NOB: 32-channel blocks per workgroup.  Two = every transformed patch value feeds 64 output channels: half the transform and
// fill work per MFMA -- one producer wave has ~5 issue slots beside each fp32 MFMA of its SIMD's three consumers, and at 32
// channels its ~220 instructions per stage did not fit beside their 36 MFMAs (stage 4.1 k cycles against 2.3 k of matrix-pipe time)
template <int D, bool LINEAR, int VSH, int NOB>   // VSH: a tile row is 1 << VSH "virtual rows" of D tiles (linear order)
__global__ void __launch_bounds__(1024, 1) conv_wino4_f32_kernel(Wino4Args a) {
  using G = W4Geom<D, LINEAR>;
  static_assert(G::PATCH_BYTES <= W4_PATCH_MAX, "patch buffer");
  static_assert(VSH == 0 || LINEAR, "virtual rows exist in the linear block order only");
  // [V 0][V 1][patch 0][patch 1], 36 KB each; the epilogue's exchange takes all four (one per 8 output channels)
  __shared__ __attribute__((aligned(1024))) char lds[4 * W4_VBUF];
  char* const sV = lds;
  char* const sP = lds + 2 * W4_VBUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 12;
  const int l31 = lane & 31, half = lane >> 5;
  constexpr int VM = (1 << VSH) - 1;

  // ---- which tiles, which 32 output channels (wave-uniform) -----------------------------------------------------------------
  // order (round 6): BANDS of `band` channel blocks whose weights fit an XCD's L2 together (<= 2.4 MB: all of layer1 / layer2, two
  // blocks of layer3, one of layer4); inside a band the channel block is the FASTEST index, so the workgroups that read the same
  // tile block's patches sit next to each other in one XCD's contiguous range and the activations cross the fabric once per band
  // instead of once per channel block (round 5 order = band 1: FETCH_SIZE showed Cout / 32 x the input per launch)
  const int wg = w4_xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
  int nbg, mb;
  {
    const int full = a.nblk_n / a.band, per = a.band * a.nblk_m;
    int bi = wg / per, r = wg - bi * per, bw = a.band;
    if (bi >= full) { bi = full; r = wg - full * per; bw = a.nblk_n - full * a.band; }     // the last, narrower band
    mb = r / bw;
    nbg = bi * a.band + (r - mb * bw);
  }
  const int nb = nbg * NOB;                                     // first 32-channel block of this workgroup
  int R0, s0, tx0;
  if constexpr (LINEAR) {
    const int t0 = mb * 32;
    R0 = t0 / D; s0 = t0 - R0 * D; tx0 = 0;
  } else {
    const int rb = mb / a.nseg, seg = mb - rb * a.nseg;
    R0 = rb * G::NR; s0 = 0; tx0 = seg * D;
  }
  const int R0r = R0 >> VSH;
  const int b_first = w4_fastdiv(R0r, a.nh_mul, a.nh_sh);
  const int ty_first = R0r - b_first * a.nh;
  const int nsteps = a.Cin >> 3;

  // The two roles are separate code paths with the SAME sequence of barriers (one per stage): their register live ranges
  // never overlap, so the kernel's allocation is max(consumer, producer), not the sum.
  if (producer) {
    // ---- producers: DMA source offsets (per lane, once), patch read addresses, then fill + transform ahead of the consumers ----
    W4_STAMP(1, 0);
    const int pw = wave - 12;
    __builtin_amdgcn_s_setprio(2);                              // one producer shares its SIMD's issue slots with three MFMA waves
    // LDS-DMA fill of the raw patches: producer wave pw issues pieces pw, pw + 4, ... of every stage.  Per lane, once: the source
    // offset of its 16-byte slot -- physical slot S = piece * 64 + lane (the DMA destination is lane-linear), whose logical
    // (pixel, quad) inside its 4-pixel group is S ^ (swz << 1): the swizzle the ds_read_b32 below need.
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.in) + (size_t)b_first * a.H * a.W * a.in_ps, 0, (int)0x80000000u, 0x00020000);
    int f_off[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
      const int i = j * 4 + pw;
      const int S = i * 64 + lane;
      const int r = S / (G::PWP * 2), within = S - r * (G::PWP * 2);
      const int Rl = r / 6, ar = r - Rl * 6;
      const int xg = within >> 3, sl = within & 7;
      const int swz = (D * Rl + xg) & 3;
      const int sl2 = sl ^ (swz << 1);
      const int x = 4 * xg + (sl2 >> 1), q = sl2 & 1;
      const int V = R0 + Rl, Rr = V >> VSH, pv = V & VM;
      const int t = ty_first + (Rr - R0r);
      const int db = w4_fastdiv(t, a.nh_mul, a.nh_sh), ty = t - db * a.nh;
      const int hy = 4 * ty - 1 + ar, wx = 4 * (tx0 + pv * D) - 1 + x;
      const bool ok = i < G::PIECES && x < G::PW && Rr < a.rows && (unsigned)hy < (unsigned)a.H && (unsigned)wx < (unsigned)a.W;
      f_off[j] = ok ? (((db * a.H + hy) * a.W + wx) * a.in_ps + q * 4) * 4 : (int)0x80000000u;
      // stage 0's piece leaves as soon as its offset exists (round 6: the fills used to wait for all NJ offsets, ~2 k cycles of a
      // 14 k prologue during which nothing else runs on the CU)
      if (i < G::PIECES) w4_dma16(x_rsrc, sP + i * 1024, f_off[j], 0);
    }
#define W4_PFILL(PB, G_)                                                                             \
  do {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < G::NJ; ++j) {                                             \
      if (j * 4 + pw < G::PIECES) w4_dma16(x_rsrc, sP + (PB) * W4_VBUF + (j * 4 + pw) * 1024, f_off[j], (G_) * 32); \
    }                                                                                               \
  } while (0)
    if (nsteps > 1) W4_PFILL(1, 1);
    // this lane's (tile, channel) of the transform: tile pw * 8 + (lane >> 3), channel lane & 7
    int tb[6];
    {
      const int tl = pw * 8 + (lane >> 3), c = lane & 7;
      const int qq = s0 + tl;
      const int Rl = LINEAR ? qq / D : tl / D, col = LINEAR ? qq - Rl * D : tl - Rl * D;
      const int swz0 = (D * Rl + col) & 3, swz1 = (swz0 + 1) & 3;
      const int q = c >> 2, w = c & 3;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int grp8 = b < 4 ? col : col + 1;
        const int sl = (2 * (b & 3) + q) ^ ((b < 4 ? swz0 : swz1) << 1);
        tb[b] = Rl * 6 * G::ROW_BYTES + (grp8 * 8 + sl) * 16 + w * 4;
      }
    }
    const int vw_off = pw * 256 + lane * 4;                     // V[f][tile][channel]: + f * 1024

    // V = B^T d B of this lane's (tile, channel): patch buffer PB -> V buffer VB.  All 36 reads are issued before the first
    // use (a wave that waits for each column's reads in turn spends 6 LDS latencies per stage).
#define W4_TRANSFORM(PB, VB, NEXT_FILL)                                                              \
  do {                                                                                              \
    const char* pp = sP + (PB) * W4_VBUF;                                                           \
    float xx[6][6];                                                                                 \
    _Pragma("unroll") for (int b = 0; b < 6; ++b)                                                   \
      _Pragma("unroll") for (int ar = 0; ar < 6; ++ar)                                              \
        xx[ar][b] = *reinterpret_cast<const float*>(pp + tb[b] + ar * G::ROW_BYTES);                \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    NEXT_FILL;                                  /* the next DMA pieces go out while the 36 reads are in flight */ \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    float tt[6][6];                                                                                 \
    _Pragma("unroll") for (int b = 0; b < 6; ++b) {                                                 \
      float x[6], v[6];                                                                             \
      _Pragma("unroll") for (int ar = 0; ar < 6; ++ar) x[ar] = xx[ar][b];                           \
      w4_bt6(x, v);                                                                                 \
      _Pragma("unroll") for (int k = 0; k < 6; ++k) tt[k][b] = v[k];                                \
    }                                                                                               \
    char* vp = sV + (VB) * W4_VBUF + vw_off;                                                        \
    _Pragma("unroll") for (int k = 0; k < 6; ++k) {                                                 \
      float v[6];                                                                                   \
      w4_bt6(tt[k], v);                                                                             \
      _Pragma("unroll") for (int n = 0; n < 6; ++n) *reinterpret_cast<float*>(vp + (k * 6 + n) * 1024) = v[n]; \
    }                                                                                               \
  } while (0)

    __syncthreads();                                           // stage 0 has landed (the consumers' DMA)
    W4_TRANSFORM(0, 0, );
    __syncthreads();
    W4_STAMP(1, 1);
    for (int g = 0; g < nsteps; ++g) {
      W4_STAMP(1, 2 + 3 * g);
#if !(defined(W4_ABL) && W4_ABL == 2)       // timing-only ablation (tools/build_variant.sh, EXTRA_FLAGS=-DW4_ABL=n): 2 = no transform in the loop
#if defined(W4_ABL) && W4_ABL == 5           // 5 = fill + transform in every other pair of stages only: the producer load per MFMA that
      if (g + 1 < nsteps && (g & 2) == 0) { //     64 output channels per workgroup would have
#else
      if (g + 1 < nsteps) {
#endif
        // (the patch buffer of stage g was transformed during stage g - 1: stage g + 2 may land in it)
        if ((g & 1) == 0) W4_TRANSFORM(1, 1, if (g + 2 < nsteps) W4_PFILL(0, g + 2)); else W4_TRANSFORM(0, 0, if (g + 2 < nsteps) W4_PFILL(1, g + 2));
      }
#endif
      W4_STAMP(1, 3 + 3 * g);
      __syncthreads();
      W4_STAMP(1, 4 + 3 * g);
    }
#undef W4_TRANSFORM
#undef W4_PFILL
  }
  if (!producer) {
    // ---- consumers -----------------------------------------------------------------------------------------------------------
    W4_STAMP(0, 0);
    const int xi = wave >> 1, hh = wave & 1;
    const int f0 = xi * 6 + 3 * hh;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u) + (size_t)nb * nsteps * 9216, 0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u) + (size_t)(nb + NOB - 1) * nsteps * 9216, 0, (int)0x80000000u, 0x00020000);
    const int w_off = f0 * 1024 + lane * 16;
    const int v_off = f0 * 1024 + l31 * 32 + half * 16;         // V fragment / exchange block of frequency f0 (+ n * 1024)
    f32x16 acc[3][NOB];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int o = 0; o < NOB; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][o][r] = 0.f;

    // Operands per FREQUENCY, two register sets: while the 4 NOB MFMAs of frequency n run, the V fragment (LDS) and the weight
    // fragments (L2) of the next frequency -- of the next stage behind the last one -- are on their way.  (NOB = 2: 96 accumulator
    // registers + 2 x 12 operand registers: a per-stage set would not fit the 128 a 16-wave workgroup leaves each wave.)
    float4 wq[2][NOB], vq[2];
#define W4_LOADOP(SET, N, STEP, VP)                                                                  \
  do {                                                                                              \
    vq[SET] = *reinterpret_cast<const float4*>((VP) + (N) * 1024);                                  \
    wq[SET][0] = w4_f4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off + (N) * 1024, (STEP) * 36864, 0)); \
    if (NOB > 1) wq[SET][NOB - 1] = w4_f4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc1, w_off + (N) * 1024, (STEP) * 36864, 0)); \
  } while (0)
#define W4_LOADW(SET, N, STEP)                                                                       \
  do {                                                                                              \
    wq[SET][0] = w4_f4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off + (N) * 1024, (STEP) * 36864, 0)); \
    if (NOB > 1) wq[SET][NOB - 1] = w4_f4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc1, w_off + (N) * 1024, (STEP) * 36864, 0)); \
  } while (0)
#if defined(W4_ABL) && W4_ABL == 4           // 4 = one MFMA of four
#define W4_NMFMA 1
#else
#define W4_NMFMA 4
#endif
#define W4_MFMAS(SET, N)                                                                             \
  do {                                                                                              \
    _Pragma("unroll") for (int o = 0; o < NOB; ++o)                                                 \
      _Pragma("unroll") for (int t = 0; t < W4_NMFMA; ++t)                                          \
        acc[N][o] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4_e(wq[SET][o], t), w4_e(vq[SET], t), acc[N][o], 0, 0, 0); \
  } while (0)

    // A consumer's barrier is a bare s_barrier: its weight fragments stay in flight across it (__syncthreads() would wait for
    // them -- the last one is requested right before the barrier); only its LDS reads are drained.
#define W4_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
    if constexpr (NOB == 2) {
      // 64 output channels per workgroup (round 6): 96 accumulator registers leave room for TWO weight-fragment sets (2 x 8
      // registers: the next frequency's weights are requested from L2 under the current one's 8 MFMAs) and ONE V fragment (4
      // registers, read from LDS right before its MFMAs: ~100 cycles that the SIMD's two other consumer waves cover).  Per stage
      // the matrix pipe works twice as long for the same fill + transform: the producers stop being the stage's critical path.
#define W4_V2(N, VP) do { vq[0] = *reinterpret_cast<const float4*>((VP) + (N) * 1024); } while (0)
#define W4_MFMAS2(WSET, N)                                                                           \
  do {                                                                                              \
    _Pragma("unroll") for (int o = 0; o < NOB; ++o)                                                 \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                 \
        acc[N][o] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4_e(wq[WSET][o], t), w4_e(vq[0], t), acc[N][o], 0, 0, 0); \
  } while (0)
      // ONE stage body (the register allocator gives two unrolled copies different accumulator registers and spills between them):
      // frequency 0 and 2 use weight set 0, frequency 1 set 1.  The next stage's frequency-0 weights are requested into set 0 as
      // soon as frequency 2's MFMAs have issued (they read their operands at issue) and stay in flight across the barrier.
      W4_LOADW(0, 0, 0);                                       // frequency 0 of stage 0
      W4_BARRIER();                                            // stage 0 (and 1) of the patch has landed
      W4_BARRIER();                                            // V of stage 0 is written
      W4_STAMP(0, 1);
      for (int g = 0; g < nsteps; ++g) {
        const int gn = g + 1 < nsteps ? g + 1 : g;              // (the last stage re-loads a valid step: no branch)
        const char* vp = sV + (g & 1) * W4_VBUF + v_off;
        W4_V2(0, vp); W4_LOADW(1, 1, g);
        __builtin_amdgcn_sched_barrier(0);
        W4_MFMAS2(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        W4_LOADW(0, 2, g);                                      // set 0 is free: frequency 0's MFMAs have issued
        W4_V2(1, vp);
        __builtin_amdgcn_sched_barrier(0);
        W4_MFMAS2(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        W4_V2(2, vp);
        __builtin_amdgcn_sched_barrier(0);
        W4_MFMAS2(0, 2);
        __builtin_amdgcn_sched_barrier(0);
        W4_LOADW(0, 0, gn);                                     // next stage, frequency 0
        __builtin_amdgcn_sched_barrier(0);
        W4_STAMP(0, 2 + 3 * g);
        W4_BARRIER();
        W4_STAMP(0, 4 + 3 * g);
      }
#undef W4_MFMAS2
#undef W4_V2
    } else {
    W4_LOADW(0, 0, 0);                                         // frequency 0 of stage 0 (its V fragment follows the barriers)
    W4_BARRIER();                                              // stage 0 (and 1) of the patch has landed
    W4_BARRIER();                                              // V of stage 0 is written
    W4_STAMP(0, 1);
    vq[0] = *reinterpret_cast<const float4*>(sV + v_off);
    for (int g = 0; g < nsteps; ++g) {
      const int gn = g + 1 < nsteps ? g + 1 : g;                // (the last stage re-loads a valid step: no branch)
      // ONE code path for both buffer parities (a runtime address): with the stage body instantiated per parity hipcc gave the two
      // copies different accumulator registers and moved all of them across after every other stage, behind an s_nop that drains
      // the matrix pipe (even stages measured 3.2-4.8 k cycles against 1.5 k for odd ones)
      const char* vp = sV + (g & 1) * W4_VBUF + v_off;
      // frequency 0 sits in set 0 (weights requested during the previous stage, V fragment right after the barrier)
      W4_LOADOP(1, 1, g, vp);
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMAS(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      W4_LOADOP(0, 2, g, vp);
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMAS(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      W4_LOADW(1, 0, gn);                                       // next stage, frequency 0: weights now, V behind the barrier
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMAS(0, 2);
      __builtin_amdgcn_sched_barrier(0);
      W4_STAMP(0, 2 + 3 * g);
      W4_BARRIER();
      W4_STAMP(0, 4 + 3 * g);
      // set 1 holds the next stage's frequency-0 weights: rename by moving 4 NOB registers (the loads are in flight: the moves
      // wait for them -- so instead the roles of the sets alternate per stage through the unrolled pair below)
      {
        const char* vn = sV + ((g + 1) & 1) * W4_VBUF + v_off;
        vq[1] = *reinterpret_cast<const float4*>(vn);
      }
      if (++g >= nsteps) break;
      // ---- odd stage: the same with the sets' roles exchanged (frequency 0 in set 1)
      const int gn2 = g + 1 < nsteps ? g + 1 : g;
      const char* vp2 = sV + (g & 1) * W4_VBUF + v_off;
      W4_LOADOP(0, 1, g, vp2);
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMAS(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      W4_LOADOP(1, 2, g, vp2);
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMAS(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      W4_LOADW(0, 0, gn2);
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMAS(1, 2);
      __builtin_amdgcn_sched_barrier(0);
      W4_STAMP(0, 2 + 3 * g);
      W4_BARRIER();
      W4_STAMP(0, 4 + 3 * g);
      vq[0] = *reinterpret_cast<const float4*>(sV + ((g + 1) & 1) * W4_VBUF + v_off);
    }
    }
#undef W4_BARRIER
#undef W4_LOADOP
#undef W4_LOADW
#undef W4_MFMAS

    // ---- A^T M A, first half.  Wave (xi, hh) folds what A^T's columns need from ITS three frequencies:
    //      hh = 0 (nu 0 1 2): P0 = M0 + M1 + M2, P1 = M1 - M2, P2 = M1 + M2;   hh = 1 (nu 3 4 5): P0 = M3 + M4, P1 = M3 - M4, P2 = M5
    //      and hands 32 channels at a time over through the whole LDS (every stage buffer is idle now): 8 channels per 36 KB region,
    //      block (xi, hh, p) at frequency slot f0 + p.  Accumulator register r of a lane: channel 8 (r >> 2) + 4 half + (r & 3).
#define W4_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
#pragma unroll
    for (int o = 0; o < NOB; ++o) {
      // (fold and hand-over per 32-channel block: with both blocks folded up front the two sides of the hh branch held all 96
      //  accumulators and the allocator spilled 64 of them around the first output pass)
      if (hh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float s_ = acc[1][o][r] + acc[2][o][r], d_ = acc[1][o][r] - acc[2][o][r];
          acc[0][o][r] = acc[0][o][r] + s_; acc[1][o][r] = d_; acc[2][o][r] = s_;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float s_ = acc[0][o][r] + acc[1][o][r], d_ = acc[0][o][r] - acc[1][o][r];
          acc[0][o][r] = s_; acc[1][o][r] = d_;
        }
      }
      // the output pass of the previous 32 channels has read the LDS.  Bare barriers behind the LDS traffic only: __syncthreads()
      // would also wait for that pass's global stores to drain (vmcnt(0))
      if (o > 0) W4_LDS_BARRIER();
#pragma unroll
      for (int rd = 0; rd < 4; ++rd)
#pragma unroll
        for (int n = 0; n < 3; ++n)
          *reinterpret_cast<float4*>(lds + rd * W4_VBUF + v_off + n * 1024) =
              make_float4(acc[n][o][4 * rd + 0], acc[n][o][4 * rd + 1], acc[n][o][4 * rd + 2], acc[n][o][4 * rd + 3]);
      if (NOB > 1) W4_LDS_BARRIER(); else __syncthreads();
      W4_STAMP(0, 60);
      w4_output_pass<D, LINEAR, VSH>(a, lds, wave, lane, nb + o, R0, s0, tx0);
    }
    W4_STAMP(0, 61);
    return;
  }
  // ---- producers join the output passes (they hold no accumulators; the barrier sequence matches the consumers') --------------
#pragma unroll
  for (int o = 0; o < NOB; ++o) {
    if (o > 0) W4_LDS_BARRIER();
    if (NOB > 1) W4_LDS_BARRIER(); else __syncthreads();
    W4_STAMP(1, 60);
    w4_output_pass<D, LINEAR, VSH>(a, lds, wave, lane, nb + o, R0, s0, tx0);
  }
  W4_STAMP(1, 61);
}

}  // namespace

static void w4_magic(int d, uint32_t& mul, uint32_t& sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  int l = 0;
  while ((1LL << l) < d) ++l;
  mul = (uint32_t)(((1ULL << (31 + l)) + (uint64_t)d - 1) / (uint64_t)d);
  sh = (uint32_t)(l - 1);
}

// tile-block count of the geometry the launch takes for this map width
static long long w4_blocks(long long rows, int nw) {
  if (nw == 7 || nw == 14) return (rows * nw + 31) / 32;
  if (nw <= 2) return (rows + 15) / 16;
  return (rows + 7) / 8 * ((nw + 3) / 4);
}

template <int D, bool LINEAR, int VSH, int NOB>
static int w4_launch_nob(Wino4Args& a, hipStream_t stream) {
  using G = W4Geom<D, LINEAR>;
  const long long rows = a.rows;
  long long nblk_m;
  if (LINEAR) {
    nblk_m = (rows * (D << VSH) + 31) / 32;
    a.nseg = 1;
  } else {
    a.nseg = (a.nw + D - 1) / D;
    nblk_m = (rows + G::NR - 1) / G::NR * a.nseg;
  }
  a.nblk_n = a.Cout / (32 * NOB);                                // workgroups along the channels
  {
    // a workgroup's weights: Cin / 8 stages x 36 KB per 32 channels
    const long long wbytes = (long long)(a.Cin / 8) * 36864 * NOB;
    long long band = W4_BAND_BYTES / wbytes;
#ifdef HANDS_W4_BAND_ENV      // dev build only (tools/experiments/w4_band.py: EXTRA_FLAGS=-DHANDS_W4_BAND_ENV tools/build_variant.sh w4env):
    const char* env = getenv("HANDS_W4_BAND");                   // HANDS_W4_BAND=1 is the round-5 order; the shipped library reads no environment
    if (env && atoi(env) > 0) band = atoi(env);
#endif
    a.band = (int)(band < 1 ? 1 : (band > a.nblk_n ? a.nblk_n : band));
  }
  if (nblk_m <= 0 || nblk_m * a.nblk_n > 0x7fffffffLL) return HANDS_EINVAL;
  a.nblk_m = (int)nblk_m;
  const long long nwg = nblk_m * a.nblk_n;        // one workgroup (16 waves, the whole LDS) per CU at a time
  w4_magic(a.nh, a.nh_mul, a.nh_sh);
  const long long imgs = G::NR / a.nh + 2;
  if (imgs * a.H * a.W * a.in_ps * 4 >= 0x7fffffffLL) return HANDS_EINVAL;
  hipLaunchKernelGGL((conv_wino4_f32_kernel<D, LINEAR, VSH, NOB>), dim3((unsigned)nwg), dim3(1024), 0, stream, a);
  return (int)hipGetLastError();
}

template <int D, bool LINEAR, int VSH = 0>
static int w4_launch(Wino4Args& a, hipStream_t stream) {
  // NOB = 2 (64 output channels per workgroup: half the transform / fill work per MFMA) where the channel count allows it (round 6:
  // 96 accumulator registers + two weight-fragment sets + one V fragment = 124 VGPRs, no spill in the k-loop); 32 otherwise.
#ifndef HANDS_W4_NOB1        // (dev switch: -DHANDS_W4_NOB1 builds the round-5 form for A/B)
  if (a.Cout % 64 == 0) return w4_launch_nob<D, LINEAR, VSH, 2>(a, stream);
#endif
  return w4_launch_nob<D, LINEAR, VSH, 1>(a, stream);
}

static bool w4_ok(const hands_conv_desc* d) {
  if (d->B <= 0 || d->H <= 0 || d->W <= 0) return false;
  const int act = d->act & HANDS_ACT_MASK;
  const long long nh = (d->H + 3) / 4, nw = (d->W + 3) / 4, rows = (long long)d->B * nh;
  if (rows >= 0x7fffff00LL || rows * nw >= 0x7fffff00LL) return false;
  if (w4_blocks(rows, (int)nw) * (d->Cout / 32) > 0x7fffffffLL) return false;
  const long long ps = d->in_pix_stride > d->out_pix_stride ? d->in_pix_stride : d->out_pix_stride;
  if ((16 / nh + 2) * d->H * d->W * ps * 4 >= 0x7fffffffLL) return false;
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->H && d->Wo == d->W && d->Cin >= 16 &&
         d->Cin % 8 == 0 && d->Cout >= 32 && d->Cout % 32 == 0 && d->in_pix_stride >= d->Cin && d->out_pix_stride >= d->Cout &&
         d->in_pix_stride % 4 == 0 && d->out_pix_stride % 4 == 0 &&
         (act == HANDS_ACT_NONE || act == HANDS_ACT_RELU || act == HANDS_ACT_LEAKY_RELU) && !(d->act & HANDS_MATH_BF16X3);
}

extern "C" int hands_conv3x3_winograd4_supported(const hands_conv_desc* d) { return d && w4_ok(d) ? 1 : 0; }

// Multiply-accumulates the matrix cores execute on this route (idle tile lanes included): 36 frequencies x 32-tile blocks x
// Cin x Cout.  The algorithmic count of the layer is 9 * H * W * Cin * Cout per image.
extern "C" long long hands_conv3x3_winograd4_executed_macs(const hands_conv_desc* d) {
  if (!d || !w4_ok(d)) return 0;
  const long long nh = (d->H + 3) / 4, nw = (d->W + 3) / 4;
  return 36LL * w4_blocks((long long)d->B * nh, (int)nw) * 32 * d->Cin * d->Cout;
}

extern "C" int hands_conv3x3_winograd4_f32(const hands_conv_desc* d, const float* in, const float* u_packed, const float* bias,
                                           float* out, hands_stream_t stream) {
  if (!d || !in || !u_packed || !bias || !out || !w4_ok(d)) return HANDS_EINVAL;
  if ((((uintptr_t)in) | ((uintptr_t)u_packed) | ((uintptr_t)bias) | ((uintptr_t)out)) & 15) return HANDS_EINVAL;
  Wino4Args a;
  a.in = in; a.u = u_packed; a.bias = bias; a.out = out;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
  a.nh = (d->H + 3) / 4; a.nw = (d->W + 3) / 4;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.act = d->act & HANDS_ACT_MASK;
  a.rows = d->B * a.nh;
  a.nblk_m = a.nblk_n = a.nseg = 0; a.band = 1;
  hipStream_t s = (hipStream_t)stream;
  if (a.nw == 7) return w4_launch<7, true>(a, s);
  if (a.nw == 14) return w4_launch<7, true, 1>(a, s);
  if (a.nw <= 2) return w4_launch<2, false>(a, s);
  return w4_launch<4, false>(a, s);
}
