// conv_wino.hip -- 3x3 / stride 1 / pad 1 NHWC convolution as Winograd F(2x2, 3x3) on the gfx950 fp32 matrix cores.
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray; d = 4x4 input patch, Y = 2x2 output pixels)
//
// 16 multiplications per 2x2 outputs and (cin, cout) pair instead of 36: the direct implicit GEMM (conv_igemm.hip)
// runs these layers at its k-loop ceiling (131-134 TFLOP/s), so the only way to make them faster is to execute
// fewer matrix-core FLOPs.  The 3x3 / stride-1 convolutions are 36 % of a hands_light forward.
//
// Per frequency f = (xi, nu) the layer is a GEMM   M_f[o, t] = sum_c U_f[o, c] * V_f[t, c]   (t = 2x2 output tile):
//   * U = G g G^T is computed ONCE on the host in fp64 (hands_pack_conv3x3_winograd_f64) and stored in MFMA-A operand
//     order: a wave's 16-byte-per-lane load IS its fragment (1 KB contiguous per instruction, no LDS, no shuffles);
//   * V = B^T d B is computed on the fly: the workgroup keeps the RAW input patch of its 32 tiles in LDS (16 channels
//     per stage, filled by LDS-DMA `buffer_load_dwordx4 ... lds`: no staging registers, out-of-image pixels arrive as
//     the buffer unit's zeros = the convolution's zero padding), wave xi combines the two patch rows B^T selects
//     (xi: rows (0,-2) (1,+2) (2,-1) (1,-3)), then the four column combinations nu -- 2 LDS reads + 2 adds per V value;
//   * wave xi accumulates its four nu in 4 x 16 accumulator registers (32 tiles x 32 output channels each), applies
//     the nu half of A^T . A in registers, and the xi half goes through LDS (32 KB) once per workgroup.
// Workgroup = 4 waves = 32 tiles (128 output pixels) x 32 output channels, 16 MFMAs (v_mfma_f32_32x32x2_f32) per
// wave and 8-channel step.  Tiles: flattened tile rows R = b * nh + ty; a block is NR rows x D columns (D = 4 or 8,
// "rect") or 32 consecutive tiles of the row-major order (D = nw = 7, "linear": 14x14 maps, no idle lanes).
//
// Numerics: every product and sum is fp32, in a fixed order that depends on the layer only (batch-size invariant,
// run-to-run deterministic).  Winograd re-associates the 3x3 sum, so results differ from the direct kernel by fp32
// rounding (tools/winograd_parity.py: the end-to-end vertex error against an fp64 forward is the same 1e-7 m as the
// direct algorithm's).  Replaces F.conv2d(3x3, s1, p1) + eval BatchNorm2d (folded) + ReLU of
// src/nets/backbone/resnet.py:140-142 (conv2 / bn2 / relu of every stride-1 Bottleneck).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

struct WinoArgs {
  const float* __restrict__ in;
  const float* __restrict__ u;      // [Cout/32][Cin/8][xi 4][nu 4][lane 64][4]
  const float* __restrict__ bias;
  float* out;
  int B, H, W, Cin, Cout, nh, nw;
  int in_ps, out_ps, act;
  int nblk_m, nblk_n, nseg;
  int rows;                         // B * nh flattened tile rows
};

__device__ __forceinline__ int wino_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float4 f4(const u32x4& v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float f4e(const float4& v, int t) { return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 fma4(float s, const float4& a, const float4& b) {
  return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));   // s = +-1: exact add / sub
}

// LDS-DMA: 64 lanes x 16 B from (descriptor base + voff + soff) to the wave-uniform LDS address dst + lane * 16.  An
// out-of-range voff (0x80000000) writes ZEROS (probed on gfx950: tools/wino_probe/lds_dma_oob.hip).  (A plain device
// function: with the builtin written inside the kernel template hipcc's host pass silently drops the kernel's stub.)
__device__ __forceinline__ void wino_dma16(__amdgpu_buffer_rsrc_t rsrc, float* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)dst, 16, voff, soff, 0, 0);
}

template <int D, bool LINEAR>
struct WinoGeom {
  static constexpr int PW = 2 * D + 2;                          // patch pixels per input row
  static constexpr int PWP = PW;                                // row pitch (pixels)
  static constexpr int RP = 4 * PWP + 1;                        // pitch of a tile row's 4 input rows: odd, so consecutive
                                                                // tile rows start in different bank classes
  static constexpr int NR = LINEAR ? (D - 1 + 32 + D - 1) / D : 32 / D;   // tile rows a block can touch
  static constexpr int SLOTS = NR * RP * 4;                     // 16-byte slots of one 16-channel stage
  static constexpr int NJW = (SLOTS + 63) / 64;                 // LDS-DMA wave instructions per stage
  static constexpr int NJ = (NJW + 3) / 4;                      // per wave
  static constexpr int BUF_FLOATS = NJW * 256;                  // 1 KB per wave instruction
};

#ifndef WINO_WAVES
#define WINO_WAVES 3
#endif
constexpr int ZBUF_FLOATS = 4 * 2 * 32 * 32;                    // epilogue exchange: [xi][j][tile 32][32 channels]

// One workgroup: 32 tiles x 32 output channels, all 16 frequencies, all input channels.
template <int D, bool LINEAR>
__global__ void __launch_bounds__(256, WINO_WAVES) conv_wino_f32_kernel(WinoArgs a) {
  using G = WinoGeom<D, LINEAR>;
  // (at least 41 KB: 3 workgroups per CU is what the ~160 registers allow anyway, and hipcc then schedules for that)
  constexpr int LDS_FLOATS = 2 * G::BUF_FLOATS > 10496 ? 2 * G::BUF_FLOATS : 10496;
  static_assert(LDS_FLOATS >= ZBUF_FLOATS, "the epilogue exchange reuses the patch buffers");
  __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = frequency row
  const int l31 = lane & 31, half = lane >> 5;

  const int ntiles = a.nblk_m * a.nblk_n;
  const int tile = wino_xcd_remap(blockIdx.x, ntiles);
  const int mb = tile / a.nblk_n, nb = tile - mb * a.nblk_n;    // n fastest: the channel blocks of one patch share an L2

  int R0, s0, tx0;                                              // (all 32-bit: B * nh * nw < 2^31 is checked by the host)
  if constexpr (LINEAR) {
    const int t0 = mb * 32;
    R0 = t0 / D; s0 = t0 - R0 * D; tx0 = 0;
  } else {
    const int rb = mb / a.nseg, seg = mb - rb * a.nseg;
    R0 = rb * G::NR; s0 = 0; tx0 = seg * D;
  }
  const int b_first = R0 / a.nh;                                // first image this block touches (wave-uniform)

  // ---- LDS-DMA fill assignment: slot = j * 256 + tid  ->  (tile row, input row a, patch pixel x, 16-byte slot) ----
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in) + (size_t)b_first * a.H * a.W * a.in_ps, 0, (int)0x80000000u, 0x00020000);
  int f_off[G::NJ];
#pragma unroll
  for (int j = 0; j < G::NJ; ++j) {
    const int slot = j * 256 + tid;
    const int P = slot >> 2, qs = slot & 3;
    const int Rl = P / G::RP, rem = P - Rl * G::RP;
    const int ar = rem / G::PWP, x = rem - ar * G::PWP;
    const int R = R0 + Rl;
    const int b = R / a.nh, ty = R - b * a.nh;
    const int hy = 2 * ty - 1 + ar, wx = 2 * tx0 - 1 + x;
    const bool ok = Rl < G::NR && ar < 4 && R < a.rows && (unsigned)hy < (unsigned)a.H && (unsigned)wx < (unsigned)a.W;
    const int q = qs ^ ((x >> 1) & 3);                          // source-side swizzle (the DMA destination is lane-linear)
    f_off[j] = ok ? ((((b - b_first) * a.H + hy) * a.W + wx) * a.in_ps + q * 4) * 4 : (int)0x80000000u;
  }
#define WINO_FILL(BUF_OFF, CH)                                                                      \
  do {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < G::NJ; ++j) {                                             \
      if ((j + 1) * 4 <= G::NJW || j * 4 + xi < G::NJW)          /* only the last one can be partial */ \
        wino_dma16(x_rsrc, lds + (BUF_OFF) + (j * 4 + xi) * 256, f_off[j], (CH) * 64);                 \
    }                                                                                               \
  } while (0)

  // ---- weights: MFMA-A fragments straight from L2, one 16-byte load per lane, frequency and 8-channel step ----
  const int nc8 = a.Cin >> 3;
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.u) + (size_t)nb * nc8 * 4096, 0, (int)0x80000000u, 0x00020000);
  const int w_off = (xi * 1024 + lane * 4) * 4;
  float4 wr[2][4];
#define WINO_LOADW(SET, C8)                                                                         \
  do {                                                                                              \
    _Pragma("unroll") for (int nu = 0; nu < 4; ++nu)                                                \
      wr[SET][nu] = f4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off + nu * 1024, (C8) * 16384, 0)); \
  } while (0)

  // ---- this lane's tile and its patch read addresses ---------------------------------------------------------------
  const int qq = s0 + l31;
  const int Rl = qq / D, col = qq - Rl * D;
  // rows B^T selects for frequency row xi: r = d[a1] + sg * d[a2]
  const int a1 = (xi == 0) ? 0 : (xi == 2 ? 2 : 1);
  const int a2 = (xi == 0) ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sg = (xi == 1) ? 1.f : -1.f;
  // byte address of (row a, pixel 2 col + b, quad 2 s + half): pixel (Rl RP + a PWP + 2 col + b) * 64 + ((2 s + half) ^ swz) * 16,
  // swz = (col + (b >> 1)) & 3
  const int pix0 = Rl * G::RP + 2 * col;
  int rd[2][2];       // [row a1 / a2][b >> 1], s = 0; s = 1 is the address ^ 32
#pragma unroll
  for (int bc = 0; bc < 2; ++bc) {
    const int sw = (col + bc) & 3;
    rd[0][bc] = (pix0 + a1 * G::PWP) * 64 + ((half ^ sw) * 16);
    rd[1][bc] = (pix0 + a2 * G::PWP) * 64 + ((half ^ sw) * 16);
  }
  const char* ldsb = reinterpret_cast<const char*>(lds);

  f32x16 acc[4];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;

#define WINO_STEP(BOFF, S, SET)                                                                     \
  do {                                                                                              \
    float4 rr[4];                                                                                   \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                 \
      const float4 d1 = *reinterpret_cast<const float4*>(ldsb + (((rd[0][b >> 1] + (BOFF)) ^ ((S) * 32)) + b * 64)); \
      const float4 d2 = *reinterpret_cast<const float4*>(ldsb + (((rd[1][b >> 1] + (BOFF)) ^ ((S) * 32)) + b * 64)); \
      rr[b] = fma4(sg, d2, d1);                                                                     \
    }                                                                                               \
    float4 v[4];                                                                                    \
    v[0] = sub4(rr[0], rr[2]); v[1] = add4(rr[1], rr[2]); v[2] = sub4(rr[2], rr[1]); v[3] = sub4(rr[1], rr[3]); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                 \
      _Pragma("unroll") for (int nu = 0; nu < 4; ++nu)                                              \
        acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wr[SET][nu], t), f4e(v[nu], t), acc[nu], 0, 0, 0); \
    }                                                                                               \
  } while (0)

  // stage ch lives in buffer ch & 1; its DMA was issued one stage ahead.  One barrier per stage (32 MFMAs per wave).
  const int nch = a.Cin >> 4;
  WINO_FILL(0, 0);
  WINO_LOADW(0, 0);
  __syncthreads();                       // (the compiler's fence waits for this wave's DMA)
  for (int ch = 0; ch < nch; ++ch) {
    const int boff = (ch & 1) * (G::BUF_FLOATS * 4);           // bytes (a multiple of 1024: the ^ 32 of step 1 is unaffected)
    const int foff = ((ch & 1) ^ 1) * G::BUF_FLOATS;
    if (ch + 1 < nch) WINO_FILL(foff, ch + 1);
    WINO_LOADW(1, 2 * ch + 1);
    WINO_STEP(boff, 0, 0);
    WINO_LOADW(0, ch + 1 < nch ? 2 * ch + 2 : 2 * ch + 1);     // (the last stage re-loads a valid step: no branch)
    WINO_STEP(boff, 1, 1);
    __syncthreads();
  }
#undef WINO_STEP
#undef WINO_LOADW
#undef WINO_FILL

  // ---- epilogue: nu half of A^T . A in registers, xi half through LDS, bias + activation, 16-byte NHWC stores -----
  // accumulator register r of a lane: channel 8 (r >> 2) + 4 half + (r & 3) of tile l31
  {
    float* sZ = lds;                     // [xi][j][tile][8 quads], quad cq of tile t at position cq ^ (t & 7)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 z0, z1;
      z0.x = (acc[0][4 * g + 0] + acc[1][4 * g + 0]) + acc[2][4 * g + 0];
      z0.y = (acc[0][4 * g + 1] + acc[1][4 * g + 1]) + acc[2][4 * g + 1];
      z0.z = (acc[0][4 * g + 2] + acc[1][4 * g + 2]) + acc[2][4 * g + 2];
      z0.w = (acc[0][4 * g + 3] + acc[1][4 * g + 3]) + acc[2][4 * g + 3];
      z1.x = (acc[1][4 * g + 0] - acc[2][4 * g + 0]) - acc[3][4 * g + 0];
      z1.y = (acc[1][4 * g + 1] - acc[2][4 * g + 1]) - acc[3][4 * g + 1];
      z1.z = (acc[1][4 * g + 2] - acc[2][4 * g + 2]) - acc[3][4 * g + 2];
      z1.w = (acc[1][4 * g + 3] - acc[2][4 * g + 3]) - acc[3][4 * g + 3];
      const int cq = 2 * g + half;
      const int pos = (l31 * 8 + (cq ^ (l31 & 7))) * 4;
      *reinterpret_cast<float4*>(sZ + (xi * 2 + 0) * 1024 + pos) = z0;
      *reinterpret_cast<float4*>(sZ + (xi * 2 + 1) * 1024 + pos) = z1;
    }
  }
  __syncthreads();
  {
    const float* sZ = lds;
    const int tl = tid >> 3, cq = tid & 7;           // output: tile tl, channels 4 cq .. 4 cq + 3 of this block's 32
    const int pos = (tl * 8 + (cq ^ (tl & 7))) * 4;
    float4 z[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int j = 0; j < 2; ++j) z[x][j] = *reinterpret_cast<const float4*>(sZ + (x * 2 + j) * 1024 + pos);
    const int q2 = s0 + tl;
    const int Rl2 = q2 / D, col2 = q2 - Rl2 * D;
    const int R = R0 + Rl2;
    const int tx = tx0 + col2;
    if (R < a.rows && tx < a.nw) {
      const int b = R / a.nh, ty = R - b * a.nh;
      const int n = nb * 32 + cq * 4;
      const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
      const int act = a.act;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oy = 2 * ty + i;
        if (oy >= a.H) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ox = 2 * tx + j;
          if (ox >= a.W) continue;
          float4 y = i == 0 ? add4(add4(z[0][j], z[1][j]), z[2][j]) : sub4(sub4(z[1][j], z[2][j]), z[3][j]);
          y = add4(y, bv);
          if (act == HANDS_ACT_RELU) {
            y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
          } else if (act == HANDS_ACT_LEAKY_RELU) {
            y.x = y.x > 0.f ? y.x : 0.01f * y.x; y.y = y.y > 0.f ? y.y : 0.01f * y.y;
            y.z = y.z > 0.f ? y.z : 0.01f * y.z; y.w = y.w > 0.f ? y.w : 0.01f * y.w;
          }
          *reinterpret_cast<float4*>(a.out + (((size_t)b * a.H + oy) * a.W + ox) * (size_t)a.out_ps + n) = y;
        }
      }
    }
  }
}

}  // namespace

template <int D, bool LINEAR>
static int wino_launch(WinoArgs& a, hipStream_t stream) {
  using G = WinoGeom<D, LINEAR>;
  const long long rows = a.rows;
  if (rows * a.nw >= 0x7fffff00LL) return HANDS_EINVAL;
  long long nblk_m;
  if (LINEAR) {
    nblk_m = (rows * D + 31) / 32;
    a.nseg = 1;
  } else {
    a.nseg = (a.nw + D - 1) / D;
    nblk_m = (rows + G::NR - 1) / G::NR * a.nseg;
  }
  a.nblk_n = a.Cout / 32;
  const long long nwg = nblk_m * a.nblk_n;
  if (nwg <= 0 || nwg > 0x7fffffffLL) return HANDS_EINVAL;
  a.nblk_m = (int)nblk_m;
  // 32-bit byte offsets from the first image a block touches
  const long long imgs = G::NR / a.nh + 2;
  if (imgs * a.H * a.W * a.in_ps * 4 >= 0x7fffffffLL) return HANDS_EINVAL;
  hipLaunchKernelGGL((conv_wino_f32_kernel<D, LINEAR>), dim3((unsigned)nwg), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

static bool wino_ok(const hands_conv_desc* d) {
  const int act = d->act & HANDS_ACT_MASK;
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->H && d->Wo == d->W && d->B > 0 && d->H > 0 &&
         d->W > 0 && d->Cin >= 16 && d->Cin % 16 == 0 && d->Cout >= 32 && d->Cout % 32 == 0 && d->in_pix_stride >= d->Cin &&
         d->out_pix_stride >= d->Cout && d->in_pix_stride % 4 == 0 && d->out_pix_stride % 4 == 0 &&
         (act == HANDS_ACT_NONE || act == HANDS_ACT_RELU || act == HANDS_ACT_LEAKY_RELU) && !(d->act & HANDS_MATH_BF16X3);
}

extern "C" int hands_conv3x3_winograd_supported(const hands_conv_desc* d) { return d && wino_ok(d) ? 1 : 0; }

extern "C" int hands_conv3x3_winograd_f32(const hands_conv_desc* d, const float* in, const float* u_packed, const float* bias,
                                          float* out, hands_stream_t stream) {
  if (!d || !in || !u_packed || !bias || !out || !wino_ok(d)) return HANDS_EINVAL;
  WinoArgs a;
  a.in = in; a.u = u_packed; a.bias = bias; a.out = out;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
  a.nh = (d->H + 1) / 2; a.nw = (d->W + 1) / 2;
  a.in_ps = d->in_pix_stride; a.out_ps = d->out_pix_stride; a.act = d->act & HANDS_ACT_MASK;
  if ((long long)d->B * a.nh >= 0x7fffff00LL) return HANDS_EINVAL;
  a.rows = d->B * a.nh;
  a.nblk_m = a.nblk_n = a.nseg = 0;
  hipStream_t s = (hipStream_t)stream;
  if (a.nw == 7) return wino_launch<7, true>(a, s);
  if (a.nw % 4 == 0 || a.nw < 8) return wino_launch<4, false>(a, s);
  return wino_launch<8, false>(a, s);
}
