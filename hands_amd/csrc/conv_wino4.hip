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
//   * epilogue per 32-channel block: the consumers fold the nu half of A^T . A that lies inside their three frequencies,
//     hand 8 channels at a time to waves 0-3 through the idle V buffer; those finish A^T . A, add bias, activate and store.
//
// Numerics: every product and sum is fp32 in a fixed order that depends on the layer only (batch-size invariant,
// run-to-run deterministic).  The transforms of F(4x4) have larger constants than F(2x2)'s: the per-layer error against an fp64
// convolution is ~20x larger (tools/winograd_f43_parity.py: 4e-5 at output scale 4) while the end-to-end vertex error of
// hands_light stays at 1e-7 m.  HandOccNet keeps F(2x2) (DESIGN.md "Conditioning note").
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include "hands_hip.h"
#include "common.h"

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
  int rows;                         // B * nh flattened tile rows
  int nbw, ngrp;                    // channel blocks per workgroup (divides nblk_n), groups = nblk_n / nbw
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

template <int D, bool LINEAR, int VSH = 0>   // VSH: a tile row is 1 << VSH "virtual rows" of D tiles (linear order)
__global__ void __launch_bounds__(1024, 1) conv_wino4_f32_kernel(Wino4Args a) {
  using G = W4Geom<D, LINEAR>;
  static_assert(G::PATCH_BYTES <= W4_PATCH_MAX, "patch buffer");
  static_assert(VSH == 0 || LINEAR, "virtual rows exist in the linear block order only");
  __shared__ __attribute__((aligned(1024))) char lds[2 * W4_VBUF + 2 * G::PATCH_BYTES];
  char* const sV = lds;
  char* const sP = lds + 2 * W4_VBUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 12;
  const int l31 = lane & 31, half = lane >> 5;
  constexpr int VM = (1 << VSH) - 1;

  // ---- which tiles, which channel blocks (wave-uniform) ------------------------------------------------------------------
  // order: all tile blocks of channel-block group 0, then group 1, ...: the weights of one group (<= 2.4 MB per 32 channels
  // at Cin = 512) stay in the XCD's L2 while the (small) input of a deep layer is re-read per group
  const int wg = w4_xcd_remap(blockIdx.x, a.nblk_m * a.ngrp);
  const int grp = wg / a.nblk_m, mb = wg - grp * a.nblk_m;
  const int nb0 = grp * a.nbw;
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
  const int nc8 = a.Cin >> 3;
  const int nsteps = nc8 * a.nbw;

  // The two roles are separate code paths with the SAME sequence of barriers (one per stage, eight per epilogue): their
  // register live ranges never overlap, so the kernel's allocation is max(consumer, producer), not the sum.
  if (producer) {
    // ---- producers: DMA source offsets (per lane, once), patch read addresses, then fill + transform ahead of the consumers ----
    const int pw = wave - 12;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.in) + (size_t)b_first * a.H * a.W * a.in_ps, 0, (int)0x80000000u, 0x00020000);
    int f_off[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
      const int i = j * 4 + pw;                                 // piece: LDS bytes [i * 1024, + 1024), lane-linear
      const int S = i * 64 + lane;                              // physical 16-byte slot
      const int r = S / (G::PWP * 2), within = S - r * (G::PWP * 2);
      const int Rl = r / 6, ar = r - Rl * 6;
      const int xg = within >> 3, sl = within & 7;
      const int swz = (D * Rl + xg) & 3;                        // the slot's logical (pixel, quad) inside its 4-pixel group
      const int sl2 = sl ^ (swz << 1);
      const int x = 4 * xg + (sl2 >> 1), q = sl2 & 1;
      const int V = R0 + Rl, Rr = V >> VSH, pv = V & VM;
      const int t = ty_first + (Rr - R0r);
      const int db = w4_fastdiv(t, a.nh_mul, a.nh_sh), ty = t - db * a.nh;
      const int hy = 4 * ty - 1 + ar, wx = 4 * (tx0 + pv * D) - 1 + x;
      const bool ok = i < G::PIECES && r < G::NR * 6 && x < G::PW && Rr < a.rows && (unsigned)hy < (unsigned)a.H &&
                      (unsigned)wx < (unsigned)a.W;
      f_off[j] = ok ? (((db * a.H + hy) * a.W + wx) * a.in_ps + q * 4) * 4 : (int)0x80000000u;
    }
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

    // DMA of stage g (channels 8 (g % nc8) ...) into patch buffer PB
#define W4_FILL(PB, G_)                                                                              \
  do {                                                                                              \
    const int chs = ((G_) % nc8) * 32;                                                              \
    _Pragma("unroll") for (int j = 0; j < G::NJ; ++j) {                                             \
      if (j * 4 + pw < G::PIECES) w4_dma16(x_rsrc, sP + (PB) * G::PATCH_BYTES + (j * 4 + pw) * 1024, f_off[j], chs); \
    }                                                                                               \
  } while (0)

    // V = B^T d B of this lane's (tile, channel): patch buffer PB -> V buffer VB
#define W4_TRANSFORM(PB, VB)                                                                         \
  do {                                                                                              \
    const char* pp = sP + (PB) * G::PATCH_BYTES;                                                    \
    float tt[6][6];                                                                                 \
    _Pragma("unroll") for (int b = 0; b < 6; ++b) {                                                 \
      float x[6], v[6];                                                                             \
      _Pragma("unroll") for (int ar = 0; ar < 6; ++ar)                                              \
        x[ar] = *reinterpret_cast<const float*>(pp + tb[b] + ar * G::ROW_BYTES);                    \
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

    W4_FILL(0, 0);
    __syncthreads();                                           // (the fence waits for this wave's DMA)
    if (nsteps > 1) W4_FILL(1, 1);
    W4_TRANSFORM(0, 0);
    __syncthreads();
    int ch = 0;
    for (int g = 0; g < nsteps; ++g) {
      const int par = g & 1;
      if (g + 2 < nsteps) W4_FILL(par, g + 2);
      if (g + 1 < nsteps) {
        if (par == 0) W4_TRANSFORM(1, 1); else W4_TRANSFORM(0, 0);
      }
      __syncthreads();
      if (++ch < nc8) continue;
      ch = 0;
#pragma unroll
      for (int rd = 0; rd < 8; ++rd) __syncthreads();          // the consumers' epilogue: four exchange rounds, two barriers each
    }
#undef W4_FILL
#undef W4_TRANSFORM
    return;
  }

  // ---- consumers -------------------------------------------------------------------------------------------------------------
  const int xi = wave >> 1, hh = wave & 1;
  const int f0 = xi * 6 + 3 * hh;
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.u) + (size_t)nb0 * nc8 * 9216, 0, (int)0x80000000u, 0x00020000);
  const int w_off = f0 * 1024 + lane * 16;
  const int v_off = f0 * 1024 + l31 * 32 + half * 16;           // V fragment / exchange block of frequency f0 (+ n * 1024)
  float4 wr[2][3];
  f32x16 acc[3];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

#define W4_LOADW(SET, STEP)                                                                          \
  do {                                                                                              \
    _Pragma("unroll") for (int n = 0; n < 3; ++n)                                                   \
      wr[SET][n] = w4_f4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off + n * 1024, (STEP) * 36864, 0)); \
  } while (0)
#define W4_MFMA(SET, PAR)                                                                            \
  do {                                                                                              \
    float4 vf[3];                                                                                   \
    _Pragma("unroll") for (int n = 0; n < 3; ++n) vf[n] = *reinterpret_cast<const float4*>(sV + (PAR) * W4_VBUF + v_off + n * 1024); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                   \
      _Pragma("unroll") for (int n = 0; n < 3; ++n)                                                 \
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4_e(wr[SET][n], t), w4_e(vf[n], t), acc[n], 0, 0, 0); \
  } while (0)

  W4_LOADW(0, 0);
  __syncthreads();
  __syncthreads();

  // ---- the output side (waves 0-3): thread = (output column j = wave, tile, channel quad of the round) ----------------------------
  const int oj = wave & 3;
  const int otl = lane >> 1, ocq = lane & 1;
  const int oqq = s0 + otl;
  const int oRl = LINEAR ? oqq / D : otl / D, ocol = LINEAR ? oqq - oRl * D : otl - oRl * D;
  const int oV = R0 + oRl, oR = oV >> VSH, otx = tx0 + (oV & VM) * D + ocol;
  const int ob = w4_fastdiv(oR, a.nh_mul, a.nh_sh), oty = oR - ob * a.nh;
  const bool o_ok = wave < 4 && oR < a.rows && otx < a.nw && 4 * otx + oj < a.W;
  float* const o_base = a.out + (((size_t)ob * a.H + 4 * oty) * a.W + 4 * otx + oj) * (size_t)a.out_ps + nb0 * 32 + ocq * 4;
  const int e_rd = otl * 32 + ocq * 16;

  int nbi = 0, ch = 0;
  for (int g = 0; g < nsteps; ++g) {
    const int par = g & 1;
    const int gn = g + 1 < nsteps ? g + 1 : g;                  // (the last stage re-loads a valid step: no branch)
    if (par == 0) {
      W4_LOADW(1, gn);
      __builtin_amdgcn_sched_barrier(0);                        // (hipcc otherwise sinks the loads to their first use)
      W4_MFMA(0, 0);
    } else {
      W4_LOADW(0, gn);
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA(1, 1);
    }
    __syncthreads();
    if (++ch < nc8) continue;
    ch = 0;

    // ---- end of a channel block: A^T M A.  Wave (xi, hh) folds what A^T's columns need from ITS three frequencies:
    //      hh = 0 (nu 0 1 2): P0 = M0 + M1 + M2, P1 = M1 - M2, P2 = M1 + M2;   hh = 1 (nu 3 4 5): P0 = M3 + M4, P1 = M3 - M4, P2 = M5
    //      column j of Z = M A:  j0 = P0 + P0',  j1 = P1 + 2 P1',  j2 = P2 + 4 P0',  j3 = P1 + 8 P1' + P2'   (' = the hh = 1 wave)
    //      8 channels per round through the V buffer this stage just released (block (xi, hh, p) at frequency slot f0 + p).
    //      Accumulator register r of a lane: channel 8 (r >> 2) + 4 half + (r & 3) of tile l31.
    char* sE = sV + par * W4_VBUF;
    if (hh == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = acc[1][r] + acc[2][r], d = acc[1][r] - acc[2][r];
        acc[0][r] = acc[0][r] + s; acc[1][r] = d; acc[2][r] = s;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = acc[0][r] + acc[1][r], d = acc[0][r] - acc[1][r];
        acc[0][r] = s; acc[1][r] = d;
      }
    }
    const int n_ch = (nb0 + nbi) * 32;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
#pragma unroll
      for (int n = 0; n < 3; ++n)
        *reinterpret_cast<float4*>(sE + v_off + n * 1024) =
            make_float4(acc[n][4 * rd + 0], acc[n][4 * rd + 1], acc[n][4 * rd + 2], acc[n][4 * rd + 3]);
      __syncthreads();
      if (o_ok) {
        float4 z[6];
#pragma unroll
        for (int x6 = 0; x6 < 6; ++x6) {
          const char* eb = sE + x6 * 6 * 1024 + e_rd;
          const float4 p0 = *reinterpret_cast<const float4*>(eb + (oj == 0 ? 0 : (oj == 2 ? 2 : 1)) * 1024);
          const float4 q0 = *reinterpret_cast<const float4*>(eb + (oj == 0 || oj == 2 ? 3 : 4) * 1024);
          const float k = oj == 0 ? 1.f : (oj == 1 ? 2.f : (oj == 2 ? 4.f : 8.f));
          z[x6] = make_float4(fmaf(k, q0.x, p0.x), fmaf(k, q0.y, p0.y), fmaf(k, q0.z, p0.z), fmaf(k, q0.w, p0.w));
          if (oj == 3) {
            const float4 m5 = *reinterpret_cast<const float4*>(eb + 5 * 1024);
            z[x6].x += m5.x; z[x6].y += m5.y; z[x6].z += m5.z; z[x6].w += m5.w;
          }
        }
        const float4 bv = *reinterpret_cast<const float4*>(a.bias + n_ch + rd * 8 + ocq * 4);
        float* o = o_base + (size_t)nbi * 32 + rd * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4 y;
#define W4_ROW(C)                                                                                    \
          {                                                                                         \
            const float s12 = z[1].C + z[2].C, d12 = z[1].C - z[2].C, s34 = z[3].C + z[4].C, d34 = z[3].C - z[4].C; \
            y.C = i == 0 ? (z[0].C + s12) + s34 : (i == 1 ? fmaf(2.f, d34, d12) : (i == 2 ? fmaf(4.f, s34, s12) : fmaf(8.f, d34, d12) + z[5].C)); \
          }
          W4_ROW(x) W4_ROW(y) W4_ROW(z) W4_ROW(w)
#undef W4_ROW
          y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w;
          if (a.act == HANDS_ACT_RELU) {
            y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
          } else if (a.act == HANDS_ACT_LEAKY_RELU) {
            y.x = y.x > 0.f ? y.x : 0.01f * y.x; y.y = y.y > 0.f ? y.y : 0.01f * y.y;
            y.z = y.z > 0.f ? y.z : 0.01f * y.z; y.w = y.w > 0.f ? y.w : 0.01f * y.w;
          }
          if (4 * oty + i < a.H) *reinterpret_cast<float4*>(o + (size_t)i * a.W * a.out_ps) = y;
        }
      }
      __syncthreads();                    // the next round (or the next stage's transform) overwrites the buffer
    }
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    ++nbi;
  }
#undef W4_LOADW
#undef W4_MFMA
}

}  // namespace

static int w4_device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

static int w4_env_nbw() {
  static const int v = [] { const char* e = getenv("HANDS_WINO4_NBW"); return e ? atoi(e) : 0; }();
  return v;
}

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

template <int D, bool LINEAR, int VSH = 0>
static int w4_launch(Wino4Args& a, hipStream_t stream) {
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
  a.nblk_n = a.Cout / 32;
  if (nblk_m <= 0 || nblk_m * a.nblk_n > 0x7fffffffLL) return HANDS_EINVAL;
  a.nblk_m = (int)nblk_m;
  // Channel blocks per workgroup (one workgroup per CU): ~3 k cycles of setup + first fill once, per channel block 2.4 k per
  // 8-channel stage and ~4.5 k of epilogue; fewer, longer workgroups quantise worse on the CU count.  A function of the launch
  // geometry only: the arithmetic and its order never depend on it.
  const long long slots = w4_device_cus();
  const double nc8 = a.Cin / 8;
  double best = 0.0;
  a.nbw = 1;
  for (int w = 1; w <= a.nblk_n; ++w) {
    if (a.nblk_n % w) continue;
    const long long wgs = nblk_m * (a.nblk_n / w);
    const double cost = (double)((wgs + slots - 1) / slots) * (3.0 + w * (2.4 * nc8 + 4.5));
    if (w == 1 || cost < 0.99 * best) { best = cost; a.nbw = w; }
  }
  if (const int w = w4_env_nbw(); w >= 1 && a.nblk_n % w == 0) a.nbw = w;
  a.ngrp = a.nblk_n / a.nbw;
  const long long nwg = nblk_m * a.ngrp;
  w4_magic(a.nh, a.nh_mul, a.nh_sh);
  const long long imgs = G::NR / a.nh + 2;
  if (imgs * a.H * a.W * a.in_ps * 4 >= 0x7fffffffLL) return HANDS_EINVAL;
  hipLaunchKernelGGL((conv_wino4_f32_kernel<D, LINEAR, VSH>), dim3((unsigned)nwg), dim3(1024), 0, stream, a);
  return (int)hipGetLastError();
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
  a.nblk_m = a.nblk_n = a.nseg = a.nbw = a.ngrp = 0;
  hipStream_t s = (hipStream_t)stream;
  if (a.nw == 7) return w4_launch<7, true>(a, s);
  if (a.nw == 14) return w4_launch<7, true, 1>(a, s);
  if (a.nw <= 2) return w4_launch<2, false>(a, s);
  return w4_launch<4, false>(a, s);
}
