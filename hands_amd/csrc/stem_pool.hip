// stem_pool.hip -- ResNet stem as ONE kernel: conv 7x7 / stride 2 / pad 3 (RGB0 input, 64 output
// channels, eval-BatchNorm folded) + activation + max-pool 3x3 / stride 2 / pad 1.
// Replaces conv1 -> bn1 -> relu -> maxpool of src/nets/backbone/resnet.py:264-268 (and the LeakyReLU
// variant of src/models/handoccnet_light/backbone.py:44-47).
//
// Why: the stem's 112x112x64 output is written and read back once just to be pooled (1.6 GB per 256
// images).  The GEMM itself is already efficient in conv_igemm (110 TFLOP/s of executed FLOPs; only
// 62 % of them are useful: RGB0 pads K 147 -> 196 -> 208), so the gain is the saved round trip:
// 1.00 -> 0.92 ms per 256 images.  Here a workgroup owns a 7 x 8 tile of POOLED pixels:
//   * the 35 x 39 input patch it needs is loaded once into LDS (16 B = one RGB0 pixel = one filter tap);
//   * the 15 x 17 = 255 convolution pixels under the pooled tile are the M rows of a 256 x 64 GEMM;
//     each of the 4 waves owns 64 of them (2x2 v_mfma_f32_32x32x2_f32 blocks) and walks the 13 k-steps
//     with NO barrier: activation fragments are gathered straight from the patch (even / odd column
//     planes keep the ds_read_b128 conflict-free), weight fragments come straight from L2 in MFMA
//     operand order (53 KB, shared by every workgroup; a copy in LDS measured 4 % slower);
//   * the convolution tile goes through LDS once (16 channels at a time) and is pooled there; only
//     the pooled 56 x 56 x 64 map is written to HBM.  bias and the (monotonic) activation are applied
//     after the max: max_i(a_i) + b == max_i(a_i + b) exactly.
// k order and the fp32 FMA chain per output are those of conv_igemm's stem mode, so results are
// bit-identical to the unfused pair of kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PT_H = 7, PT_W = 8;                 // pooled tile
constexpr int CT_H = 2 * PT_H + 1, CT_W = 2 * PT_W + 1;   // 15 x 17 convolution pixels (255 <= 256)
constexpr int IP_H = 2 * CT_H + 5, IP_W = 2 * CT_W + 5;   // 35 x 39 input pixels
constexpr int IP_EVEN = (IP_W + 1) / 2;           // 20 even columns, then 19 odd ones
constexpr int IP_ROW = 40;                        // 16-byte slots per patch row
constexpr int KPAD = 208;                         // 49 taps x 4 channels, padded to 13 steps of 16
constexpr int ST_ROW = 20;                        // staging row: 16 channels + 4 pad floats (5 slots: odd)

__device__ __forceinline__ float f4e(const float4& v, int t) { return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }

__device__ __forceinline__ float act_scalar(float v, int act) {
  if (act == HANDS_ACT_RELU) return fmaxf(v, 0.f);
  if (act == HANDS_ACT_LEAKY_RELU) return v > 0.f ? v : 0.01f * v;
  return v;
}

__global__ void __launch_bounds__(256, 3) stem_pool_kernel(const float4* __restrict__ x4, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_x,
                                                           int tiles_per_img, int act) {
  __shared__ __attribute__((aligned(16))) float4 patch[IP_H * IP_ROW];        // 22.4 KB
  __shared__ __attribute__((aligned(16))) float stage[256 * ST_ROW];          // 20.5 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / tiles_per_img;
  const int t = blockIdx.x - b * tiles_per_img;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int cy0 = 2 * PT_H * ty - 1, cx0 = 2 * PT_W * tx - 1;      // first convolution pixel of the tile
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;                  // first input pixel of the patch

  // ---- input patch -> LDS (zero outside the image = the convolution's padding) -------------------
  const float4* img = x4 + (size_t)b * H * W;
  for (int i = tid; i < IP_H * IP_W; i += 256) {
    const int pr = i / IP_W, pc = i - pr * IP_W;
    const int gy = iy0 + pr, gx = ix0 + pc;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = img[(size_t)gy * W + gx];
    patch[pr * IP_ROW + (pc & 1) * IP_EVEN + (pc >> 1)] = v;
  }
  __syncthreads();

  // ---- 256 x 64 x 208 GEMM, 64 rows per wave ------------------------------------------------------
  const int half = lane >> 5;
  int prow[2], pcol[2];                           // patch coordinates of this lane's two pixels at tap (0,0)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int p = wave * 64 + j * 32 + (lane & 31);
    p = p < CT_H * CT_W ? p : 0;                  // the one row past 255: any valid pixel, never used
    const int cy = p / CT_W, cx = p - cy * CT_W;
    prow[j] = 2 * cy;
    pcol[j] = 2 * cx;
  }
  const float* wl = w + (size_t)(lane & 31) * KPAD + half * 4;    // + i*32*KPAD + 16*s + 8*kk
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < KPAD / 16; ++s) {
    float4 wf[2][2], xf[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      int tap = 4 * s + 2 * kk + half;            // this lane's filter tap in MFMA steps 4*kk .. 4*kk+3
      tap = tap < 49 ? tap : 48;                  // padded taps have zero weights: read any finite pixel
      const int kh = tap / 7, kw = tap - kh * 7;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = pcol[j] + kw;             // pcol is even: the parity of col is that of kw
        xf[j][kk] = patch[(prow[j] + kh) * IP_ROW + (kw & 1) * IP_EVEN + (col >> 1)];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
        wf[i][kk] = *reinterpret_cast<const float4*>(wl + (size_t)i * 32 * KPAD + 16 * s + 8 * kk);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf[i][kk], tt), f4e(xf[j][kk], tt), acc[i][j], 0, 0, 0);
  }

  // ---- pool through LDS, 16 channels per round (a 20 KB stage keeps 3 workgroups per CU) ------------
  // D layout: row (channel) = 8*q + 4*half + e, col (pixel) = lane & 31.
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = r >> 1, q0 = (r & 1) * 2;
    if (r) __syncthreads();                       // the previous round's readers are done with the stage
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* dst = stage + (wave * 64 + j * 32 + (lane & 31)) * ST_ROW + half * 4;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        *reinterpret_cast<float4*>(dst + q * 8) =
            make_float4(acc[i][j][(q0 + q) * 4 + 0], acc[i][j][(q0 + q) * 4 + 1], acc[i][j][(q0 + q) * 4 + 2],
                        acc[i][j][(q0 + q) * 4 + 3]);
    }
    __syncthreads();
    if (tid < PT_H * PT_W * 4) {                  // 56 pooled pixels x 4 groups of 4 channels
      const int g = tid & 3, pp = tid >> 2;
      const int py = pp / PT_W, px = pp - py * PT_W;
      const int gy = PT_H * ty + py, gx = PT_W * tx + px;   // pooled pixel in the image
      if (gy < Hp && gx < Wp) {
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int ly = 2 * py + dy;                          // row in the convolution tile
          if ((unsigned)(cy0 + ly) >= (unsigned)Hc) continue;  // max-pool padding
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int lx = 2 * px + dx;
            if ((unsigned)(cx0 + lx) >= (unsigned)Wc) continue;
            const float4 v = *reinterpret_cast<const float4*>(stage + (ly * CT_W + lx) * ST_ROW + g * 4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
          }
        }
        const int c = i * 32 + q0 * 8 + g * 4;
        const float4 bv = *reinterpret_cast<const float4*>(bias + c);
        float4 o;
        o.x = act_scalar(m.x + bv.x, act); o.y = act_scalar(m.y + bv.y, act);
        o.z = act_scalar(m.z + bv.z, act); o.w = act_scalar(m.w + bv.w, act);
        *reinterpret_cast<float4*>(out + (((size_t)b * Hp + gy) * Wp + gx) * 64 + c) = o;
      }
    }
  }
}


// ---- planar variant: reads the reference's NCHW image directly, K = 3 planes x 52 (49 taps + 3 pad) = 156 -> 160
// instead of 49 taps x RGB0 = 196 -> 208: 23 % fewer MFMAs, and the NCHW -> NHWC4 conversion launch disappears.
// A k-group of four consecutive k is four consecutive taps of ONE colour plane (52 % 4 == 0); their patch offsets
// come from a 160-entry table built once per workgroup, the four values are gathered with ds_read_b32.
constexpr int KPL = 52;                           // k per colour plane
constexpr int KPADP = 160;                        // 3 * 52 = 156, padded to 10 steps of 16
constexpr int PL_ROW = 40;                        // floats per patch row of one plane (20 even + 19 odd columns + 1)
constexpr int PL_SIZE = IP_H * PL_ROW;            // one plane of the patch

__global__ void __launch_bounds__(256, 3) stem_pool_planar_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_x,
                                                                  int tiles_per_img, int act) {
  __shared__ __attribute__((aligned(16))) float patch[3 * PL_SIZE];           // 16.8 KB
  __shared__ __attribute__((aligned(16))) float stage[256 * ST_ROW];          // 20.5 KB
  __shared__ __attribute__((aligned(16))) int tapoff[KPADP];                  // patch offset of k at tap (0,0) origin
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / tiles_per_img;
  const int t = blockIdx.x - b * tiles_per_img;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int cy0 = 2 * PT_H * ty - 1, cx0 = 2 * PT_W * tx - 1;      // first convolution pixel of the tile
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;                  // first input pixel of the patch

  if (tid < KPADP) {
    const int c = tid / KPL, tp = tid - c * KPL;
    const int tap = tp < 49 ? tp : 48;            // padded taps have zero weights: any finite value will do
    const int kh = tap / 7, kw = tap - kh * 7;
    tapoff[tid] = (c < 3 ? c : 2) * PL_SIZE + kh * PL_ROW + (kw & 1) * IP_EVEN + (kw >> 1);
  }
  // ---- input patch -> LDS, one plane after the other (zero outside the image = the convolution's padding) ------
  // thread = (patch column, row group): 27 unconditional loads (three colour planes x nine rows; offset 0 where the
  // pixel lies outside the image, zeroed afterwards) are all in flight before the first LDS store, with one 32-bit
  // offset per row -- the flat predicated load-then-store loop over the 4095 patch elements spent ~45 vector
  // instructions per element on 64-bit index arithmetic and one memory latency per iteration: 22.6 of a
  // workgroup's 52.8 us (tools/stem_prof.py)
  const float* img = x + (size_t)b * 3 * H * W;
  {
    constexpr int ROWS_IT = (IP_H + 3) / 4;                        // 9 rows per thread: rg, rg + 4, ...
    const int pc = tid & 63, rg = tid >> 6;
    const int gx = ix0 + pc;
    const bool colok = pc < IP_W && (unsigned)gx < (unsigned)W;
    const int HW = H * W;
    const float* img1 = img + HW;
    const float* img2 = img + 2 * HW;
    float pv[3][ROWS_IT];
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < ROWS_IT; ++it) {
      const int gy = iy0 + rg + 4 * it;
      const bool ok = colok && rg + 4 * it < IP_H && (unsigned)gy < (unsigned)H;
      const int off = ok ? gy * W + gx : 0;
      pv[0][it] = img[off]; pv[1][it] = img1[off]; pv[2][it] = img2[off];
      okmask |= ok ? 1u << it : 0u;
    }
    if (pc < IP_W) {
      float* dst = patch + rg * PL_ROW + (pc & 1) * IP_EVEN + (pc >> 1);
#pragma unroll
      for (int it = 0; it < ROWS_IT; ++it) {
        if (rg + 4 * it < IP_H) {
          const bool ok = (okmask >> it) & 1u;
#pragma unroll
          for (int c = 0; c < 3; ++c) dst[c * PL_SIZE + 4 * it * PL_ROW] = ok ? pv[c][it] : 0.f;
        }
      }
    }
  }
  __syncthreads();

  // ---- 256 x 64 x 160 GEMM, 64 rows per wave --------------------------------------------------------------
  const int half = lane >> 5;
  int pbase[2];                                   // patch offset of this lane's two pixels at tap (0,0), plane 0
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int p = wave * 64 + j * 32 + (lane & 31);
    p = p < CT_H * CT_W ? p : 0;                  // the one row past 255: any valid pixel, never used
    const int cy = p / CT_W, cx = p - cy * CT_W;
    pbase[j] = 2 * cy * PL_ROW + cx;              // column 2*cx + kw -> parity plane (kw & 1), index cx + (kw >> 1)
  }
  const float* wl = w + (size_t)(lane & 31) * KPADP + half * 4;    // + i*32*KPADP + 16*s + 8*kk
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < KPADP / 16; ++s) {
    float4 wf[2][2], xf[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int4 to = *reinterpret_cast<const int4*>(&tapoff[16 * s + 8 * kk + 4 * half]);   // this lane's four k
#pragma unroll
      for (int j = 0; j < 2; ++j)
        xf[j][kk] = make_float4(patch[pbase[j] + to.x], patch[pbase[j] + to.y], patch[pbase[j] + to.z], patch[pbase[j] + to.w]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        wf[i][kk] = *reinterpret_cast<const float4*>(wl + (size_t)i * 32 * KPADP + 16 * s + 8 * kk);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf[i][kk], tt), f4e(xf[j][kk], tt), acc[i][j], 0, 0, 0);
  }

  // ---- pool through LDS, 16 channels per round (same max order as stem_pool_kernel).  Everything that does not
  // depend on the round -- which of the nine taps lie inside the convolution map, their LDS offsets, the bias values
  // of all four rounds -- is worked out once, before the first barrier
  const bool pooler = tid < PT_H * PT_W * 4;
  const int pg = tid & 3, pp = tid >> 2;
  const int ppy = pp / PT_W, ppx = pp - ppy * PT_W;
  const int pgy = PT_H * ty + ppy, pgx = PT_W * tx + ppx;
  const bool pool_ok = pooler && pgy < Hp && pgx < Wp;
  unsigned tapmask = 0;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
      if ((unsigned)(cy0 + 2 * ppy + dy) < (unsigned)Hc && (unsigned)(cx0 + 2 * ppx + dx) < (unsigned)Wc) tapmask |= 1u << (dy * 3 + dx);
  const float* pst = stage + ((2 * ppy) * CT_W + 2 * ppx) * ST_ROW + pg * 4;
  float4 pbias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    pbias[r] = pool_ok ? *reinterpret_cast<const float4*>(bias + (r >> 1) * 32 + (r & 1) * 16 + pg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float* pout = out + (((size_t)b * Hp + (pool_ok ? pgy : 0)) * Wp + (pool_ok ? pgx : 0)) * 64 + pg * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = r >> 1, q0 = (r & 1) * 2;
    if (r) __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* dst = stage + (wave * 64 + j * 32 + (lane & 31)) * ST_ROW + half * 4;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        *reinterpret_cast<float4*>(dst + q * 8) =
            make_float4(acc[i][j][(q0 + q) * 4 + 0], acc[i][j][(q0 + q) * 4 + 1], acc[i][j][(q0 + q) * 4 + 2],
                        acc[i][j][(q0 + q) * 4 + 3]);
    }
    __syncthreads();
    if (pool_ok) {
      float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float4 v = *reinterpret_cast<const float4*>(pst + (dy * CT_W + dx) * ST_ROW);
          if (tapmask & (1u << (dy * 3 + dx))) {
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
          }
        }
      float4 o;
      o.x = act_scalar(m.x + pbias[r].x, act); o.y = act_scalar(m.y + pbias[r].y, act);
      o.z = act_scalar(m.z + pbias[r].z, act); o.w = act_scalar(m.w + pbias[r].w, act);
      *reinterpret_cast<float4*>(pout + (r >> 1) * 32 + (r & 1) * 16) = o;
    }
  }
}

}  // namespace

extern "C" int hands_stem_conv_maxpool_nhwc_f32(const float* x4, const float* w_packed, const float* bias, float* out,
                                                int B, int H, int W, int act, hands_stream_t stream) {
  if (!x4 || !w_packed || !bias || !out || B <= 0 || H < 7 || W < 7) return HANDS_EINVAL;
  if (act != HANDS_ACT_NONE && act != HANDS_ACT_RELU && act != HANDS_ACT_LEAKY_RELU) return HANDS_EINVAL;
  const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;
  const int Hp = (Hc + 2 - 3) / 2 + 1, Wp = (Wc + 2 - 3) / 2 + 1;
  const int tiles_y = (Hp + PT_H - 1) / PT_H, tiles_x = (Wp + PT_W - 1) / PT_W;
  const long long nwg = (long long)B * tiles_y * tiles_x;
  if (nwg > 0x7fffffffLL) return HANDS_EINVAL;
  hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)x4, w_packed, bias, out, H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y * tiles_x, act);
  HANDS_LAUNCH_CHECK();
}

extern "C" int hands_stem_conv_maxpool_nchw_f32(const float* x_nchw, const float* w_planar, const float* bias, float* out,
                                                int B, int H, int W, int act, hands_stream_t stream) {
  if (!x_nchw || !w_planar || !bias || !out || B <= 0 || H < 7 || W < 7) return HANDS_EINVAL;
  if (act != HANDS_ACT_NONE && act != HANDS_ACT_RELU && act != HANDS_ACT_LEAKY_RELU) return HANDS_EINVAL;
  const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;
  const int Hp = (Hc + 2 - 3) / 2 + 1, Wp = (Wc + 2 - 3) / 2 + 1;
  const int tiles_y = (Hp + PT_H - 1) / PT_H, tiles_x = (Wp + PT_W - 1) / PT_W;
  const long long nwg = (long long)B * tiles_y * tiles_x;
  if (nwg > 0x7fffffffLL) return HANDS_EINVAL;
  hipLaunchKernelGGL(stem_pool_planar_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, x_nchw, w_planar, bias,
                     out, H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y * tiles_x, act);
  HANDS_LAUNCH_CHECK();
}
