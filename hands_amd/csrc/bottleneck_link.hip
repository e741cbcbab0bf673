// bottleneck_link.hip -- the seam between two ResNet bottlenecks of layer1 as ONE kernel (gfx950, fp32 MFMA):
//
//     out = relu(conv3_1x1(t2) + bn3 + identity)          resnet.py:146-154 of block i      (64 -> 256 channels)
//     t1' = relu(conv1_1x1(out) + bn1)                     resnet.py:137-139 of block i + 1  (256 -> 64 | 128)
//
// At 56x56 both layers are HBM-bound (the expand convolution moves 2.3 KB per pixel for 32 kFLOP): run separately,
// `out` is written by the first launch and read back by the second.  Here a workgroup owns 64 pixels x ALL 256
// channels of `out`: the tile is written to HBM once (it is the next block's identity) and stays in LDS as the B
// operand of the second product, so the 1 KB per pixel of re-read disappears and the second layer's MFMAs run
// under the first one's memory time.
//
//   phase 1  acc3[64 ch x 64 px per wave] = W3[256 x 64] . t2^T   -- wave w owns channels 64 w .. 64 w + 63 (its W3
//            slice comes from L2, half a slice at a time); the t2 operand is loaded straight into MFMA layout.  Everything
//            a tile reads from HBM (t2, the first identity rows) is requested under phase 3 of the previous tile
//   phase 2  accumulators -> LDS tile [64 px][256 ch] -> row-wise read-back: + bias + identity (1 KB contiguous per
//            wave instruction), ReLU, store `out`, and the final value back into the LDS tile
//   phase 3  acc1 = W1[C1 x 256] . out^T from the LDS tile (weights straight from L2 in operand order), + bias,
//            ReLU, store t1'
//
// Every output is the SAME chain of fp32 FMAs as in conv_igemm.hip (k ascending in steps of 16, inside a step
// the pairs (8 kk + t, 8 kk + 4 + t)), the same (acc + bias) + identity order and the same ReLU: the fused launch is
// bit-identical to the two separate launches (tests/test_gpu_parity.py::test_bottleneck_link_is_bit_identical).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float f4e(const float4& v, int t) { return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }

struct LinkArgs {
  const float* __restrict__ t2;    // [M][64]
  const float* __restrict__ w3;    // packed [256][64]
  const float* __restrict__ b3;    // [256]
  const float* __restrict__ res;   // [M][256] identity
  float* __restrict__ out;         // [M][256]
  const float* __restrict__ w1;    // packed [>= C1][256]
  const float* __restrict__ b1;    // [C1]
  float* __restrict__ t1;          // [M][C1]
  int ntiles;                      // M / 64
};

constexpr int LK_C3IN = 64, LK_C3OUT = 256, LK_PX = 64;
constexpr int LK_ROW = LK_C3OUT + 4;          // LDS row: 260 floats (65 x 16 B: odd -> ds_read/write_b128 conflict-free)

template <int C1>
__global__ void __launch_bounds__(256, 2) bottleneck_link_kernel(LinkArgs a) {
  static_assert(C1 == 64 || C1 == 128, "second product: 64 or 128 output channels");
  __shared__ __attribute__((aligned(16))) float sO[LK_PX * LK_ROW];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- per-workgroup constants ----------------------------------------------------------------------------
  const float4 bias3 = *reinterpret_cast<const float4*>(a.b3 + 4 * lane);       // phase 2: lane -> channels 4 lane .. + 3
  // second product: C1 = 64: wave -> (channel block wave >> 1, pixel block wave & 1); C1 = 128: wave -> channel block
  // wave, both pixel blocks (one weight fragment feeds two MFMAs)
  constexpr int NJB = C1 / 64;
  const int cb1 = C1 == 64 ? (wave >> 1) : wave;
  const int pj0 = C1 == 64 ? (wave & 1) : 0;
  float4 b1f[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) b1f[q] = *reinterpret_cast<const float4*>(a.b1 + cb1 * 32 + 8 * q + 4 * half);
  const float* w1p = a.w1 + (size_t)(cb1 * 32 + l31) * LK_C3OUT + 4 * half;
  // W3 rows 64 wave + 32 i + l31, k = 8 kk + 4 half .. + 3: re-read per tile (L2), half a slice (32 registers) at a time
  const float* w3p = a.w3 + (size_t)(64 * wave + l31) * LK_C3IN + 4 * half;

  // everything a tile reads from HBM is requested one phase ahead: the t2 operand (MFMA layout: rows m0 + 32 j + l31,
  // k = 8 kk + 4 half .. + 3), the first 8 of this wave's 16 identity rows, and the first half of the W3 slice
  float4 xf[2][8], rv[8], w3a[8];
#define LK_PREFETCH(TILE)                                                                                             \
  do {                                                                                                                 \
    const size_t pm0 = (size_t)(TILE) * LK_PX;                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                      \
      _Pragma("unroll") for (int kk = 0; kk < 8; ++kk)                                                                 \
        xf[j][kk] = *reinterpret_cast<const float4*>(a.t2 + (pm0 + 32 * j + l31) * LK_C3IN + 8 * kk + 4 * half);       \
    _Pragma("unroll") for (int r = 0; r < 8; ++r)                                                                      \
      rv[r] = *reinterpret_cast<const float4*>(a.res + (pm0 + 16 * wave + r) * LK_C3OUT + 4 * lane);                   \
    _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) w3a[kk] = *reinterpret_cast<const float4*>(w3p + 8 * kk);         \
  } while (0)

  int tile = blockIdx.x;
#ifdef HANDS_LINK_STAGGER
  // the second workgroup of a CU (dispatch fills every CU once before it doubles up) starts half a tile late, so that its
  // memory phase falls under the first one's MFMA phases for the rest of the (persistent) launch
  if (blockIdx.x >= gridDim.x / 2) {
#pragma unroll
    for (int i = 0; i < HANDS_LINK_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  if (tile < a.ntiles) LK_PREFETCH(tile);

  for (; tile < a.ntiles; tile += gridDim.x) {
    const size_t m0 = (size_t)tile * LK_PX;
    // ---- phase 1: acc3 = W3 slice . t2^T (K = 64), one 32-channel half of the slice at a time ------------------
    float4 w3b[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) w3b[kk] = *reinterpret_cast<const float4*>(w3p + (size_t)32 * LK_C3IN + 8 * kk);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 acc3[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc3[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(i ? w3b[kk] : w3a[kk], t), f4e(xf[j][kk], t), acc3[j], 0, 0, 0);
      if (i == 0) __syncthreads();              // the previous tile's phase 3 is done reading the LDS tile
      // accumulators -> LDS [px][ch]; D row = channel 8 q + 4 half + e, D col = pixel l31
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(sO + (32 * j + l31) * LK_ROW + 64 * wave + 32 * i + 8 * q + 4 * half) =
              make_float4(acc3[j][4 * q + 0], acc3[j][4 * q + 1], acc3[j][4 * q + 2], acc3[j][4 * q + 3]);
    }
    __syncthreads();
    // ---- phase 2: rows 16 wave .. 16 wave + 15, lane -> channels 4 lane .. + 3: identity loads and `out` stores
    //      are 1 KB contiguous per wave instruction ----------------------------------------------------------------
    {
      const float* rp = a.res + (m0 + 16 * wave) * LK_C3OUT + 4 * lane;
      float* op = a.out + (m0 + 16 * wave) * LK_C3OUT + 4 * lane;
      float* sp = sO + (16 * wave) * LK_ROW + 4 * lane;
      float4 rv2[8];                            // second batch of identity rows: in flight under the first batch
#pragma unroll
      for (int r = 0; r < 8; ++r) rv2[r] = *reinterpret_cast<const float4*>(rp + (size_t)(8 + r) * LK_C3OUT);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 idv = r < 8 ? rv[r & 7] : rv2[r & 7];
        const float4 t = *reinterpret_cast<const float4*>(sp + r * LK_ROW);
        float4 v;                               // (acc + bias) + identity, then ReLU: conv_igemm's epilogue order
        v.x = (t.x + bias3.x) + idv.x; v.y = (t.y + bias3.y) + idv.y;
        v.z = (t.z + bias3.z) + idv.z; v.w = (t.w + bias3.w) + idv.w;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<float4*>(op + (size_t)r * LK_C3OUT) = v;
        *reinterpret_cast<float4*>(sp + r * LK_ROW) = v;
      }
    }
    __syncthreads();
    // the next tile's HBM operands: in flight under phase 3
    if (tile + (int)gridDim.x < a.ntiles) LK_PREFETCH(tile + gridDim.x);
    // ---- phase 3: acc1 = W1 block . out^T (K = 256) from the LDS tile ----------------------------------------
    f32x16 acc1[NJB];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[jb][r] = 0.f;
    const float* xb = sO + (32 * pj0 + l31) * LK_ROW + 4 * half;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {           // ks = 2 kt + kk: 8 k per step
      const float4 wf = *reinterpret_cast<const float4*>(w1p + 8 * ks);
      float4 xb4[NJB];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) xb4[jb] = *reinterpret_cast<const float4*>(xb + jb * 32 * LK_ROW + 8 * ks);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb)
          acc1[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf, t), f4e(xb4[jb], t), acc1[jb], 0, 0, 0);
    }
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
      float* tp = a.t1 + (m0 + 32 * (pj0 + jb) + l31) * C1 + cb1 * 32 + 4 * half;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(acc1[jb][4 * q + 0] + b1f[q].x, acc1[jb][4 * q + 1] + b1f[q].y,
                               acc1[jb][4 * q + 2] + b1f[q].z, acc1[jb][4 * q + 3] + b1f[q].w);
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<float4*>(tp + 8 * q) = v;
      }
    }
  }
#undef LK_PREFETCH
}

int link_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus;
}

}  // namespace

extern "C" int hands_bottleneck_link_f32(const float* t2, const float* w3_packed, const float* bias3, const float* identity,
                                         float* out, const float* w1_packed, const float* bias1, float* t1, long long M,
                                         int C1, hands_stream_t stream) {
  if (!t2 || !w3_packed || !bias3 || !identity || !out || !w1_packed || !bias1 || !t1) return HANDS_EINVAL;
  if (M <= 0 || M % LK_PX || M / LK_PX > 0x7fffffffLL || (C1 != 64 && C1 != 128)) return HANDS_EINVAL;
  LinkArgs a;
  a.t2 = t2; a.w3 = w3_packed; a.b3 = bias3; a.res = identity; a.out = out; a.w1 = w1_packed; a.b1 = bias1; a.t1 = t1;
  a.ntiles = (int)(M / LK_PX);
  const int g = a.ntiles < 2 * link_cus() ? a.ntiles : 2 * link_cus();      // persistent: 2 workgroups per CU (66.5 KB of LDS each)
  if (C1 == 64) hipLaunchKernelGGL(bottleneck_link_kernel<64>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(bottleneck_link_kernel<128>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
  HANDS_LAUNCH_CHECK();
}
