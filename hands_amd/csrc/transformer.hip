// transformer.hip -- kernels of the hamer_light path (ViT-H/16 backbone + cross-attention decoder head)
// that are not GEMMs: bilinear resize+crop to NHWC4, LayerNorm, position/KPE adds, fp32-MFMA
// multi-head attention (192 tokens, 16 heads x 80), single-query cross-attention.
// Reference: src/models/hamer_light/{model.py:75-151, vit.py:89-151,320-342, pos_emb.py:28-64,
// pose_transformer.py:89-124,160-201}.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// exp(x) for finite x <= 0: exp2 of a compensated x * log2(e) (v_exp_f32 on [-0.5, 0.5] + v_ldexp_f32, ~1 ulp; the
// library's expf carries range checks that cannot trigger here)
__device__ __forceinline__ float exp_nonpos(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
  const float n = rintf(x * L2E_HI);
  float f = fmaf(x, L2E_HI, -n);
  f = fmaf(x, L2E_LO, f);
  return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

__device__ __forceinline__ float f4e(const float4& v, int t) {
  return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w));
}

// ---- F.interpolate(bilinear, align_corners=False) + column crop, NCHW3 -> NHWC4 ---------------------
// model.py:82-100: resize (Hin,Win) -> (S,S), then keep columns [col0, col0+Wc).
__global__ void resize_crop_kernel(const float* __restrict__ in, float4* __restrict__ out, int B, int Hin,
                                   int Win, int S, int col0, int Wc) {
  const long long total = (long long)B * S * Wc;
  const float sh = (float)Hin / (float)S, sw = (float)Win / (float)S;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wc);
    long long t = i / Wc;
    const int yo = (int)(t % S);
    const int b = (int)(t / S);
    // area_pixel_compute_source_index: max(0, (dst + 0.5) * scale - 0.5)
    float sy = ((float)yo + 0.5f) * sh - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = ((float)(xo + col0) + 0.5f) * sw - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* p = in + ((long long)b * 3 + c) * Hin * Win;
      const float p00 = p[y0 * Win + x0], p01 = p[y0 * Win + x1], p10 = p[y1 * Win + x0], p11 = p[y1 * Win + x1];
      r[c] = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
    }
    out[i] = make_float4(r[0], r[1], r[2], 0.f);
  }
}

// ---- LayerNorm over the last dim, one wave per row; optional per-group row vector added after --------
template <int VPL>   // float4 per lane: C = 256 * VPL
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ out,
                                                        const float* __restrict__ addvec, int rows_per_vec,
                                                        int M, float eps) {
  constexpr int C = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * C);
  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = xr[lane + 64 * i];
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  const float4* a4 = addvec ? reinterpret_cast<const float4*>(addvec + (long long)(row / rows_per_vec) * C) : nullptr;
  float4* orow = reinterpret_cast<float4*>(out + (long long)row * C);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float4 g = g4[lane + 64 * i], bb = b4[lane + 64 * i];
    float4 y;
    y.x = (v[i].x - mean) * rstd * g.x + bb.x;
    y.y = (v[i].y - mean) * rstd * g.y + bb.y;
    y.z = (v[i].z - mean) * rstd * g.z + bb.z;
    y.w = (v[i].w - mean) * rstd * g.w + bb.w;
    if (a4) { const float4 a = a4[lane + 64 * i]; y.x += a.x; y.y += a.y; y.z += a.z; y.w += a.w; }
    orow[lane + 64 * i] = y;
  }
}

// ---- x[b,t,:] = ((x + pos[1+t]) + pos[0]) + vec[b]   (vit.py:326-330) --------------------------------
__global__ void add_pos_kernel(float4* __restrict__ x, const float4* __restrict__ pos, const float4* __restrict__ vec,
                               int B, int T, int C4) {
  const long long total = (long long)B * T * C4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const long long bt = i / C4;
    const int t = (int)(bt % T), b = (int)(bt / T);
    float4 v = x[i];
    const float4 p1 = pos[(long long)(1 + t) * C4 + c], p0 = pos[c];
    v.x = (v.x + p1.x) + p0.x; v.y = (v.y + p1.y) + p0.y; v.z = (v.z + p1.z) + p0.z; v.w = (v.w + p1.w) + p0.w;
    if (vec) { const float4 k = vec[(long long)b * C4 + c]; v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w; }
    x[i] = v;
  }
}

// ---- KPE encoding rows [center 4L | corner 16L | 0-pad]  (pos_emb.py:53-67, same layout as model.py:444-460)
__global__ void kpe_encode_kernel(const float* __restrict__ center, const float* __restrict__ corner,
                                  float* __restrict__ out, int B, int ld, int L) {
  const int total = B * ld;
  const int nce = 4 * L, nco = 16 * L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / ld, e = i - b * ld;
    float v = 0.f;
    if (e < nce + nco) {
      const bool ce = e < nce;
      const int ee = ce ? e : e - nce;
      const int nc = ce ? 2 : 8;
      const float* ang = ce ? center + b * 2 : corner + b * 8;
      const int sc = ee & 1, ci = (ee >> 1) % nc, k = (ee >> 1) / nc;
      const float xx = (float)(1 << k) * ang[ci];
      v = sc ? cosf(xx) : sinf(xx);
    }
    out[i] = v;
  }
}

// ---- multi-head self-attention on fp32 MFMA -----------------------------------------------------------
// One workgroup per (head, crop): T = 16 * TW tokens, TW waves of 16 queries each, on v_mfma_f32_16x16x4_f32.
//   S^T[key][query] = sum_d K[key][d] * (scale * Q[query][d])      (K rows = MFMA A, Q rows = MFMA B)
// A lane holds, for ITS query (lane & 15), four keys of every 16-key block (keys 16 kb + 4 g + i, g = lane >> 4): the
// softmax over keys is lane-local plus two exchanges (lane ^ 16, lane ^ 32).  The probabilities feed the second product
// straight from the accumulator registers:
//   O^T[d][query] = sum_key V^T[d][key] * P[key][query]           (V^T rows = MFMA A, P = MFMA B)
// where MFMA step i of key block kb contracts the keys {16 kb + 4 g + i : g = 0..3} -- exactly the keys the four lane
// groups hold in accumulator register i.
//
// Why 12 waves of 16 queries and not 6 waves of 32 (the form of rounds 1-3, on v_mfma_f32_32x32x2_f32): tools/prof_attn.py
// (per-wave phase stamps + HW_ID) showed that a 6-wave workgroup lands 2-2-1-1 on a CU's four SIMDs -- with two such
// workgroups resident two SIMDs carry four waves and two carry two, the light SIMDs' waves wait at the workgroup barriers for
// the loaded ones (p90 25 us of a 60 us wave life) and the matrix pipe sat at 48 % busy.  Twelve waves are three per SIMD;
// the head dimension 80 is five 16-row blocks exactly (the 32-row form padded it to 96: 17 % of the P.V MFMAs) and the score
// registers halve (48 per lane): 80 VGPRs and 64.5 KB of LDS (V^T takes K's place) = TWO workgroups per CU, six waves per
// SIMD, so one workgroup's fill and barrier phases run under the other's MFMAs; V is requested right after the first barrier
// and waits in registers under Q.K^T.  358 -> 266 us per call at 128 crops (round 4; own LDS regions for K and V^T, one
// workgroup per CU: 284 us).
template <int TW, int D>
constexpr int attention_lds_bytes() {
  return 4 * (16 * TW * (D + 4) > D * (16 * TW + 4) ? 16 * TW * (D + 4) : D * (16 * TW + 4));     // K, then V^T in its place
}

template <int TW, int D>
__global__ void __launch_bounds__(64 * TW) __attribute__((amdgpu_waves_per_eu(6, 6))) attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int heads, float scale) {
  constexpr int T = 16 * TW;
  constexpr int KR = D + 4;          // K row (floats): (D + 4) * 4 B = an odd number of 16-byte slots for D = 80
  constexpr int VR = T + 4;          // V^T row
  constexpr int NT = 64 * TW;
  constexpr int NKK = D / 16;        // 16-float steps of the head dimension: one float4 per lane group and step
  constexpr int DB = D / 16;         // 16-row blocks of V^T / O^T
  constexpr int FILL = T * (D / 4) / NT;
  static_assert(D % 16 == 0 && T * (D / 4) % NT == 0 && NT % T == 0, "fill loops assume whole iterations");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sK = lds;                   // [T][KR]
  float* sV = lds;                   // [D][VR]  (V transposed) in K's place, once every wave is done with K

  const int h = blockIdx.x, b = blockIdx.y;
  const int C = heads * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const float* base = qkv + (long long)b * T * 3 * C + h * D;

  // K tile -> LDS: consecutive lanes walk the 16-byte slots of a token's row (coalesced 320-byte rows); all loads of a
  // thread in flight before its first LDS store
  {
    float4 kv[FILL];
#pragma unroll
    for (int it = 0; it < FILL; ++it) {
      const int i = tid + it * NT;
      const int t = i / (D / 4), dq = i - t * (D / 4);
      kv[it] = *reinterpret_cast<const float4*>(base + (long long)t * 3 * C + C + dq * 4);
    }
#pragma unroll
    for (int it = 0; it < FILL; ++it) {
      const int i = tid + it * NT;
      const int t = i / (D / 4), dq = i - t * (D / 4);
      *reinterpret_cast<float4*>(sK + t * KR + dq * 4) = kv[it];
    }
  }
  // this wave's Q fragments, pre-scaled (vit.py:118 scales q before the matmul): lane (query l15, group g) holds
  // d = 16 kk + 4 g + j
  float4 qf[NKK];
  {
    const float* qrow = base + (long long)(wave * 16 + l15) * 3 * C + 4 * g;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      float4 v = *reinterpret_cast<const float4*>(qrow + kk * 16);
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      qf[kk] = v;
    }
  }
  __syncthreads();
  // V: requested now (after the barrier, whose s_waitcnt would otherwise wait for it), parked in registers under Q.K^T and
  // transposed into LDS afterwards.  thread = (token ft, 16-byte slot fq + it * NT / T): lanes walk tokens,
  // so the transposed scalar LDS writes are conflict-free
  const int ft = tid % T, fq = tid / T;
  float4 vv[FILL];
#pragma unroll
  for (int it = 0; it < FILL; ++it) vv[it] = *reinterpret_cast<const float4*>(base + (long long)ft * 3 * C + 2 * C + (fq + it * (NT / T)) * 4);

  // S^T = K . (scale Q)^T: TW independent accumulation chains (one per key block), the k-step outermost
  f32x4 s[TW];
#pragma unroll
  for (int kb = 0; kb < TW; ++kb) { s[kb][0] = 0.f; s[kb][1] = 0.f; s[kb][2] = 0.f; s[kb][3] = 0.f; }
  {
    const float* krow = sK + l15 * KR + 4 * g;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
      for (int kb = 0; kb < TW; ++kb) {
        const float4 kf = *reinterpret_cast<const float4*>(krow + kb * 16 * KR + kk * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          s[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(kf, j), f4e(qf[kk], j), s[kb], 0, 0, 0);
      }
    }
  }
  __syncthreads();   // every wave is done with K
  // V^T -> LDS in K's place (lanes walk tokens: the transposed scalar writes are conflict-free)
#pragma unroll
  for (int it = 0; it < FILL; ++it) {
    float* d = sV + (fq + it * (NT / T)) * 4 * VR + ft;
    d[0 * VR] = vv[it].x;
    d[1 * VR] = vv[it].y;
    d[2 * VR] = vv[it].z;
    d[3 * VR] = vv[it].w;
  }
  // softmax over the keys of this lane's query: exp(x - max) * (1 / sum): one reciprocal per query; e * (1 / sum) vs
  // e / sum differ by the last rounding only
  float m = s[0][0];
#pragma unroll
  for (int kb = 0; kb < TW; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kb][r]);
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < TW; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[kb][r] = exp_nonpos(s[kb][r] - m); sum += s[kb][r]; }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv_sum = 1.0f / sum;
#pragma unroll
  for (int kb = 0; kb < TW; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[kb][r] *= inv_sum;
  __syncthreads();   // V^T complete

  // O^T = V^T . P: DB independent chains (one per 16-row block of V^T), the key block outermost
  f32x4 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) { o[db][0] = 0.f; o[db][1] = 0.f; o[db][2] = 0.f; o[db][3] = 0.f; }
  {
    const float* vrow = sV + l15 * VR + 4 * g;
#pragma unroll
    for (int kb = 0; kb < TW; ++kb) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const float4 vf = *reinterpret_cast<const float4*>(vrow + db * 16 * VR + kb * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(vf, i), s[kb][i], o[db], 0, 0, 0);
      }
    }
  }
  // O^T[d = 16 db + 4 g + i][query l15] -> out[(b*T + query)*C + h*D + d], 4 consecutive d per store
  float* orow = out + ((long long)b * T + wave * 16 + l15) * C + h * D + 4 * g;
#pragma unroll
  for (int db = 0; db < DB; ++db)
    *reinterpret_cast<float4*>(orow + db * 16) = make_float4(o[db][0], o[db][1], o[db][2], o[db][3]);
}

// ---- single-query cross-attention (decoder head): one wave per (batch, head), head dim 64 ------------
// dots[t] = (q . k[t]) * scale; softmax over t; out = sum_t attn[t] v[t]   (pose_transformer.py:113-123)
__global__ void __launch_bounds__(64) cross_attention_1q_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                                float* __restrict__ out, int T, int heads, float scale) {
  constexpr int D = 64;
  __shared__ float sq[D];
  __shared__ float sattn[1024];
  const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int inner = heads * D;
  sq[lane] = q[(long long)b * inner + h * D + lane];
  __syncthreads();
  const float* kbase = kv + (long long)b * T * 2 * inner + h * D;
  float mx = -INFINITY;
  for (int t = lane; t < T; t += 64) {
    const float4* kr = reinterpret_cast<const float4*>(kbase + (long long)t * 2 * inner);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < D / 4; ++i) {
      const float4 kk = kr[i];
      acc += sq[4 * i] * kk.x; acc += sq[4 * i + 1] * kk.y; acc += sq[4 * i + 2] * kk.z; acc += sq[4 * i + 3] * kk.w;
    }
    acc *= scale;
    sattn[t] = acc;
    mx = fmaxf(mx, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int t = lane; t < T; t += 64) { const float e = expf(sattn[t] - mx); sattn[t] = e; sum += e; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  __syncthreads();
  const float* vbase = kbase + inner + lane;
  float acc = 0.f;
  for (int t = 0; t < T; ++t) acc += (sattn[t] / sum) * vbase[(long long)t * 2 * inner];
  out[(long long)b * inner + h * D + lane] = acc;
}

// ---- rot6d_to_rotmat, COLUMNS convention (geometry.py:47-62) ------------------------------------------
__global__ void rot6d_cols_kernel(const float* __restrict__ pose6d, int ld6, float* __restrict__ rotmat, int B) {
  const int total = B * 16;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i >> 4, j = i & 15;
    const float* s = pose6d + (long long)b * ld6 + j * 6;
    const float a1x = s[0], a1y = s[1], a1z = s[2], a2x = s[3], a2y = s[4], a2z = s[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    float b2x = a2x - d * b1x, b2y = a2y - d * b1y, b2z = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
    b2x /= n2; b2y /= n2; b2z /= n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    float* o = rotmat + (long long)i * 9;
    o[0] = b1x; o[1] = b2x; o[2] = b3x;
    o[3] = b1y; o[4] = b2y; o[5] = b3y;
    o[6] = b1z; o[7] = b2z; o[8] = b3z;
  }
}

}  // namespace

extern "C" {

int hands_resize_crop_nchw3_to_nhwc4_f32(const float* in, float* out, int B, int Hin, int Win, int S, int col0,
                                         int Wc, hands_stream_t stream) {
  if (!in || !out || B <= 0 || S <= 0 || col0 < 0 || col0 + Wc > S) return HANDS_EINVAL;
  hipLaunchKernelGGL(resize_crop_kernel, dim3(hands_grid_1d((long long)B * S * Wc, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, (float4*)out, B, Hin, Win, S, col0, Wc);
  HANDS_LAUNCH_CHECK();
}

int hands_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out, const float* addvec,
                        int rows_per_vec, int M, int C, float eps, hands_stream_t stream) {
  if (!x || !gamma || !beta || !out || M <= 0 || (addvec && rows_per_vec <= 0)) return HANDS_EINVAL;
  const dim3 grid((M + 3) / 4), block(256);
  if (C == 1280)
    hipLaunchKernelGGL(layernorm_kernel<5>, grid, block, 0, (hipStream_t)stream, x, gamma, beta, out, addvec,
                       rows_per_vec, M, eps);
  else if (C == 1024)
    hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, (hipStream_t)stream, x, gamma, beta, out, addvec,
                       rows_per_vec, M, eps);
  else if (C == 256)
    hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, (hipStream_t)stream, x, gamma, beta, out, addvec,
                       rows_per_vec, M, eps);
  else
    return HANDS_EINVAL;
  HANDS_LAUNCH_CHECK();
}

int hands_add_pos_f32(float* x, const float* pos, const float* vec, int B, int T, int C, hands_stream_t stream) {
  if (!x || !pos || B <= 0 || T <= 0 || C % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(add_pos_kernel, dim3(hands_grid_1d((long long)B * T * C / 4, 256)), dim3(256), 0,
                     (hipStream_t)stream, (float4*)x, (const float4*)pos, (const float4*)vec, B, T, C / 4);
  HANDS_LAUNCH_CHECK();
}

int hands_kpe_encode_f32(const float* center_angle, const float* corner_angle, float* out, int B, int ld,
                         int n_freq, hands_stream_t stream) {
  if (!center_angle || !corner_angle || !out || B <= 0 || n_freq < 1 || n_freq > 16 || ld < 20 * n_freq)
    return HANDS_EINVAL;
  hipLaunchKernelGGL(kpe_encode_kernel, dim3(hands_grid_1d((long long)B * ld, 256)), dim3(256), 0,
                     (hipStream_t)stream, center_angle, corner_angle, out, B, ld, n_freq);
  HANDS_LAUNCH_CHECK();
}

int hands_attention_f32(const float* qkv, float* out, int B, int T, int heads, int head_dim, float scale,
                        hands_stream_t stream) {
  if (!qkv || !out || B <= 0 || heads <= 0) return HANDS_EINVAL;
  if (T == 192 && head_dim == 80) {
    // 127 KB of dynamic LDS (> the 64 KB default cap of a launch): raise the kernel's limit once per device
    static std::atomic<int> lds_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return HANDS_EINVAL;
    if (!lds_set[dev].load(std::memory_order_acquire)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<12, 80>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, attention_lds_bytes<12, 80>());
      if (e != hipSuccess) return (int)e;
      lds_set[dev].store(1, std::memory_order_release);
    }
    hipLaunchKernelGGL((attention_kernel<12, 80>), dim3(heads, B), dim3(768), (attention_lds_bytes<12, 80>()), (hipStream_t)stream,
                       qkv, out, heads, scale);
  } else {
    return HANDS_EINVAL;
  }
  HANDS_LAUNCH_CHECK();
}

int hands_cross_attention_1q_f32(const float* q, const float* kv, float* out, int B, int T, int heads,
                                 int head_dim, float scale, hands_stream_t stream) {
  if (!q || !kv || !out || B <= 0 || T <= 0 || T > 1024 || head_dim != 64 || heads <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(cross_attention_1q_kernel, dim3(heads, B), dim3(64), 0, (hipStream_t)stream, q, kv, out, T,
                     heads, scale);
  HANDS_LAUNCH_CHECK();
}

int hands_rot6d_to_matrix_cols_f32(const float* pose6d, int ld6, float* rotmat, int B, hands_stream_t stream) {
  if (!pose6d || !rotmat || B <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(rot6d_cols_kernel, dim3(hands_grid_1d((long long)B * 16, 256)), dim3(256), 0,
                     (hipStream_t)stream, pose6d, ld6, rotmat, B);
  HANDS_LAUNCH_CHECK();
}

}  // extern "C"
