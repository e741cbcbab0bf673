// handocc.hip -- kernels of the handoccnet_light path that are not convolutions / linear layers:
// FPN top-down (bilinear upsample + add), 2x2 average / max pooling, CBAM spatial gate, FIT/SET
// embedding adds, fp32-MFMA flash attention over 1024 tokens (4 heads x 64) with the FIT sigmoid
// gate, pre-activation BatchNorm+LeakyReLU, hourglass nearest-upsample + add, spatial softmax.
// Reference: src/models/handoccnet_light/{backbone.py:40-65, cbam.py:66-82, transformer.py:71-157,
// hand_head.py:62-94,117-235,266-280}.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float f4e(const float4& v, int t) {
  return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w));
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

#define GRID_STRIDE(i, total) \
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

// out = F.interpolate(x, (H,W), bilinear, align_corners=False) + y      (backbone.py:40-42)
__global__ void upsample_bilinear_add_kernel(const float4* __restrict__ x, const float4* __restrict__ y,
                                             float4* __restrict__ out, int B, int h, int w, int H, int W, int C4) {
  const long long total = (long long)B * H * W * C4;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int xo = (int)(t % W); t /= W;
    const int yo = (int)(t % H);
    const int b = (int)(t / H);
    float sy = ((float)yo + 0.5f) * sh - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = ((float)xo + 0.5f) * sw - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float4* base = x + (long long)b * h * w * C4 + c;
    const float4 p00 = base[(long long)(y0 * w + x0) * C4], p01 = base[(long long)(y0 * w + x1) * C4];
    const float4 p10 = base[(long long)(y1 * w + x0) * C4], p11 = base[(long long)(y1 * w + x1) * C4];
    const float4 yy = y[i];
    float4 r;
    r.x = (hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x)) + yy.x;
    r.y = (hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y)) + yy.y;
    r.z = (hy * (hx * p00.z + lx * p01.z) + ly * (hx * p10.z + lx * p11.z)) + yy.z;
    r.w = (hy * (hx * p00.w + lx * p01.w) + ly * (hx * p10.w + lx * p11.w)) + yy.w;
    out[i] = r;
  }
}

// 2x2 stride-2 pooling, mode 0 = average (backbone.py:38), 1 = max (hand_head.py:219,276)
__global__ void pool2x2_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B, int H, int W, int C4,
                               int mode) {
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)B * Ho * Wo * C4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float4* p = in + ((long long)(b * H + 2 * yo) * W + 2 * xo) * C4 + c;
    const float4 a = p[0], bb = p[C4], cc = p[(long long)W * C4], d = p[(long long)W * C4 + C4];
    float4 r;
    if (mode == 0) {
      r.x = (((a.x + bb.x) + cc.x) + d.x) * 0.25f; r.y = (((a.y + bb.y) + cc.y) + d.y) * 0.25f;
      r.z = (((a.z + bb.z) + cc.z) + d.z) * 0.25f; r.w = (((a.w + bb.w) + cc.w) + d.w) * 0.25f;
    } else {
      r.x = fmaxf(fmaxf(a.x, bb.x), fmaxf(cc.x, d.x)); r.y = fmaxf(fmaxf(a.y, bb.y), fmaxf(cc.y, d.y));
      r.z = fmaxf(fmaxf(a.z, bb.z), fmaxf(cc.z, d.z)); r.w = fmaxf(fmaxf(a.w, bb.w), fmaxf(cc.w, d.w));
    }
    out[i] = r;
  }
}

// ChannelPool (cbam.py:66-68) for C = 256: one wave per pixel -> [max_c, mean_c, 0, 0]
__global__ void __launch_bounds__(256) channel_pool256_kernel(const float4* __restrict__ x, float4* __restrict__ out,
                                                              long long npix) {
  const int lane = threadIdx.x & 63;
  const long long pix = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const float4 v = x[pix * 64 + lane];
  float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
  float sm = (v.x + v.y) + (v.z + v.w);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); sm += __shfl_xor(sm, o); }
  if (lane == 0) out[pix] = make_float4(mx, sm / 256.0f, 0.f, 0.f);
}

// SpatialGate tail (cbam.py:78-82): scale = sigmoid(logit); primary = x*scale; secondary = x*(1-scale)
__global__ void gate_apply_kernel(const float4* __restrict__ x, const float* __restrict__ logit, int logit_stride,
                                  float4* __restrict__ primary, float4* __restrict__ secondary, long long npix,
                                  int C4) {
  const long long total = npix * C4;
  GRID_STRIDE(i, total) {
    const long long pix = i / C4;
    const float s = sigmoidf(logit[pix * logit_stride]);
    const float t = 1.0f - s;
    const float4 v = x[i];
    primary[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
    secondary[i] = make_float4(v.x * t, v.y * t, v.z * t, v.w * t);
  }
}

// transformer.py:123-134: q_embed = (query + q_embedding) + kpe ; k_embed = (key + k_embedding) + kpe
__global__ void add_embed2_kernel(const float4* __restrict__ query, const float4* __restrict__ key,
                                  const float4* __restrict__ qemb, const float4* __restrict__ kemb,
                                  const float4* __restrict__ kpe, float4* __restrict__ oq, float4* __restrict__ ok,
                                  int B, int N, int C4) {
  const long long total = (long long)B * N * C4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4);
    const long long bt = i / C4;
    const int t = (int)(bt % N), b = (int)(bt / N);
    const float4 kp = kpe[(long long)b * C4 + c];
    oq[i] = f4add(f4add(query[i], qemb[(long long)t * C4 + c]), kp);
    ok[i] = f4add(f4add(key[i], kemb[(long long)t * C4 + c]), kp);
  }
}

// x[b,t,:] (+)= vec[b,:]  (model.py:88-89)
__global__ void add_rowvec_kernel(const float4* __restrict__ x, const float4* __restrict__ vec, float4* __restrict__ out,
                                  int B, int N, int C4) {
  const long long total = (long long)B * N * C4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4);
    const int b = (int)(i / C4 / N);
    out[i] = f4add(x[i], vec[(long long)b * C4 + c]);
  }
}

// out[b,c] = sum_t x[b,t,c]   (the key sum of the FIT gate).  One workgroup per (sample, 64 channels):
// 16 token groups x 16 float4 columns; group g adds tokens g, g+16, ... in order, then the 16 partial
// sums are added in group order -- a fixed order for every batch size.
__global__ void __launch_bounds__(256) token_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int N, int C) {
  __shared__ float4 part[16][16];
  const int cblocks = C / 64;
  const int b = blockIdx.x / cblocks, cb = blockIdx.x - b * cblocks;
  const int g = threadIdx.x >> 4, c4 = threadIdx.x & 15;
  const float4* p = reinterpret_cast<const float4*>(x + ((long long)b * N) * C + cb * 64) + c4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = g; t < N; t += 16) {
    const float4 v = p[(long long)t * (C / 4)];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  part[g][c4] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    float4 a = part[0][threadIdx.x];
    for (int k = 1; k < 16; ++k) {
      const float4 v = part[k][threadIdx.x];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (long long)b * C + cb * 64 + threadIdx.x * 4) = a;
  }
}

// pre-activation BatchNorm (folded to scale/shift) + LeakyReLU(0.01)  (hand_head.py:131-133,170-172)
__global__ void bn_leaky_kernel(const float4* __restrict__ x, const float4* __restrict__ scale,
                                const float4* __restrict__ shift, float4* __restrict__ out, long long npix, int C4) {
  const long long total = npix * C4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4);
    const float4 v = x[i], s = scale[c], t = shift[c];
    float4 r = make_float4(v.x * s.x + t.x, v.y * s.y + t.y, v.z * s.z + t.z, v.w * s.w + t.w);
    r.x = r.x > 0.f ? r.x : 0.01f * r.x; r.y = r.y > 0.f ? r.y : 0.01f * r.y;
    r.z = r.z > 0.f ? r.z : 0.01f * r.z; r.w = r.w > 0.f ? r.w : 0.01f * r.w;
    out[i] = r;
  }
}

// out = up1 + nearest_upsample_2x(low)   (hand_head.py:228-229)
__global__ void upsample_nearest2x_add_kernel(const float4* __restrict__ low, const float4* __restrict__ up1,
                                              float4* __restrict__ out, int B, int h, int w, int C4) {
  const int H = 2 * h, W = 2 * w;
  const long long total = (long long)B * H * W * C4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int xo = (int)(t % W); t /= W;
    const int yo = (int)(t % H);
    const int b = (int)(t / H);
    out[i] = f4add(up1[i], low[((long long)(b * h + yo / 2) * w + xo / 2) * C4 + c]);
  }
}

// spatial_softmax (hand_head.py:62-67): softmax over the N positions of latents[b,:,j] * beta[j];
// latents rows have stride ld_in (>= J), heat-map rows stride ld_out (pad channels written as 0).
__global__ void __launch_bounds__(256) spatial_softmax_kernel(const float* __restrict__ lat, int ld_in,
                                                              const float* __restrict__ betas, float* __restrict__ out,
                                                              int ld_out, int N, int J) {
  __shared__ double red[8];
  const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
  float* o = out + (long long)b * N * ld_out + j;
  if (j >= J) {
    for (int t = tid; t < N; t += 256) o[(long long)t * ld_out] = 0.f;
    return;
  }
  const float* p = lat + (long long)b * N * ld_in + j;
  // fp64 inside (round 6): this softmax is where handoccnet_light amplifies rounding most (the logits reach +-50: an fp32
  // product latent * beta alone moves a probability by |x| 2^-24 relative, tools/hon_error_stages.py), and it is 21 x 1024
  // values per crop -- the exact product, exp and sum in fp64, ONE rounding of each probability
  const double beta = (double)betas[j];
  double mx = -INFINITY;
  for (int t = tid; t < N; t += 256) mx = fmax(mx, (double)p[(long long)t * ld_in] * beta);
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) mx = fmax(mx, __shfl_xor(mx, s));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  double sum = 0.0;
  for (int t = tid; t < N; t += 256) sum += exp((double)p[(long long)t * ld_in] * beta - mx);
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) sum += __shfl_xor(sum, s);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  for (int t = tid; t < N; t += 256) o[(long long)t * ld_out] = (float)(exp((double)p[(long long)t * ld_in] * beta - mx) / sum);
}

// ---- flash attention on fp32 MFMA: N tokens (multiple of 128), head dim 64 ---------------------------
// 1-D grid of B * heads * N/128 workgroups (XCD-aware order: the N/128 query chunks of one (crop, head) run on ONE
// XCD, so its K / V rows are fetched into one L2 instead of eight); 4 waves x 32 queries.  Loop over key tiles of
// 128 with online softmax (running max / sum per query, lane-local + one exchange with lane^32), S^T = K Q^T and
// O^T = V^T P as in transformer.hip's attention_kernel: the probabilities feed the second product straight from the
// accumulator registers.
//   * K and V tiles (32 KB each, single-buffered) arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers,
//     no ds_write pass).  The DMA destination is lane-linear, so the K tile's 16-byte chunk c of key row r is FETCHED by
//     the lane whose slot is c ^ (r & 15) (swizzle on the source address) and read back with the same XOR: the
//     ds_read_b128 fragment reads (32 rows, one chunk column) are bank-conflict free.  V stays row-major [key][64] and is
//     read as the A operand with conflict-free 32-bit reads (lanes 0-31 = 32 consecutive d of one key): no transpose.
//   * Two barriers per key tile, each preceded by vmcnt(0): V(t) is issued when every wave has finished P V(t-1) and
//     flies under Q K^T(t); K(t+1) is issued when every wave has finished Q K^T(t) and flies under softmax + P V(t).
//   * FOLD: scale is a power of two (64^-0.5 = 0.125), so q * scale is exact and (q k) * scale of transformer.py:81
//     equals (q * scale) k bit for bit: the 64 multiplies per lane and key tile disappear.  exp(x), x <= 0, is
//     exp2 of a compensated x * log2(e) (v_exp_f32 on [-0.5, 0.5] + v_ldexp_f32: ~1 ulp, 7 instructions instead of
//     the library's 13 with its range checks).
// Epilogue: O / l, optional FIT gate sigmoid((q2 . sum_j k2_j) * scale) (transformer.py:84-90, the key
// sum hoisted out of the N x N product), optional residual (SET: query + attn).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ float exp_nonpos(float x) {     // x <= 0, finite
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
  const float n = rintf(x * L2E_HI);
  float f = fmaf(x, L2E_HI, -n);                           // x * log2(e) - n with one rounding
  f = fmaf(x, L2E_LO, f);
  return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

template <bool FOLD>
__global__ void __launch_bounds__(256, 2) flash_attention64_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, const float* __restrict__ q2,
                                                                   const float* __restrict__ k2sum, const float* __restrict__ resid,
                                                                   float* __restrict__ out, int N, int heads, float scale) {
  constexpr int D = 64, KT = 128;
  __shared__ __attribute__((aligned(1024))) float lds[2 * KT * D];
  float* sK = lds;
  float* sV = lds + KT * D;
  // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs; give each XCD a contiguous range
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int qd = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = ((xcd < r) ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int nq = N / 128;
  const int qc = bid % nq, h = (bid / nq) % heads, b = bid / (nq * heads);
  const int C = heads * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
  const long long row0 = (long long)b * N;
  const int qrow = qc * 128 + wave * 32 + (lane & 31);

  // DMA source of this lane for piece i (4 key rows = 1 KB) of this wave's 8 pieces of a tile: LDS position p = lane
  // -> row 4 * (8 wave + i) + (lane >> 4), slot lane & 15
  const int r_in = lane >> 4, slot = lane & 15;
  const float* kb0 = k + row0 * C + h * D;
  const float* vb0 = v + row0 * C + h * D;
  // source address = wave-uniform row base (scalar) + a 32-bit lane offset; (r & 15) = 4 (i & 3) + r_in, so the K tile
  // needs four lane offsets and the V tile one
  int koff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) koff[i] = r_in * C + ((slot ^ (4 * i + r_in)) << 2);
  const int voff = r_in * C + (slot << 2);
#define ISSUE_TILE(BASE, DST, KTI, SWZ)                                                                          \
  do {                                                                                                            \
    const float* ub = (BASE) + (long long)((KTI) * KT + wave * 32) * C;                                           \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                 \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(ub + i * 4 * C + ((SWZ) ? koff[i & 3] : voff)),              \
                                       (lds_void_t*)((DST) + (wave * 8 + i) * 256), 16, 0, 0);                    \
  } while (0)

  ISSUE_TILE(kb0, sK, 0, true);

  float4 qf[D / 8];
  {
    const float* qp = q + (row0 + qrow) * C + h * D + half * 4;
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk) {
      qf[kk] = *reinterpret_cast<const float4*>(qp + kk * 8);
      if constexpr (FOLD) { qf[kk].x *= scale; qf[kk].y *= scale; qf[kk].z *= scale; qf[kk].w *= scale; }
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int swz = lane & 15;                                  // (key row & 15) of this lane's fragment rows
  const float* vrd = sV + (lane & 31) + half * 4 * D;         // A operand of P V: V[key][db * 32 + (lane & 31)]
  const int ntile = N / KT;

  for (int kt = 0; kt < ntile; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // K(kt) has landed (this wave's pieces) ...
    __syncthreads();                                          // ... everybody's; and every wave is done with V(kt-1)
    ISSUE_TILE(vb0, sV, kt, false);                           // flies under Q K^T

    f32x16 s[4];
    float tmax = -INFINITY;
#pragma unroll
    for (int kbk = 0; kbk < 4; ++kbk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kbk][r] = 0.f;
      const float* krow = sK + (kbk * 32 + (lane & 31)) * D;
#pragma unroll
      for (int kk = 0; kk < D / 8; ++kk) {
        const float4 kf = *reinterpret_cast<const float4*>(krow + (((2 * kk + half) ^ swz) << 2));
#pragma unroll
        for (int t = 0; t < 4; ++t)
          s[kbk] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(kf, t), f4e(qf[kk], t), s[kbk], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (!FOLD) s[kbk][r] *= scale;
        tmax = fmaxf(tmax, s[kbk][r]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // V(kt) has landed ...
    __syncthreads();                                          // ... everybody's; and every wave is done with K(kt)
    if (kt + 1 < ntile) ISSUE_TILE(kb0, sK, kt + 1, true);    // flies under softmax + P V

    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mnew = fmaxf(m, tmax);
    const float alpha = expf(m - mnew);     // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kbk][r] = exp_nonpos(s[kbk][r] - mnew); psum += s[kbk][r]; }
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = mnew;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    // P V in blocks of 32 keys (round 6): each block's product starts from zero and joins the running output by ONE add, so
    // no fp32 chain is longer than 32 keys + the 32 block sums of a row (one 1024-key FMA chain per output before: the FIT / SET
    // outputs sat 2-2.5x farther from an fp64 evaluation than a path with exact convolutions needs -- tools/hon_error_stages.py)
#pragma unroll
    for (int kbk = 0; kbk < 4; ++kbk) {
      f32x16 t0, t1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { t0[r] = 0.f; t1[r] = 0.f; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // register r of block kbk holds P for key kbk * 32 + 8 (r >> 2) + (r & 3) + 4 half
        const float* vp = vrd + (kbk * 32 + 8 * (r >> 2) + (r & 3)) * D;
        const float a0 = vp[0], a1 = vp[32];
        t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[kbk][r], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[kbk][r], t1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] += t0[r]; o[1][r] += t1[r]; }
    }
  }
#undef ISSUE_TILE

  float gate = 1.0f;
  if (q2) {
    const float* q2p = q2 + (row0 + qrow) * C + h * D + half * 32;
    const float* ks = k2sum + (long long)b * C + h * D + half * 32;
    float acc = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const float4 a = *reinterpret_cast<const float4*>(q2p + d4 * 4), bb = *reinterpret_cast<const float4*>(ks + d4 * 4);
      acc += a.x * bb.x; acc += a.y * bb.y; acc += a.z * bb.z; acc += a.w * bb.w;
    }
    acc += __shfl_xor(acc, 32);
    gate = sigmoidf(acc * scale);
  }
  const float inv = 1.0f / l;
  float* orow = out + (row0 + qrow) * C + h * D;
  const float* rrow = resid ? resid + (row0 + qrow) * C + h * D : nullptr;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int d0 = db * 32 + q4 * 8 + half * 4;
      float4 r = make_float4(o[db][q4 * 4 + 0] * inv * gate, o[db][q4 * 4 + 1] * inv * gate,
                             o[db][q4 * 4 + 2] * inv * gate, o[db][q4 * 4 + 3] * inv * gate);
      if (rrow) r = f4add(*reinterpret_cast<const float4*>(rrow + d0), r);
      *reinterpret_cast<float4*>(orow + d0) = r;
    }
}

}  // namespace

#define S(st) ((hipStream_t)(st))

extern "C" {

int hands_upsample_bilinear_add_f32(const float* x, const float* y, float* out, int B, int h, int w, int H, int W,
                                    int C, hands_stream_t stream) {
  if (!x || !y || !out || B <= 0 || C % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(upsample_bilinear_add_kernel, dim3(hands_grid_1d((long long)B * H * W * C / 4, 256)), dim3(256), 0,
                     S(stream), (const float4*)x, (const float4*)y, (float4*)out, B, h, w, H, W, C / 4);
  HANDS_LAUNCH_CHECK();
}

int hands_pool2x2_nhwc_f32(const float* in, float* out, int B, int H, int W, int C, int mode, hands_stream_t stream) {
  if (!in || !out || B <= 0 || C % 4 || H % 2 || W % 2 || (mode != 0 && mode != 1)) return HANDS_EINVAL;
  hipLaunchKernelGGL(pool2x2_kernel, dim3(hands_grid_1d((long long)B * (H / 2) * (W / 2) * C / 4, 256)), dim3(256), 0,
                     S(stream), (const float4*)in, (float4*)out, B, H, W, C / 4, mode);
  HANDS_LAUNCH_CHECK();
}

int hands_channel_pool_f32(const float* x, float* out, long long npix, int C, hands_stream_t stream) {
  if (!x || !out || npix <= 0 || C != 256) return HANDS_EINVAL;
  hipLaunchKernelGGL(channel_pool256_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, S(stream),
                     (const float4*)x, (float4*)out, npix);
  HANDS_LAUNCH_CHECK();
}

int hands_gate_apply_f32(const float* x, const float* logit, int logit_stride, float* primary, float* secondary,
                         long long npix, int C, hands_stream_t stream) {
  if (!x || !logit || !primary || !secondary || npix <= 0 || C % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(gate_apply_kernel, dim3(hands_grid_1d(npix * C / 4, 256)), dim3(256), 0, S(stream),
                     (const float4*)x, logit, logit_stride, (float4*)primary, (float4*)secondary, npix, C / 4);
  HANDS_LAUNCH_CHECK();
}

int hands_add_embed2_f32(const float* query, const float* key, const float* q_emb, const float* k_emb,
                         const float* kpe, float* out_q, float* out_k, int B, int N, int C, hands_stream_t stream) {
  if (!query || !key || !q_emb || !k_emb || !kpe || !out_q || !out_k || B <= 0 || C % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(add_embed2_kernel, dim3(hands_grid_1d((long long)B * N * C / 4, 256)), dim3(256), 0, S(stream),
                     (const float4*)query, (const float4*)key, (const float4*)q_emb, (const float4*)k_emb,
                     (const float4*)kpe, (float4*)out_q, (float4*)out_k, B, N, C / 4);
  HANDS_LAUNCH_CHECK();
}

int hands_add_rowvec_f32(const float* x, const float* vec, float* out, int B, int N, int C, hands_stream_t stream) {
  if (!x || !vec || !out || B <= 0 || C % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(hands_grid_1d((long long)B * N * C / 4, 256)), dim3(256), 0, S(stream),
                     (const float4*)x, (const float4*)vec, (float4*)out, B, N, C / 4);
  HANDS_LAUNCH_CHECK();
}

int hands_token_sum_f32(const float* x, float* out, int B, int N, int C, hands_stream_t stream) {
  if (!x || !out || B <= 0 || N <= 0 || C <= 0 || C % 64) return HANDS_EINVAL;
  hipLaunchKernelGGL(token_sum_kernel, dim3((unsigned)(B * (C / 64))), dim3(256), 0, S(stream), x, out, B, N, C);
  HANDS_LAUNCH_CHECK();
}

int hands_bn_leaky_f32(const float* x, const float* scale, const float* shift, float* out, long long npix, int C,
                       hands_stream_t stream) {
  if (!x || !scale || !shift || !out || npix <= 0 || C % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(bn_leaky_kernel, dim3(hands_grid_1d(npix * C / 4, 256)), dim3(256), 0, S(stream),
                     (const float4*)x, (const float4*)scale, (const float4*)shift, (float4*)out, npix, C / 4);
  HANDS_LAUNCH_CHECK();
}

int hands_upsample_nearest2x_add_f32(const float* low, const float* up1, float* out, int B, int h, int w, int C,
                                     hands_stream_t stream) {
  if (!low || !up1 || !out || B <= 0 || C % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(upsample_nearest2x_add_kernel, dim3(hands_grid_1d((long long)B * 4 * h * w * C / 4, 256)), dim3(256),
                     0, S(stream), (const float4*)low, (const float4*)up1, (float4*)out, B, h, w, C / 4);
  HANDS_LAUNCH_CHECK();
}

int hands_spatial_softmax_f32(const float* latents, int ld_in, const float* betas, float* heatmaps, int ld_out, int B,
                              int N, int J, hands_stream_t stream) {
  if (!latents || !betas || !heatmaps || B <= 0 || J <= 0 || ld_in < J || ld_out < J) return HANDS_EINVAL;
  hipLaunchKernelGGL(spatial_softmax_kernel, dim3(ld_out, B), dim3(256), 0, S(stream), latents, ld_in, betas, heatmaps,
                     ld_out, N, J);
  HANDS_LAUNCH_CHECK();
}

int hands_flash_attention_f32(const float* q, const float* k, const float* v, const float* q2, const float* k2sum,
                              const float* resid, float* out, int B, int N, int heads, int head_dim, float scale,
                              hands_stream_t stream) {
  if (!q || !k || !v || !out || B <= 0 || heads <= 0 || head_dim != 64 || N % 128 || (q2 && !k2sum)) return HANDS_EINVAL;
  const long long nwg = (long long)B * heads * (N / 128);
  if (nwg > 0x7fffffffLL) return HANDS_EINVAL;
  // scale a power of two (head_dim 64 -> 0.125): folded into q exactly; any other value is applied after the product
  const bool pow2 = scale > 0.f && (__builtin_bit_cast(unsigned, scale) & 0x007fffffu) == 0u && scale >= 1e-30f;
  if (pow2)
    hipLaunchKernelGGL(flash_attention64_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, S(stream), q, k, v, q2, k2sum,
                       resid, out, N, heads, scale);
  else
    hipLaunchKernelGGL(flash_attention64_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, S(stream), q, k, v, q2, k2sum,
                       resid, out, N, heads, scale);
  HANDS_LAUNCH_CHECK();
}

}  // extern "C"
