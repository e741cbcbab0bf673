// elementwise.hip -- the HBM-bound glue kernels of the hands_light forward path (gfx950).
// Layout conversion, max-pool, sum-pool, key-point-encoding concat, HMR state init, 6D->matrix,
// flip/swap, grasp-input assembly.  All are streaming kernels: 16-byte accesses, consecutive lanes
// on consecutive addresses, grid-stride over <= 2048 blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"
#include "rot_device.h"

namespace {

// ---- (B,3,H,W) -> (B,H,W,4) ----------------------------------------------------------------------
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ in, float4* __restrict__ out,
                                      long long npix_total, int HW) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix_total;
       p += (long long)gridDim.x * blockDim.x) {
    const long long b = p / HW;
    const int r = (int)(p - b * HW);
    const float* src = in + b * 3LL * HW + r;
    out[p] = make_float4(src[0], src[HW], src[2LL * HW], 0.f);
  }
}

// ---- MaxPool 3x3 s2 p1, NHWC, 4 channels per thread ---------------------------------------------
__global__ void maxpool3x3s2_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B,
                                    int H, int W, int C4, int Ho, int Wo) {
  const long long total = (long long)B * Ho * Wo * C4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hi = ho * 2 - 1 + dh;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int wi = wo * 2 - 1 + dw;
        if ((unsigned)wi >= (unsigned)W) continue;
        const float4 v = in[((long long)(b * H + hi) * W + wi) * C4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    out[i] = m;
  }
}

// ---- feat_vec[b,c] = sum_p feat[b,p,c] ------------------------------------------------------------
__global__ void sumpool_kernel(const float4* __restrict__ in, float* __restrict__ out, int B, int HW,
                               int C4, int out_stride) {
  const int total = B * C4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / C4, c = i - b * C4;
    const float4* src = in + (long long)b * HW * C4 + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < HW; ++p) {
      const float4 v = src[(long long)p * C4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (long long)b * out_stride + c * 4) = s;
  }
}

// ---- feat[b,c] = (sum_p feat[b,p,c]) / HW: nn.AdaptiveAvgPool2d(1) of HandHMR.forward(use_pool=True) (hand_hmr.py:73-78),
//      the `no_crops` route of model.py:316-318 (arctic_light).  Same sequential sum as sumpool_kernel, then one division.
__global__ void avgpool_kernel(const float4* __restrict__ in, float* __restrict__ out, int B, int HW, int C4, int out_stride) {
  const int total = B * C4;
  const float n = (float)HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / C4, c = i - b * C4;
    const float4* src = in + (long long)b * HW * C4 + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < HW; ++p) {
      const float4 v = src[(long long)p * C4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (long long)b * out_stride + c * 4) = make_float4(s.x / n, s.y / n, s.z / n, s.w / n);
  }
}

// ---- cat([crop+glb, center_enc, corner_enc]) ------------------------------------------------------
// encoding element e of an angle vector with nc components: layout (L, nc, 2): k = e/(2nc),
// ci = (e/2)%nc, sc = e&1 -> sin/cos(2^k * angle[ci])            (model.py:444-460)
__device__ __forceinline__ float kpe_elem(const float* ang, int nc, int e) {
  const int sc = e & 1, ci = (e >> 1) % nc, k = (e >> 1) / nc;
  const float x = (float)(1 << k) * ang[ci];
  return sc ? cosf(x) : sinf(x);
}

__global__ void kpe_concat_kernel(const float4* __restrict__ crop, const float4* __restrict__ glb,
                                  const float* __restrict__ center, const float* __restrict__ corner,
                                  float4* __restrict__ out, int B2, int Bg, int HW, int C4, int L) {
  const int nce = 4 * L, nco = 16 * L;          // 2*L*2, 2*L*8
  const int Co4 = C4 + (nce + nco) / 4;
  const long long total = (long long)B2 * HW * Co4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Co4);
    const long long pix = i / Co4;              // b2*HW + p
    const int b2 = (int)(pix / HW);
    const int p = (int)(pix - (long long)b2 * HW);
    float4 v;
    if (c < C4) {
      const float4 a = crop[pix * C4 + c];
      if (glb) {                                  // use_glb_feat (model.py:263-264); nullptr: the crop features alone (:266-267)
        const float4 g = glb[((long long)(b2 % Bg) * HW + p) * C4 + c];
        v = make_float4(a.x + g.x, a.y + g.y, a.z + g.z, a.w + g.w);
      } else {
        v = a;
      }
    } else {
      const int e0 = (c - C4) * 4;
      float r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u;
        r[u] = e < nce ? kpe_elem(center + b2 * 2, 2, e) : kpe_elem(corner + b2 * 8, 8, e - nce);
      }
      v = make_float4(r[0], r[1], r[2], r[3]);
    }
    out[i] = v;
  }
}

// ---- image-level positional encodings (pos_enc = 'center' | 'corner' | 'center+corner', model.py:203-218): the crop image
//      with the per-sample encoding repeated over every pixel as extra input channels, written as the NHWC tensor
//      (B, H, W, Cpad) the general convolution reads: [r g b | center enc (4 L) | corner enc (16 L) | zeros].
//      mode bit 0: center, bit 1: corner.
__global__ void image_posenc_kernel(const float* __restrict__ img, const float* __restrict__ center,
                                    const float* __restrict__ corner, float4* __restrict__ out, int B, int HW, int L, int mode,
                                    int Cp4) {
  const int nce = (mode & 1) ? 4 * L : 0, nco = (mode & 2) ? 16 * L : 0;
  const long long total = (long long)B * HW * Cp4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % Cp4);
    const long long pix = i / Cp4;
    const int b = (int)(pix / HW);
    const int p = (int)(pix - (long long)b * HW);
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = c4 * 4 + u;
      if (ch < 3) r[u] = img[((long long)b * 3 + ch) * HW + p];
      else if (ch < 3 + nce) r[u] = kpe_elem(center + b * 2, 2, ch - 3);
      else if (ch < 3 + nce + nco) r[u] = kpe_elem(corner + b * 8, 8, ch - 3 - nce);
      else r[u] = 0.f;
    }
    out[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
}


// ---- per-pixel ("dense") positional encodings: pos_enc = 'dense' | 'dense_latent' | 'cam_conv' ------------------------------
// (model.py:462-481).  Source value of encoding channel ch at pixel `pix` of one sample: L > 0: sin / cos(2^k angle[ci]) with
// ch = (k Ca + ci) 2 + {sin, cos} (torch.cat([sin, cos], dim=3).reshape(bz, -1, w, h) of the (bz, L, c, w, h) products), L == 0:
// the raw map ('cam_conv'); both times the crop mask.
__device__ __forceinline__ float dense_src(const float* __restrict__ ang, const float* __restrict__ msk, int Ca, int HsWs,
                                           int pix, int L, int ch) {
  const float m = msk[pix];
  if (L == 0) return ang[(long long)ch * HsWs + pix] * m;
  const int sc = ch & 1, q = ch >> 1, ci = q % Ca, k = q / Ca;
  const float x = (float)(1 << k) * ang[(long long)ci * HsWs + pix];
  return (sc ? cosf(x) : sinf(x)) * m;
}

// F.interpolate(mode='bilinear', align_corners=True) tap of output index `dst` (ATen compute_source_index_and_lambda:
// equal sizes copy; otherwise src = dst (in - 1) / (out - 1), index truncated, lambda clamped to [0, 1]).
struct AcTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ AcTap ac_tap(int dst, int in, int out) {
  AcTap t;
  if (in == out) { t.i0 = t.i1 = dst; t.l0 = 1.f; t.l1 = 0.f; return t; }
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)dst;
  int i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  const float l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  t.i0 = i0; t.i1 = i0 + (i0 < in - 1 ? 1 : 0); t.l1 = l1; t.l0 = 1.f - l1;
  return t;
}

// the encoding resized to (R, R) at pixel (y, x): first interpolation of model.py:471 / 480
__device__ __forceinline__ float dense_stage1(const float* __restrict__ ang, const float* __restrict__ msk, int Ca, int Hs, int Ws,
                                              int L, int ch, int R, int y, int x) {
  const AcTap ty = ac_tap(y, Hs, R), tx = ac_tap(x, Ws, R);
  const int HsWs = Hs * Ws;
  float top = tx.l0 * dense_src(ang, msk, Ca, HsWs, ty.i0 * Ws + tx.i0, L, ch);
  if (tx.l1 != 0.f) top += tx.l1 * dense_src(ang, msk, Ca, HsWs, ty.i0 * Ws + tx.i1, L, ch);
  if (ty.l1 == 0.f) return ty.l0 * top;
  float bot = tx.l0 * dense_src(ang, msk, Ca, HsWs, ty.i1 * Ws + tx.i0, L, ch);
  if (tx.l1 != 0.f) bot += tx.l1 * dense_src(ang, msk, Ca, HsWs, ty.i1 * Ws + tx.i1, L, ch);
  return ty.l0 * top + ty.l1 * bot;
}

// out (B, Ho, Wo, ld) NHWC: channels [c_off, c_off + Cenc) = the encoding resized to (R, R) and then to (Ho, Wo).  With `img`
// (B, 3, Ho, Wo) NCHW the kernel writes the whole pixel: [r g b | encoding | zeros] (c_off = 3, the widened conv1's input).
__global__ void dense_posenc_kernel(const float* __restrict__ angle, const float* __restrict__ mask, const float* __restrict__ img,
                                    float* __restrict__ out, int B, int Ca, int Hs, int Ws, int L, int R, int Ho, int Wo, int ld,
                                    int c_off, int Cenc) {
  const int ncol = img ? ld : Cenc;
  const long long total = (long long)B * Ho * Wo * ncol;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ncol);
    const long long pix = i / ncol;
    const int x = (int)(pix % Wo);
    const int y = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long long)Wo * Ho));
    const int ch = img ? c - c_off : c;
    float v;
    if (img && c < 3) {
      v = img[(((long long)b * 3 + c) * Ho + y) * Wo + x];
    } else if (ch >= Cenc) {
      v = 0.f;
    } else {
      const float* ang = angle + (long long)b * Ca * Hs * Ws;
      const float* msk = mask + (long long)b * Hs * Ws;
      if (Ho == R && Wo == R) {
        v = dense_stage1(ang, msk, Ca, Hs, Ws, L, ch, R, y, x);
      } else {                                   // second interpolation (model.py:248-249 / 280-281)
        const AcTap ty = ac_tap(y, R, Ho), tx = ac_tap(x, R, Wo);
        float top = tx.l0 * dense_stage1(ang, msk, Ca, Hs, Ws, L, ch, R, ty.i0, tx.i0);
        if (tx.l1 != 0.f) top += tx.l1 * dense_stage1(ang, msk, Ca, Hs, Ws, L, ch, R, ty.i0, tx.i1);
        v = ty.l0 * top;
        if (ty.l1 != 0.f) {
          float bot = tx.l0 * dense_stage1(ang, msk, Ca, Hs, Ws, L, ch, R, ty.i1, tx.i0);
          if (tx.l1 != 0.f) bot += tx.l1 * dense_stage1(ang, msk, Ca, Hs, Ws, L, ch, R, ty.i1, tx.i1);
          v += ty.l1 * bot;
        }
      }
    }
    out[pix * ld + (img ? c : c_off + c)] = v;
  }
}

// out[b, p, :] = [a[b, p, :Ca] (+ add[b % Bg, p, :Ca]) | extra[(b, ) p, :Cb] | zeros up to ld]: torch.cat along the channels of an
// NHWC map (model.py:252-256, 284-288: features (+ global features) with the resized maps; :183: the depth head's grid)
__global__ void concat_nhwc_kernel(const float* __restrict__ a, int lda, int Ca, const float* __restrict__ add, int ld_add,
                                   const float* __restrict__ extra, long long extra_bs, int Cb, float* __restrict__ out, int ld,
                                   int B, int Bg, int HW) {
  const long long total = (long long)B * HW * ld;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld);
    const long long pix = i / ld;
    const int b = (int)(pix / HW);
    const int p = (int)(pix - (long long)b * HW);
    float v = 0.f;
    if (c < Ca) {
      v = a[pix * lda + c];
      if (add) v += add[((long long)(b % Bg) * HW + p) * ld_add + c];
    } else if (c < Ca + Cb) {
      v = extra[(long long)b * extra_bs + (long long)p * Cb + (c - Ca)];
    }
    out[i] = v;
  }
}

// F.interpolate / nn.Upsample(mode='bilinear', align_corners=True) of an NHWC map (model.py:141, 146, 151)
__global__ void upsample_bilinear_ac_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B, int h, int w, int H,
                                            int W, int C4) {
  const long long total = (long long)B * H * W * C4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const long long pix = i / C4;
    const int X = (int)(pix % W);
    const int Y = (int)((pix / W) % H);
    const int b = (int)(pix / ((long long)W * H));
    const AcTap ty = ac_tap(Y, h, H), tx = ac_tap(X, w, W);
    const float4* src = in + (long long)b * h * w * C4 + c;
    const float4 v00 = src[((long long)ty.i0 * w + tx.i0) * C4], v01 = src[((long long)ty.i0 * w + tx.i1) * C4];
    const float4 v10 = src[((long long)ty.i1 * w + tx.i0) * C4], v11 = src[((long long)ty.i1 * w + tx.i1) * C4];
    float4 r;
    r.x = ty.l0 * (tx.l0 * v00.x + tx.l1 * v01.x) + ty.l1 * (tx.l0 * v10.x + tx.l1 * v11.x);
    r.y = ty.l0 * (tx.l0 * v00.y + tx.l1 * v01.y) + ty.l1 * (tx.l0 * v10.y + tx.l1 * v11.y);
    r.z = ty.l0 * (tx.l0 * v00.z + tx.l1 * v01.z) + ty.l1 * (tx.l0 * v10.z + tx.l1 * v11.z);
    r.w = ty.l0 * (tx.l0 * v00.w + tx.l1 * v01.w) + ty.l1 * (tx.l0 * v10.w + tx.l1 * v11.w);
    out[i] = r;
  }
}

// ---- corrections of the global rotation (joint 0) -------------------------------------------------------------------------
__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}

// pos_enc = 'pcl' (model.py:330-334): pose[b, 0] = rot[b] @ pose[b, 0], in place on the heads' output (before the flip swap)
__global__ void rot_leftmul_kernel(float* __restrict__ rotmat, const float* __restrict__ rot, int B) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    float m[9], r[9], o[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) { m[e] = rotmat[(long long)b * 144 + e]; r[e] = rot[(long long)b * 9 + e]; }
    mat3_mul(r, m, o);
#pragma unroll
    for (int e = 0; e < 9; ++e) rotmat[(long long)b * 144 + e] = o[e];
  }
}

// pos_enc = 'perspective_correction' (model.py:370-376, after the flip swap): pose[b, 0] = euler_xyz(-center[b, 0], -center[b, 1], 0)
// @ pose[b, 0] on the swapped rotations `rot_m`.  The reference does it IN PLACE: in a batch without a flipped sample `pose_r` is the
// heads' own output tensor, which the grasp head reads afterwards -- then (and only then) the corrected matrix also goes to `rotmat`.
__global__ void persp_correct_kernel(float* __restrict__ rot_m, float* __restrict__ rotmat, const float* __restrict__ center,
                                     const long long* __restrict__ flipped, int Bg) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < 2 * Bg; b += gridDim.x * blockDim.x) {
    bool any = false;
    for (int s = 0; s < Bg; ++s) any |= flipped[s] != 0;
    const float e0 = -center[b * 2], e1 = -center[b * 2 + 1];
    const float c0 = cosf(e0), s0 = sinf(e0), c1 = cosf(e1), s1 = sinf(e1);
    const float rx[9] = {1.f, 0.f, 0.f, 0.f, c0, -s0, 0.f, s0, c0};
    const float ry[9] = {c1, 0.f, s1, 0.f, 1.f, 0.f, -s1, 0.f, c1};
    float rxy[9], m[9], o[9];
    mat3_mul(rx, ry, rxy);                       // (Rx Ry) Rz with Rz = identity (third angle is zero)
#pragma unroll
    for (int e = 0; e < 9; ++e) m[e] = rot_m[(long long)b * 144 + e];
    mat3_mul(rxy, m, o);
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      rot_m[(long long)b * 144 + e] = o[e];
      if (!any) rotmat[(long long)b * 144 + e] = o[e];
    }
  }
}

// ---- HMR state init --------------------------------------------------------------------------------
// row layout (ld = F + 112): [feat F | pose6d 96 | shape 10 | 0 0 | cam 3 | 0]
__global__ void hmr_init_kernel(float* __restrict__ state, const float* __restrict__ cam_init, int B,
                                int ld, int F) {
  const int total = B * 112;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / 112, e = i - b * 112;
    float v = 0.f;
    if (e < 96) { const int r = e % 6; v = (r == 0 || r == 4) ? 1.f : 0.f; }   // identity 6D
    else if (e >= 108 && e < 111) v = cam_init[b * 4 + (e - 108)];
    state[(long long)b * ld + F + e] = v;
  }
}

// ---- rotation_6d_to_matrix (rows) ------------------------------------------------------------------
__global__ void rot6d_kernel(const float* __restrict__ pose6d, int ld6, float* __restrict__ rotmat,
                             int B) {
  const int total = B * 16;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i >> 4, j = i & 15;
    const float* s = pose6d + (long long)b * ld6 + j * 6;
    const float a1x = s[0], a1y = s[1], a1z = s[2], a2x = s[3], a2y = s[4], a2z = s[5];
    // F.normalize: x / max(||x||, 1e-12)
    float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    float b2x = a2x - d * b1x, b2y = a2y - d * b1y, b2z = a2z - d * b1z;
    float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
    b2x /= n2; b2y /= n2; b2z /= n2;
    float* o = rotmat + (long long)i * 9;
    o[0] = b1x; o[1] = b1y; o[2] = b1z;
    o[3] = b2x; o[4] = b2y; o[5] = b2z;
    o[6] = b1y * b2z - b1z * b2y; o[7] = b1z * b2x - b1x * b2z; o[8] = b1x * b2y - b1y * b2x;
  }
}

// ---- is_flipped swap (model.py:341-368) -------------------------------------------------------------
__global__ void flip_swap_kernel(const int64_t* __restrict__ flipped, const float* __restrict__ rotmat,
                                 const float* __restrict__ shape, const float* __restrict__ cam,
                                 const float* __restrict__ cam_init, float* __restrict__ rotmat_o,
                                 float* __restrict__ shape_o, float* __restrict__ cam_o,
                                 float* __restrict__ cam_init_o, int Bg) {
  const int total = 2 * Bg * 16;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i >> 4, j = i & 15;         // row in [0,2Bg): right rows then left rows
    const int b = row % Bg;
    const bool f = flipped[b] != 0;
    const int src = f ? (row < Bg ? row + Bg : row - Bg) : row;   // take the other hand when flipped
    const float* m = rotmat + ((long long)src * 16 + j) * 9;
    float* o = rotmat_o + ((long long)row * 16 + j) * 9;
    if (f) {
      float aa[3];
      hands::matrix_to_axis_angle(m, aa);
      aa[1] = -aa[1]; aa[2] = -aa[2];
      hands::axis_angle_to_matrix(aa, o);
    } else {
#pragma unroll
      for (int e = 0; e < 9; ++e) o[e] = m[e];
    }
    if (j < 10) shape_o[row * 10 + j] = shape[src * 10 + j];
    if (j < 3) {
      const float sg = (f && j == 1) ? -1.f : 1.f;   // * [1,-1,1]
      cam_o[row * 3 + j] = cam[src * 3 + j] * sg;
      cam_init_o[row * 3 + j] = cam_init[src * 3 + j] * sg;
    }
  }
}

// ---- grasp-head input rows: [feat_vec F | rotmat 144 | shape 10 | 0-pad] ------------------------------
__global__ void grasp_input_kernel(const float* __restrict__ shape, int ld_shape,
                                   const float* __restrict__ rotmat, const float* __restrict__ feat_vec,
                                   float* __restrict__ out, int B2, int Bg, int F, int ld_out) {
  const long long total = (long long)B2 * ld_out;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ld_out), e = (int)(i - (long long)b * ld_out);
    float v = 0.f;
    if (e < F) v = feat_vec[(long long)(b % Bg) * F + e];
    else if (e < F + 144) v = rotmat[(long long)b * 144 + (e - F)];
    else if (e < F + 154) v = shape[(long long)b * ld_shape + (e - F - 144)];
    out[i] = v;
  }
}

// ---- rotation conversions alone (the device functions mano_pose_kernel / flip_swap_kernel use) -------
__global__ void matrix_to_axis_angle_kernel(const float* __restrict__ rotmat, float* __restrict__ aa, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v[3];
    hands::matrix_to_axis_angle(rotmat + i * 9, v);
    aa[i * 3 + 0] = v[0]; aa[i * 3 + 1] = v[1]; aa[i * 3 + 2] = v[2];
  }
}

__global__ void axis_angle_to_matrix_kernel(const float* __restrict__ aa, float* __restrict__ rotmat, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float m[9];
    hands::axis_angle_to_matrix(aa + i * 3, m);
#pragma unroll
    for (int e = 0; e < 9; ++e) rotmat[i * 9 + e] = m[e];
  }
}

}  // namespace

extern "C" {

int hands_nchw3_to_nhwc4_f32(const float* in, float* out, int B, int H, int W, hands_stream_t stream) {
  if (!in || !out || B <= 0) return HANDS_EINVAL;
  const long long n = (long long)B * H * W;
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, (float4*)out, n, H * W);
  HANDS_LAUNCH_CHECK();
}

int hands_maxpool3x3s2_nhwc_f32(const float* in, float* out, int B, int H, int W, int C,
                                hands_stream_t stream) {
  if (!in || !out || B <= 0 || C % 4) return HANDS_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long n = (long long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(hands_grid_1d(n, 256, 256 * 16)), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)in, (float4*)out, B, H, W, C / 4, Ho, Wo);
  HANDS_LAUNCH_CHECK();
}

int hands_sumpool_nhwc_f32(const float* feat, float* out, int B, int HW, int C, int out_stride,
                           hands_stream_t stream) {
  if (!feat || !out || B <= 0 || C % 4 || out_stride % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(sumpool_kernel, dim3(hands_grid_1d((long long)B * C / 4, 64)), dim3(64), 0,
                     (hipStream_t)stream, (const float4*)feat, out, B, HW, C / 4, out_stride);
  HANDS_LAUNCH_CHECK();
}

int hands_avgpool_nhwc_f32(const float* feat, float* out, int B, int HW, int C, int out_stride,
                           hands_stream_t stream) {
  if (!feat || !out || B <= 0 || HW <= 0 || C % 4 || out_stride % 4) return HANDS_EINVAL;
  hipLaunchKernelGGL(avgpool_kernel, dim3(hands_grid_1d((long long)B * C / 4, 64)), dim3(64), 0,
                     (hipStream_t)stream, (const float4*)feat, out, B, HW, C / 4, out_stride);
  HANDS_LAUNCH_CHECK();
}

int hands_image_posenc_nhwc_f32(const float* img_nchw, const float* center_angle, const float* corner_angle, float* out,
                                int B, int H, int W, int n_freq, int mode, int Cpad, hands_stream_t stream) {
  if (!img_nchw || !out || B <= 0 || H <= 0 || W <= 0 || n_freq < 1 || n_freq > 16 || mode < 1 || mode > 3 || Cpad % 4)
    return HANDS_EINVAL;
  if (((mode & 1) && !center_angle) || ((mode & 2) && !corner_angle)) return HANDS_EINVAL;
  if (Cpad < 3 + ((mode & 1) ? 4 * n_freq : 0) + ((mode & 2) ? 16 * n_freq : 0)) return HANDS_EINVAL;
  const long long n = (long long)B * H * W * (Cpad / 4);
  hipLaunchKernelGGL(image_posenc_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, img_nchw,
                     center_angle, corner_angle, (float4*)out, B, H * W, n_freq, mode, Cpad / 4);
  HANDS_LAUNCH_CHECK();
}


int hands_dense_posenc_f32(const float* angle, const float* mask, const float* img_nchw, float* out, int B, int Ca, int Hs, int Ws,
                           int n_freq, int R, int Ho, int Wo, int ld, int c_off, hands_stream_t stream) {
  if (!angle || !mask || !out || B <= 0 || Ca <= 0 || Hs <= 0 || Ws <= 0 || n_freq < 0 || n_freq > 16 || R <= 0 || Ho <= 0 ||
      Wo <= 0 || c_off < 0)
    return HANDS_EINVAL;
  const int Cenc = n_freq ? 2 * n_freq * Ca : Ca;
  if (ld < c_off + Cenc || (img_nchw && c_off != 3)) return HANDS_EINVAL;
  const long long n = (long long)B * Ho * Wo * (img_nchw ? ld : Cenc);
  hipLaunchKernelGGL(dense_posenc_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, angle, mask, img_nchw, out,
                     B, Ca, Hs, Ws, n_freq, R, Ho, Wo, ld, c_off, Cenc);
  HANDS_LAUNCH_CHECK();
}

int hands_concat_nhwc_f32(const float* a, int lda, int Ca, const float* add, int ld_add, const float* extra,
                          long long extra_batch_stride, int Cb, float* out, int ld, int B, int Bg, int HW, hands_stream_t stream) {
  if (!a || !out || B <= 0 || Bg <= 0 || HW <= 0 || Ca <= 0 || lda < Ca || Cb < 0 || (Cb && !extra) || ld < Ca + Cb ||
      (add && ld_add < Ca))
    return HANDS_EINVAL;
  const long long n = (long long)B * HW * ld;
  hipLaunchKernelGGL(concat_nhwc_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, a, lda, Ca, add, ld_add,
                     extra, extra_batch_stride, Cb, out, ld, B, Bg, HW);
  HANDS_LAUNCH_CHECK();
}

int hands_upsample_bilinear_ac_f32(const float* x, float* out, int B, int h, int w, int H, int W, int C, hands_stream_t stream) {
  if (!x || !out || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return HANDS_EINVAL;
  const long long n = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(upsample_bilinear_ac_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)x, (float4*)out, B, h, w, H, W, C / 4);
  HANDS_LAUNCH_CHECK();
}

int hands_rot_leftmul_f32(float* rotmat, const float* rot, int B, hands_stream_t stream) {
  if (!rotmat || !rot || B <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(rot_leftmul_kernel, dim3(hands_grid_1d(B, 256)), dim3(256), 0, (hipStream_t)stream, rotmat, rot, B);
  HANDS_LAUNCH_CHECK();
}

int hands_perspective_correction_f32(float* rot_swapped, float* rotmat, const float* center_angle, const int64_t* is_flipped,
                                     int Bg, hands_stream_t stream) {
  if (!rot_swapped || !rotmat || !center_angle || !is_flipped || Bg <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(persp_correct_kernel, dim3(hands_grid_1d(2LL * Bg, 256)), dim3(256), 0, (hipStream_t)stream, rot_swapped,
                     rotmat, center_angle, (const long long*)is_flipped, Bg);
  HANDS_LAUNCH_CHECK();
}

int hands_kpe_concat_f32(const float* crop, const float* glb, const float* center_angle,
                         const float* corner_angle, float* out, int B2, int Bg, int HW, int C,
                         int n_freq, hands_stream_t stream) {
  if (!crop || !center_angle || !corner_angle || !out || B2 <= 0 || Bg <= 0 || C % 4 ||
      n_freq < 1 || n_freq > 16)
    return HANDS_EINVAL;
  const long long n = (long long)B2 * HW * (C / 4 + 5 * n_freq);
  hipLaunchKernelGGL(kpe_concat_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)crop, (const float4*)glb, center_angle, corner_angle, (float4*)out,
                     B2, Bg, HW, C / 4, n_freq);
  HANDS_LAUNCH_CHECK();
}

int hands_hmr_init_f32(float* state, const float* cam_init, int B, int ld, int F, hands_stream_t stream) {
  if (!state || !cam_init || B <= 0 || ld < F + 112) return HANDS_EINVAL;
  hipLaunchKernelGGL(hmr_init_kernel, dim3(hands_grid_1d((long long)B * 112, 256)), dim3(256), 0,
                     (hipStream_t)stream, state, cam_init, B, ld, F);
  HANDS_LAUNCH_CHECK();
}

int hands_rot6d_to_matrix_f32(const float* pose6d, int ld6, float* rotmat, int B, hands_stream_t stream) {
  if (!pose6d || !rotmat || B <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(rot6d_kernel, dim3(hands_grid_1d((long long)B * 16, 256)), dim3(256), 0,
                     (hipStream_t)stream, pose6d, ld6, rotmat, B);
  HANDS_LAUNCH_CHECK();
}

int hands_flip_swap_f32(const int64_t* is_flipped, const float* rotmat, const float* shape,
                        const float* cam, const float* cam_init, float* rotmat_out, float* shape_out,
                        float* cam_out, float* cam_init_out, int Bg, hands_stream_t stream) {
  if (!is_flipped || !rotmat || !shape || !cam || !cam_init || !rotmat_out || !shape_out || !cam_out ||
      !cam_init_out || Bg <= 0)
    return HANDS_EINVAL;
  hipLaunchKernelGGL(flip_swap_kernel, dim3(hands_grid_1d((long long)Bg * 32, 256)), dim3(256), 0,
                     (hipStream_t)stream, is_flipped, rotmat, shape, cam, cam_init, rotmat_out,
                     shape_out, cam_out, cam_init_out, Bg);
  HANDS_LAUNCH_CHECK();
}

int hands_grasp_input_f32(const float* shape, int ld_shape, const float* rotmat, const float* feat_vec,
                          float* out, int B2, int Bg, int F, int ld_out, hands_stream_t stream) {
  if (!shape || !rotmat || !feat_vec || !out || B2 <= 0 || Bg <= 0 || ld_out < F + 154)
    return HANDS_EINVAL;
  hipLaunchKernelGGL(grasp_input_kernel, dim3(hands_grid_1d((long long)B2 * ld_out, 256)), dim3(256), 0,
                     (hipStream_t)stream, shape, ld_shape, rotmat, feat_vec, out, B2, Bg, F, ld_out);
  HANDS_LAUNCH_CHECK();
}

int hands_matrix_to_axis_angle_f32(const float* rotmat, float* axis_angle, long long n, hands_stream_t stream) {
  if (!rotmat || !axis_angle || n <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(matrix_to_axis_angle_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     rotmat, axis_angle, n);
  HANDS_LAUNCH_CHECK();
}

int hands_axis_angle_to_matrix_f32(const float* axis_angle, float* rotmat, long long n, hands_stream_t stream) {
  if (!rotmat || !axis_angle || n <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(axis_angle_to_matrix_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     axis_angle, rotmat, n);
  HANDS_LAUNCH_CHECK();
}

}  // extern "C"
