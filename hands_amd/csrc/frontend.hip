// frontend.hip -- the step in front of the hot path: hand boxes from 2-D joints,
// the square crop windows, their KPE angles, and the cubic crop-resize + clip + ImageNet normalise.
//
// Replaces (test-time branch, no augmentation):
//   src/datasets/hands_light_dataset.py:137-152 (boxes), :165-178 (crops + Normalize), :256-279 (angles)
//   common/data_utils.py:495-509 crop_and_pad, :56-91 gen_trans_from_patch_cv,
//   :423-460 generate_patch_image_clean = cv2.warpAffine(..., INTER_CUBIC), constant-0 border.
// cv2 is a third-party dependency that is absent from the reference tree: the warp follows OpenCV's
// published algorithm (double-precision inverse map, AB_BITS=10 / INTER_BITS=5 fixed-point source
// coordinates, float32 cubic weights with A=-0.75, 4x4 taps).  HBM-bound gather; one thread per
// output pixel, the three colour planes share the coordinates and weights.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"

namespace {

struct BoxArgs {
  const float* j2d[2];
  const float* K;
  int32_t* bbox[2];
  int32_t* bbox_og[2];
  float* trans[2];
  float* center[2];
  float* corner[2];
  int ld, B, img_res, out_res;
  double scale;
};

__device__ __forceinline__ float clipf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__global__ void frontend_boxes_kernel(BoxArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * a.B) return;
  const int hand = t & 1, b = t >> 1;
  const float* j = a.j2d[hand] + (size_t)b * 21 * a.ld;
  const float hi = (float)(a.img_res - 1);
  float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
  for (int k = 0; k < 21; ++k) {
    const float px = ((j[k * a.ld + 0] + 1.0f) / 2.0f) * hi;
    const float py = ((j[k * a.ld + 1] + 1.0f) / 2.0f) * hi;
    mnx = fminf(mnx, px); mxx = fmaxf(mxx, px);
    mny = fminf(mny, py); mxy = fmaxf(mxy, py);
  }
  mnx = clipf(mnx, 0.f, hi); mny = clipf(mny, 0.f, hi); mxx = clipf(mxx, 0.f, hi); mxy = clipf(mxy, 0.f, hi);
  const int x0 = (int)mnx, y0 = (int)mny, w = (int)(mxx - mnx), h = (int)(mxy - mny);   // astype(int16): truncation
  const bool none = (w == 0) || (h == 0);
  int32_t* og = a.bbox_og[hand] + 4 * b;
  int32_t* nb = a.bbox[hand] + 4 * b;
  double pcx, pcy, psz;
  if (none) {
    og[0] = 0; og[1] = 0; og[2] = a.img_res - 1; og[3] = a.img_res - 1;
    nb[0] = 0; nb[1] = 0; nb[2] = a.img_res - 1; nb[3] = a.img_res - 1;
    pcx = pcy = a.img_res / 2.0; psz = (double)a.img_res;
  } else {
    og[0] = x0; og[1] = y0; og[2] = w; og[3] = h;
    const int x1 = x0 + w, y1 = y0 + h;
    const int xm = (x0 + x1) >> 1, ym = (y0 + y1) >> 1;
    const int size = max(w, h);
    psz = (double)size * a.scale;
    const double half = floor(psz / 2.0);
    const double lim = (double)(a.img_res - 1);
    nb[0] = (int)fmin(fmax((double)xm - half, 0.0), lim);
    nb[1] = (int)fmin(fmax((double)ym - half, 0.0), lim);
    nb[2] = (int)fmin(fmax((double)xm + half, 0.0), lim);
    nb[3] = (int)fmin(fmax((double)ym + half, 0.0), lim);
    pcx = (double)xm; pcy = (double)ym;
  }
  // gen_trans_from_patch_cv with rot = 0: three float32 point pairs, affine solved in double.
  {
    const float cx = (float)pcx, cy = (float)pcy, hs = (float)(psz * 0.5);
    const float sx2 = cx + hs, sy1 = cy + hs;                    // src_center + rightdir / downdir (float32 adds)
    const float dc = (float)(a.out_res * 0.5);
    const float d2 = dc + dc;
    const double ax = ((double)d2 - (double)dc) / ((double)sx2 - (double)cx);
    const double ay = ((double)d2 - (double)dc) / ((double)sy1 - (double)cy);
    float* tr = a.trans[hand] + 6 * b;
    tr[0] = (float)ax; tr[1] = 0.f; tr[2] = (float)((double)dc - ax * (double)cx);
    tr[3] = 0.f; tr[4] = (float)ay; tr[5] = (float)((double)dc - ay * (double)cy);
  }
  // KPE angles of the crop window (center: float64 atan2 stored as float32)
  {
    // K == NULL: args.no_intrx (hands_light_dataset.py:247-253): the encodings use the float64 stand-in
    // [[res/2, 0, res/2], [0, res/2, res/2], [0, 0, 1]] instead of the camera's intrinsics
    const float* K = a.K ? a.K + 9 * b : nullptr;
    const double half = a.img_res / 2.0;
    const double fx = K ? (double)K[0] : half, fy = K ? (double)K[4] : half, px = K ? (double)K[2] : half, py = K ? (double)K[5] : half;
    const double bx0 = nb[0], by0 = nb[1], bx1 = nb[2], by1 = nb[3];
    float* c = a.center[hand] + 2 * b;
    c[0] = (float)atan2((bx0 + bx1) / 2.0 - px, fx);
    c[1] = (float)atan2((by0 + by1) / 2.0 - py, fy);
    float* q = a.corner[hand] + 8 * b;
    float ax0, ax1, ay0, ay1;
    if (K) {
      // corners: numpy keeps int16 - float32 in float32, so the reference evaluates these in float32
      ax0 = atan2f((float)nb[0] - K[2], K[0]); ax1 = atan2f((float)nb[2] - K[2], K[0]);
      ay0 = atan2f((float)nb[1] - K[5], K[4]); ay1 = atan2f((float)nb[3] - K[5], K[4]);
    } else {
      // int16 - float64 is float64: double atan2, rounded once
      ax0 = (float)atan2(bx0 - px, fx); ax1 = (float)atan2(bx1 - px, fx);
      ay0 = (float)atan2(by0 - py, fy); ay1 = (float)atan2(by1 - py, fy);
    }
    q[0] = ax0; q[1] = ay0; q[2] = ax0; q[3] = ay1; q[4] = ax1; q[5] = ay0; q[6] = ax1; q[7] = ay1;
  }
}

// OpenCV interpolateCubic, float32, A = -0.75
__device__ __forceinline__ void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

__device__ __forceinline__ int sat_int(double v) {
  v = rint(v);
  return v >= 2147483647.0 ? 2147483647 : (v <= -2147483648.0 ? (int)0x80000000 : (int)v);
}

struct WarpArgs {
  const float* src;     // (B, 3, H, W)
  const float* trans;   // (B, 6) forward map src -> dst, or nullptr = identity
  float* out;           // (B, 3, Ho, Wo)
  int B, H, W, Ho, Wo;
  float mean[3], stdv[3];
};

__global__ void __launch_bounds__(256) warp_cubic_norm_kernel(WarpArgs a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= a.Ho * a.Wo) return;
  const int y = p / a.Wo, x = p - y * a.Wo;
  double M[6] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0};
  if (a.trans) {
    const float* t = a.trans + 6 * b;
#pragma unroll
    for (int i = 0; i < 6; ++i) M[i] = (double)t[i];
  }
  {  // invert (imgwarp.cpp warpAffine)
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
  }
  const int adelta = sat_int(M[0] * (double)x * 1024.0);
  const int bdelta = sat_int(M[3] * (double)x * 1024.0);
  const int X0 = sat_int((M[1] * (double)y + M[2]) * 1024.0) + 16;
  const int Y0 = sat_int((M[4] * (double)y + M[5]) * 1024.0) + 16;
  const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  int sx = X >> 5, sy = Y >> 5;
  sx = min(max(sx, -32768), 32767) - 1;
  sy = min(max(sy, -32768), 32767) - 1;
  float wx[4], wy[4];
  cubic_coeffs((float)(X & 31) * (1.0f / 32.0f), wx);
  cubic_coeffs((float)(Y & 31) * (1.0f / 32.0f), wy);
  const size_t plane = (size_t)a.H * a.W;
  const float* S0 = a.src + (size_t)b * 3 * plane;
  float* O = a.out + (size_t)b * 3 * a.Ho * a.Wo + p;
  const bool interior = (unsigned)sx < (unsigned)max(a.W - 3, 0) && (unsigned)sy < (unsigned)max(a.H - 3, 0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* S = S0 + c * plane;
    float sum = 0.f;
    if (interior) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* R = S + (size_t)(sy + i) * a.W + sx;
        const float row = R[0] * (wy[i] * wx[0]) + R[1] * (wy[i] * wx[1]) + R[2] * (wy[i] * wx[2]) + R[3] * (wy[i] * wx[3]);
        sum = (i == 0) ? row : sum + row;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int yy = sy + i;
        if ((unsigned)yy >= (unsigned)a.H) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int xx = sx + j;
          if ((unsigned)xx < (unsigned)a.W) sum += S[(size_t)yy * a.W + xx] * (wy[i] * wx[j]);
        }
      }
    }
    sum = fminf(fmaxf(sum, 0.f), 1.f);                     // np.clip(img_crop, 0, 1)
    O[(size_t)c * a.Ho * a.Wo] = (sum - a.mean[c]) / a.stdv[c];   // torchvision Normalize
  }
}


// ---- per-pixel angle maps of a crop window (pos_enc 'dense' / 'dense_latent' / 'cam_conv') ----------------------------------
// hands_light_dataset.py:281-333: for the window's pixels x in [x0, x1], y in [y0, y1] (first map index = x: meshgrid 'ij')
// angle = atan2(x - cx, fx), atan2(y - cy, fy) in double (int64 grid - float32 intrinsic -> float64), stored as float32 in the
// top-left corner of a zero (img_res, img_res) map; 'cam_conv' adds the centred offsets x - cx, y - cy and 2 x / img_res - 1,
// 2 y / img_res - 1; the mask marks the window.
__global__ void dense_maps_kernel(const int32_t* __restrict__ bbox, const float* __restrict__ K, float* __restrict__ angle,
                                  float* __restrict__ mask, int B, int R, int nch) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= R * R) return;
  const int i = p / R, j = p - i * R;
  const int x0 = bbox[b * 4], y0 = bbox[b * 4 + 1], x1 = bbox[b * 4 + 2], y1 = bbox[b * 4 + 3];
  const bool in = i <= x1 - x0 && j <= y1 - y0;
  const float* Kb = K ? K + b * 9 : nullptr;             // NULL: args.no_intrx, as in frontend_boxes_kernel
  const double half = R / 2.0;
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (in) {
    const double dx = (double)(x0 + i) - (Kb ? (double)Kb[2] : half), dy = (double)(y0 + j) - (Kb ? (double)Kb[5] : half);
    v[0] = (float)atan2(dx, Kb ? (double)Kb[0] : half);
    v[1] = (float)atan2(dy, Kb ? (double)Kb[4] : half);
    v[2] = (float)dx;
    v[3] = (float)dy;
    v[4] = (float)(2.0 * (double)(x0 + i) / (double)R - 1.0);
    v[5] = (float)(2.0 * (double)(y0 + j) / (double)R - 1.0);
  }
  for (int c = 0; c < nch; ++c) angle[((size_t)b * nch + c) * R * R + p] = v[c];
  mask[(size_t)b * R * R + p] = in ? 1.f : 0.f;
}

}  // namespace

extern "C" int hands_frontend_dense_maps_f32(const int32_t* bbox, const float* K, float* angle, float* mask, int B, int img_res,
                                             int nch, hands_stream_t stream) {
  if (!bbox || !angle || !mask || B <= 0 || B > 65535 || img_res < 1 || (nch != 2 && nch != 6)) return HANDS_EINVAL;   // K may be NULL
  hipLaunchKernelGGL(dense_maps_kernel, dim3((img_res * img_res + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, bbox, K, angle,
                     mask, B, img_res, nch);
  HANDS_LAUNCH_CHECK();
}

extern "C" int hands_frontend_boxes_f32(const float* j2d_r, const float* j2d_l, int ld, const float* K, int B,
                                        int img_res, int out_res, double bbox_scale,
                                        int32_t* bbox_r, int32_t* bbox_l, int32_t* bbox_og_r, int32_t* bbox_og_l,
                                        float* trans_r, float* trans_l, float* center_r, float* center_l,
                                        float* corner_r, float* corner_l, hands_stream_t stream) {
  if (!j2d_r || !j2d_l || !bbox_r || !bbox_l || !bbox_og_r || !bbox_og_l || !trans_r || !trans_l ||
      !center_r || !center_l || !corner_r || !corner_l)
    return HANDS_EINVAL;
  if (B <= 0 || ld < 2 || img_res < 2 || out_res < 1 || !(bbox_scale > 0.0)) return HANDS_EINVAL;
  BoxArgs a;
  a.j2d[0] = j2d_r; a.j2d[1] = j2d_l; a.K = K;
  a.bbox[0] = bbox_r; a.bbox[1] = bbox_l; a.bbox_og[0] = bbox_og_r; a.bbox_og[1] = bbox_og_l;
  a.trans[0] = trans_r; a.trans[1] = trans_l; a.center[0] = center_r; a.center[1] = center_l;
  a.corner[0] = corner_r; a.corner[1] = corner_l;
  a.ld = ld; a.B = B; a.img_res = img_res; a.out_res = out_res; a.scale = bbox_scale;
  hipLaunchKernelGGL(frontend_boxes_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
  HANDS_LAUNCH_CHECK();
}

extern "C" int hands_warp_affine_cubic_norm_f32(const float* src, const float* trans, float* out, int B, int H, int W,
                                                int Ho, int Wo, const float* mean3, const float* std3,
                                                hands_stream_t stream) {
  if (!src || !out || !mean3 || !std3) return HANDS_EINVAL;   // mean3/std3 are HOST pointers (3 floats each)
  if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || B > 65535) return HANDS_EINVAL;
  WarpArgs a;
  a.src = src; a.trans = trans; a.out = out; a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  for (int i = 0; i < 3; ++i) { a.mean[i] = mean3[i]; a.stdv[i] = std3[i]; }
  hipLaunchKernelGGL(warp_cubic_norm_kernel, dim3((Ho * Wo + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a);
  HANDS_LAUNCH_CHECK();
}
