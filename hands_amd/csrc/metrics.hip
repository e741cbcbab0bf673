// metrics.hip -- evaluation metrics on device for the gathered predictions:
// root-aligned MPJPE, Procrustes-aligned MPJPE (3x3 SVD per hand), hand-to-hand MRRPE, 2-D pixel error.
// Reference: src/utils/eval_modules.py:97-134,136-219,320-343,386-428; common/metrics.py:23-55;
// common/torch_utils.py:14-19 (nanmean).  One thread per sample; the work is ~2 kFLOP per hand.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"

namespace {

constexpr int NJ = 21;

// mean over joints of || (gt_j - gt_0) - (pr_j - pr_0) ||     (eval_modules.py:107-121)
__device__ float mpjpe_ra(const float* gt, const float* pr) {
  float s = 0.f;
  for (int j = 0; j < NJ; ++j) {
    float d2 = 0.f;
    for (int c = 0; c < 3; ++c) {
      const float d = (gt[3 * j + c] - gt[c]) - (pr[3 * j + c] - pr[c]);
      d2 += d * d;
    }
    s += sqrtf(d2);
  }
  return s / (float)NJ;
}

// symmetric 3x3 eigen-decomposition by cyclic Jacobi (fp64): A = V diag(w) V^T
__device__ void jacobi3(double A[3][3], double V[3][3], double w[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {           // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {           // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

// Procrustes-aligned mean joint error of ROOT-ALIGNED joints (eval_modules.py:136-219)
__device__ float mpjpe_pa(const float* gt, const float* pr) {
  double S1[NJ][3], S2[NJ][3], mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
  for (int j = 0; j < NJ; ++j)
    for (int c = 0; c < 3; ++c) {
      S1[j][c] = (double)(pr[3 * j + c] - pr[c]);      // root alignment in fp32, as the reference
      S2[j][c] = (double)(gt[3 * j + c] - gt[c]);
      mu1[c] += S1[j][c]; mu2[c] += S2[j][c];
    }
  for (int c = 0; c < 3; ++c) { mu1[c] /= NJ; mu2[c] /= NJ; }
  double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0;
  for (int j = 0; j < NJ; ++j)
    for (int a = 0; a < 3; ++a) {
      const double x1 = S1[j][a] - mu1[a];
      var1 += x1 * x1;
      for (int b = 0; b < 3; ++b) K[a][b] += x1 * (S2[j][b] - mu2[b]);
    }
  // K = U S V^T; eigen-decompose K^T K = V S^2 V^T
  double A[3][3], V[3][3], w[3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) A[a][b] = K[0][a] * K[0][b] + K[1][a] * K[1][b] + K[2][a] * K[2][b];
  jacobi3(A, V, w);
  int o[3] = {0, 1, 2};                                  // sort singular values descending
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (w[o[j]] > w[o[i]]) { const int t = o[i]; o[i] = o[j]; o[j] = t; }
  double Vs[3][3], U[3][3];
  for (int i = 0; i < 3; ++i)
    for (int r = 0; r < 3; ++r) Vs[r][i] = V[r][o[i]];
  for (int i = 0; i < 2; ++i) {                          // U_i = K V_i / |K V_i|
    double u[3], n = 0;
    for (int r = 0; r < 3; ++r) { u[r] = K[r][0] * Vs[0][i] + K[r][1] * Vs[1][i] + K[r][2] * Vs[2][i]; n += u[r] * u[r]; }
    n = sqrt(n);
    for (int r = 0; r < 3; ++r) U[r][i] = n > 0 ? u[r] / n : (r == i ? 1.0 : 0.0);
  }
  U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];      // U_3 = U_1 x U_2  (det U = +1)
  U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
  U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  const double detV = Vs[0][0] * (Vs[1][1] * Vs[2][2] - Vs[1][2] * Vs[2][1]) -
                      Vs[0][1] * (Vs[1][0] * Vs[2][2] - Vs[1][2] * Vs[2][0]) +
                      Vs[0][2] * (Vs[1][0] * Vs[2][1] - Vs[1][1] * Vs[2][0]);
  const double z = detV >= 0 ? 1.0 : -1.0;               // R = V diag(1,1,det V) U^T maximises trace(R K)
  double R[3][3], tr = 0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) R[a][b] = Vs[a][0] * U[b][0] + Vs[a][1] * U[b][1] + z * Vs[a][2] * U[b][2];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) tr += R[a][b] * K[b][a];
  const double scale = tr / var1;
  double t[3];
  for (int a = 0; a < 3; ++a) t[a] = mu2[a] - scale * (R[a][0] * mu1[0] + R[a][1] * mu1[1] + R[a][2] * mu1[2]);
  double s = 0;
  for (int j = 0; j < NJ; ++j) {
    double d2 = 0;
    for (int a = 0; a < 3; ++a) {
      const double h = scale * (R[a][0] * S1[j][0] + R[a][1] * S1[j][1] + R[a][2] * S1[j][2]) + t[a];
      d2 += (S2[j][a] - h) * (S2[j][a] - h);
    }
    s += sqrt(d2);
  }
  return (float)(s / NJ);
}

__device__ __forceinline__ float nanmean2(float a, float b) {
  const bool na = isnan(a), nb = isnan(b);
  return ((na ? 0.f : a) + (nb ? 0.f : b)) / (float)((na ? 0 : 1) + (nb ? 0 : 1));
}

__global__ void eval_metrics_kernel(hands_eval_in in, hands_eval_out out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float NaN = __builtin_nanf("");
  const float isv = in.is_valid[b];
  const float rv = in.right_valid[b] * isv, lv = in.left_valid[b] * isv;
  const float* gr = in.gt_j3d_r + (long long)b * NJ * 3;
  const float* gl = in.gt_j3d_l + (long long)b * NJ * 3;
  const float* pr = in.pred_j3d_r + (long long)b * NJ * 3;
  const float* pl = in.pred_j3d_l + (long long)b * NJ * 3;
  const float ra_r = rv != 0.f ? mpjpe_ra(gr, pr) : NaN;
  const float ra_l = lv != 0.f ? mpjpe_ra(gl, pl) : NaN;
  out.mpjpe_ra_h[b] = nanmean2(ra_r, ra_l) * 1000.0f;
  const float pa_r = mpjpe_pa(gr, pr) * rv, pa_l = mpjpe_pa(gl, pl) * lv;   // invalid -> 0, not nan (:217)
  out.mpjpe_pa_ra_r[b] = pa_r * 1000.0f;
  out.mpjpe_pa_ra_l[b] = pa_l * 1000.0f;
  out.mpjpe_pa_ra_h[b] = nanmean2(pa_r, pa_l) * 1000.0f;
  float d2 = 0.f;
  for (int c = 0; c < 3; ++c) {
    const float d = (pl[c] - pr[c]) - (gl[c] - gr[c]);
    d2 += d * d;
  }
  out.mrrpe_rl[b] = (lv * rv) != 0.f ? sqrtf(d2) * 1000.0f : NaN;
  for (int j = 0; j < NJ; ++j) {
    const long long i = (long long)b * NJ + j;
    const float dxr = in.gt_j2d_r[2 * i] - in.pred_j2d_r[2 * i], dyr = in.gt_j2d_r[2 * i + 1] - in.pred_j2d_r[2 * i + 1];
    const float dxl = in.gt_j2d_l[2 * i] - in.pred_j2d_l[2 * i], dyl = in.gt_j2d_l[2 * i + 1] - in.pred_j2d_l[2 * i + 1];
    out.pix_err_r[i] = in.joints_valid_r[i] * rv != 0.f ? sqrtf(dxr * dxr + dyr * dyr) : NaN;
    out.pix_err_l[i] = in.joints_valid_l[i] * lv != 0.f ? sqrtf(dxl * dxl + dyl * dyl) : NaN;
  }
}

// process_data_light (src/callbacks/process/process_arctic.py:41-73): camera-space targets from the
// canonical GT MANO output and the annotated camera-space joints.
__global__ void gt_targets_kernel(const float* __restrict__ joints, const float* __restrict__ verts,
                                  const float* __restrict__ j3d_full, const float* __restrict__ Kmat, float img_res,
                                  float* __restrict__ v3d_cam, float* __restrict__ cam_t, float* __restrict__ cam_t_wp,
                                  int B, int NV) {
  __shared__ float tr[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* jc = joints + (long long)b * NJ * 3;
  const float* jf = j3d_full + (long long)b * NJ * 3;
  if (tid < 3) {
    float s = 0.f;                                     // Tr0 = (j3d.full - joints).mean(dim=1)
    for (int j = 0; j < NJ; ++j) s += jf[3 * j + tid] - jc[3 * j + tid];
    tr[tid] = s / (float)NJ;
    cam_t[b * 3 + tid] = jf[tid] - jc[tid];            // root_cam - root_cano
  }
  if (tid == 0) {
    const float f = (Kmat[b * 9 + 0] + Kmat[b * 9 + 4]) / 2.0f;
    const float tz = jf[2] - jc[2];
    cam_t_wp[b * 3 + 0] = 2.0f * f / (img_res * tz + 1e-9f);    // camera.py:10-29
    cam_t_wp[b * 3 + 1] = jf[0] - jc[0];
    cam_t_wp[b * 3 + 2] = jf[1] - jc[1];
  }
  __syncthreads();
  const float* v = verts + (long long)b * NV * 3;
  float* o = v3d_cam + (long long)b * NV * 3;
  for (int i = tid; i < NV * 3; i += blockDim.x) o[i] = v[i] + tr[i % 3];
}

// unormalize_kp2d (common/data_utils.py:368-373): 0.5 * res * (x + 1)
__global__ void unnormalize_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, float img_res) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = 0.5f * img_res * (x[i] + 1.0f);
}

}  // namespace

extern "C" int hands_gt_targets_f32(const float* joints, const float* verts, const float* j3d_full, const float* K,
                                    float img_res, float* v3d_cam, float* cam_t, float* cam_t_wp, int B, int NV,
                                    hands_stream_t stream) {
  if (!joints || !verts || !j3d_full || !K || !v3d_cam || !cam_t || !cam_t_wp || B <= 0 || NV <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(gt_targets_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, joints, verts, j3d_full, K, img_res,
                     v3d_cam, cam_t, cam_t_wp, B, NV);
  HANDS_LAUNCH_CHECK();
}

extern "C" int hands_unnormalize_kp2d_f32(const float* x, float* out, long long n, float img_res, hands_stream_t stream) {
  if (!x || !out || n <= 0) return HANDS_EINVAL;
  hipLaunchKernelGGL(unnormalize_kernel, dim3(hands_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n, img_res);
  HANDS_LAUNCH_CHECK();
}

extern "C" int hands_eval_metrics_f32(const hands_eval_in* in, const hands_eval_out* out, int B, hands_stream_t stream) {
  if (!in || !out || B <= 0) return HANDS_EINVAL;
  const void* const* pi = reinterpret_cast<const void* const*>(in);
  for (size_t i = 0; i < sizeof(hands_eval_in) / sizeof(void*); ++i)
    if (!pi[i]) return HANDS_EINVAL;
  const void* const* po = reinterpret_cast<const void* const*>(out);
  for (size_t i = 0; i < sizeof(hands_eval_out) / sizeof(void*); ++i)
    if (!po[i]) return HANDS_EINVAL;
  hipLaunchKernelGGL(eval_metrics_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, *in, *out, B);
  HANDS_LAUNCH_CHECK();
}
