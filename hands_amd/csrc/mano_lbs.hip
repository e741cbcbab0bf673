// mano_lbs.hip -- MANO linear-blend-skinning layer for gfx950, split around the MFMA blend GEMM.
//
// Reference: MANOHead.forward (src/nets/hand_heads/mano_head.py:21-65), which calls
//   common/rot.py:118-193 (matrix -> quaternion -> axis-angle),
//   smplx.MANO.forward / smplx.lbs.{lbs,batch_rodrigues,blend_shapes,vertices2joints,
//   batch_rigid_transform} (third party; algorithm restated in SURVEY.md section 8a-K),
//   common/camera.py:456-474, common/transforms.py:316-329, common/data_utils.py:361-365.
//
//   mano_pose_kernel : one wave = 4 hands x 16 joints.  R -> aa -> (+pose_mean) -> Rodrigues,
//                      pose feature row for the blend GEMM, J(beta) = J_template + J_shapedirs beta,
//                      forward kinematics down the 3-deep tree, skinning transforms A_j.
//   [blend GEMM]     : v_posed = v_template + [beta | pose_feature] @ [shapedirs ; posedirs]
//                      runs in conv_igemm.hip on fp32 MFMA (M = hands, N = 2334, K = 145).
//   mano_skin_kernel : one block per hand; per-vertex T = sum_j w_vj A_j, apply to v_posed,
//                      fingertip joints, weak-perspective camera, projection, normalisation.
// HBM traffic per hand: reads 576 B rotmat + 40 B beta + 9.3 KB v_posed, writes 2 x 9.3 KB
// vertices + ~1 KB joints; lbs_weights (50 KB) and the blend matrix (1.5 MB) stay in L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hands_hip.h"
#include "common.h"
#include "rot_device.h"

namespace {

constexpr int NJ = 16;
constexpr int NV = 778;
constexpr int HANDS_PER_BLOCK = 4;

__device__ __forceinline__ int parent_of(int j) { return j == 0 ? -1 : ((j - 1) % 3 == 0 ? 0 : j - 1); }
__device__ __forceinline__ int depth_of(int j) { return j == 0 ? 0 : (j - 1) % 3 + 1; }

template <bool AA_INPUT>
__global__ void __launch_bounds__(64) mano_pose_kernel(hands_mano_consts c, const float* __restrict__ rotmat,
                                                       const float* __restrict__ betas, int ld_betas,
                                                       float* __restrict__ blend_in, int ld_blend,
                                                       float* __restrict__ A, float* __restrict__ joints16,
                                                       int B) {
  __shared__ float sR[HANDS_PER_BLOCK][NJ][9];    // local joint rotations (after Rodrigues)
  __shared__ float sJ[HANDS_PER_BLOCK][NJ][3];    // rest joints J(beta)
  __shared__ float sG[HANDS_PER_BLOCK][NJ][12];   // global transforms [R | t], row-major 3x4
  const int h = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int b = blockIdx.x * HANDS_PER_BLOCK + h;
  const bool live = b < B;

  if (live) {
    float aa[3], R[9];
    if constexpr (AA_INPUT) {        // ground-truth MANO parameters are axis-angle already (process_arctic.py:16-21)
      const float* src = rotmat + ((long long)b * NJ + j) * 3;
      aa[0] = src[0]; aa[1] = src[1]; aa[2] = src[2];
    } else {
      hands::matrix_to_axis_angle(rotmat + ((long long)b * NJ + j) * 9, aa);
    }
    aa[0] += c.pose_mean[3 * j + 0];
    aa[1] += c.pose_mean[3 * j + 1];
    aa[2] += c.pose_mean[3 * j + 2];
    hands::rodrigues(aa, R);
#pragma unroll
    for (int e = 0; e < 9; ++e) sR[h][j][e] = R[e];
    const float* be = betas + (long long)b * ld_betas;
    float* row = blend_in + (long long)b * ld_blend;
    if (j >= 1) {
#pragma unroll
      for (int e = 0; e < 9; ++e) row[10 + (j - 1) * 9 + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    if (j < 10) row[j] = be[j];
    for (int e = 145 + j; e < ld_blend; e += NJ) row[e] = 0.f;
    // J_j = J_template[j] + J_shapedirs[3j+c, :] . beta
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      float acc = c.J_template[3 * j + cc];
      const float* js = c.J_shapedirs + (3 * j + cc) * 10;
#pragma unroll
      for (int k = 0; k < 10; ++k) acc += js[k] * be[k];
      sJ[h][j][cc] = acc;
    }
  }
  __syncthreads();

  // forward kinematics: G_0 = [R_0 | J_0]; G_j = G_p * [R_j | J_j - J_p]
  const int dj = depth_of(j), pj = parent_of(j);
  for (int level = 0; level < 4; ++level) {
    if (live && dj == level) {
      float* g = sG[h][j];
      const float* r = sR[h][j];
      if (level == 0) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          g[rr * 4 + 0] = r[rr * 3 + 0]; g[rr * 4 + 1] = r[rr * 3 + 1]; g[rr * 4 + 2] = r[rr * 3 + 2];
          g[rr * 4 + 3] = sJ[h][0][rr];
        }
      } else {
        const float* gp = sG[h][pj];
        const float t0 = sJ[h][j][0] - sJ[h][pj][0], t1 = sJ[h][j][1] - sJ[h][pj][1],
                    t2 = sJ[h][j][2] - sJ[h][pj][2];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const float p0 = gp[rr * 4 + 0], p1 = gp[rr * 4 + 1], p2 = gp[rr * 4 + 2], p3 = gp[rr * 4 + 3];
          g[rr * 4 + 0] = p0 * r[0] + p1 * r[3] + p2 * r[6];
          g[rr * 4 + 1] = p0 * r[1] + p1 * r[4] + p2 * r[7];
          g[rr * 4 + 2] = p0 * r[2] + p1 * r[5] + p2 * r[8];
          g[rr * 4 + 3] = p0 * t0 + p1 * t1 + p2 * t2 + p3;
        }
      }
    }
    __syncthreads();
  }

  if (live) {
    const float* g = sG[h][j];
    float* a = A + ((long long)b * NJ + j) * 12;
    const float j0 = sJ[h][j][0], j1 = sJ[h][j][1], j2 = sJ[h][j][2];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      a[rr * 4 + 0] = g[rr * 4 + 0]; a[rr * 4 + 1] = g[rr * 4 + 1]; a[rr * 4 + 2] = g[rr * 4 + 2];
      // rel_transform = G - pad(G [J;0]):  t - R J
      a[rr * 4 + 3] = g[rr * 4 + 3] - (g[rr * 4 + 0] * j0 + g[rr * 4 + 1] * j1 + g[rr * 4 + 2] * j2);
      joints16[((long long)b * NJ + j) * 3 + rr] = g[rr * 4 + 3];
    }
  }
}

__global__ void __launch_bounds__(256) mano_skin_kernel(hands_mano_consts c, const float* __restrict__ v_posed,
                                                        int ld_vp, const float* __restrict__ A,
                                                        const float* __restrict__ joints16,
                                                        const float* __restrict__ cam_wp,
                                                        const float* __restrict__ Kmat, float img_res,
                                                        float min_s, hands_mano_out o, int B) {
  __shared__ float sA[NJ * 12];
  __shared__ float sTip[5][3];
  __shared__ float sCam[3];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid < NJ * 12) sA[tid] = A[(long long)b * NJ * 12 + tid];
  const float* Kb = Kmat + (long long)b * 9;
  if (tid == 0) {
    // weak_perspective_to_perspective_torch (camera.py:456-474) with focal = (K00 + K11)/2
    const float f = (Kb[0] + Kb[4]) / 2.0f;
    const float s = fmaxf(cam_wp[b * 3 + 0], min_s);
    sCam[0] = cam_wp[b * 3 + 1];
    sCam[1] = cam_wp[b * 3 + 2];
    sCam[2] = 2.0f * f / (img_res * s + 1e-9f);
  }
  __syncthreads();
  const float cx = sCam[0], cy = sCam[1], cz = sCam[2];
  if (tid < 3) o.cam_t[b * 3 + tid] = sCam[tid];

  const float* vp = v_posed + (long long)b * ld_vp;
  for (int v = tid; v < NV; v += 256) {
    const float4* w4 = reinterpret_cast<const float4*>(c.lbs_weights + v * NJ);
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w = w4[q];
      const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* a = sA + (q * 4 + u) * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] += ww[u] * a[e];
      }
    }
    const float x = vp[3 * v + 0], y = vp[3 * v + 1], z = vp[3 * v + 2];
    const float ox = T[0] * x + T[1] * y + T[2] * z + T[3];
    const float oy = T[4] * x + T[5] * y + T[6] * z + T[7];
    const float oz = T[8] * x + T[9] * y + T[10] * z + T[11];
    float* dv = o.vertices + ((long long)b * NV + v) * 3;
    dv[0] = ox; dv[1] = oy; dv[2] = oz;
    float* dc = o.v3d_cam + ((long long)b * NV + v) * 3;
    dc[0] = ox + cx; dc[1] = oy + cy; dc[2] = oz + cz;
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (v == c.tip_ids[t]) { sTip[t][0] = ox; sTip[t][1] = oy; sTip[t][2] = oz; }
  }
  __syncthreads();
  if (tid < 21) {
    float p[3];
#pragma unroll
    for (int e = 0; e < 3; ++e)
      p[e] = tid < NJ ? joints16[((long long)b * NJ + tid) * 3 + e] : sTip[tid - NJ][e];
    float* dj = o.joints3d + ((long long)b * 21 + tid) * 3;
    dj[0] = p[0]; dj[1] = p[1]; dj[2] = p[2];
    const float px = p[0] + cx, py = p[1] + cy, pz = p[2] + cz;
    float* dc = o.j3d_cam + ((long long)b * 21 + tid) * 3;
    dc[0] = px; dc[1] = py; dc[2] = pz;
    // project2d_batch (transforms.py:316-329): K @ X then / z;  normalize_kp2d: 2x/res - 1
    const float hx = Kb[0] * px + Kb[1] * py + Kb[2] * pz;
    const float hy = Kb[3] * px + Kb[4] * py + Kb[5] * pz;
    const float hz = Kb[6] * px + Kb[7] * py + Kb[8] * pz;
    float* d2 = o.j2d_norm + ((long long)b * 21 + tid) * 2;
    d2[0] = 2.0f * (hx / hz) / img_res - 1.0f;
    d2[1] = 2.0f * (hy / hz) / img_res - 1.0f;
  }
}

}  // namespace

extern "C" {

int hands_mano_pose_f32(const hands_mano_consts* c, const float* rotmat, const float* betas, int ld_betas,
                        float* blend_in, int ld_blend, float* A, float* joints16, int B,
                        hands_stream_t stream) {
  if (!c || !c->pose_mean || !c->J_template || !c->J_shapedirs || !rotmat || !betas || !blend_in ||
      !A || !joints16 || B <= 0 || ld_blend < 145 || ld_betas < 10)
    return HANDS_EINVAL;
  hipLaunchKernelGGL(mano_pose_kernel<false>, dim3((B + HANDS_PER_BLOCK - 1) / HANDS_PER_BLOCK), dim3(64), 0,
                     (hipStream_t)stream, *c, rotmat, betas, ld_betas, blend_in, ld_blend, A, joints16, B);
  HANDS_LAUNCH_CHECK();
}

int hands_mano_pose_aa_f32(const hands_mano_consts* c, const float* axis_angle, const float* betas, int ld_betas,
                           float* blend_in, int ld_blend, float* A, float* joints16, int B, hands_stream_t stream) {
  if (!c || !c->pose_mean || !c->J_template || !c->J_shapedirs || !axis_angle || !betas || !blend_in || !A ||
      !joints16 || B <= 0 || ld_blend < 145 || ld_betas < 10)
    return HANDS_EINVAL;
  hipLaunchKernelGGL(mano_pose_kernel<true>, dim3((B + HANDS_PER_BLOCK - 1) / HANDS_PER_BLOCK), dim3(64), 0,
                     (hipStream_t)stream, *c, axis_angle, betas, ld_betas, blend_in, ld_blend, A, joints16, B);
  HANDS_LAUNCH_CHECK();
}

int hands_mano_skin_f32(const hands_mano_consts* c, const float* v_posed, int ld_vp, const float* A,
                        const float* joints16, const float* cam_wp, const float* K, float img_res,
                        float min_s, const hands_mano_out* out, int B, hands_stream_t stream) {
  if (!c || !c->lbs_weights || !c->tip_ids || !v_posed || !A || !joints16 || !cam_wp || !K || !out ||
      !out->vertices || !out->joints3d || !out->v3d_cam || !out->j3d_cam || !out->j2d_norm ||
      !out->cam_t || B <= 0 || ld_vp < NV * 3)
    return HANDS_EINVAL;
  hipLaunchKernelGGL(mano_skin_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *c, v_posed, ld_vp, A,
                     joints16, cam_wp, K, img_res, min_s, *out, B);
  HANDS_LAUNCH_CHECK();
}

int hands_abi_version(void) { return HANDS_ABI_VERSION; }

const char* hands_error_string(int code) {
  if (code == 0) return "ok";
  if (code == HANDS_EINVAL) return "hands: invalid argument / descriptor";
  return hipGetErrorString((hipError_t)code);
}

}  // extern "C"
