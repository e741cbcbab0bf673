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


// ---------------------------------------------------------------------------------------------------------
// mano_heads_kernel: MANOHead.forward for BOTH hands in ONE launch (BASELINE configs[4]: the 6 launches of
// pose -> blend GEMM -> skin per side become 1).  A block owns 16 hands of one side and a range of 64-vertex
// chunks:
//   phase 1  one thread per (hand, joint): R -> axis-angle -> (+pose_mean) -> Rodrigues, J(beta), forward
//            kinematics, skinning transforms A -- exactly mano_pose_kernel's arithmetic -- into LDS, with the
//            blend-GEMM input rows [beta | R - I | 0] (16 x 160);
//   phase 2  per chunk: v_posed[192 x 16 hands] = Wb[192 x 160] . blend_in^T + v_template on the fp32 matrix
//            cores (v_mfma_f32_16x16x4_f32: the 16 hands are the N dimension; weight fragments straight from
//            L2 in operand order, the 16 x 160 activation operand lives in registers), staged through LDS;
//   phase 3  skinning transforms T[v] = sum_u w[v][u] A_u as a second small product on the matrix cores (per hand:
//            12 transform entries x 16 joints x 16 vertices), applied to the posed vertices, camera translation;
//            fingertip joints + projection at the end of the block that skinned the tip vertex.
// Every output element is computed by one fixed instruction sequence, independent of the grid shape.
struct ManoSideArgs {
  hands_mano_consts c;
  const float* blend_w;      // [2432][160] packed (hands_pack_mano_f32)
  const float* blend_bias;   // [2432] = v_template
  hands_mano_out o;
  const float* rot;          // this side's rows: (B,16,3,3) or (B,48) axis-angle
  const float* betas;        // (B, ld_betas)
  const float* cam_wp;       // (B,3)
};
struct ManoHeadsArgs {
  ManoSideArgs side[2];
  const float* K;            // (B,3,3) shared by both sides
  int ld_betas, B, vsplit;
  float img_res, min_s;
};

constexpr int MH = 16;             // hands per block = MFMA N tiles of 16 (32 measured no faster: the kernel is MFMA-bound)
constexpr int CHUNK_V = 64;        // vertices per chunk
constexpr int CHUNK_M = 192;       // blend outputs per chunk = 12 row tiles of 16
constexpr int NCHUNK = (NV + CHUNK_V - 1) / CHUNK_V;   // 13
constexpr int VROW = 196;          // LDS row of the v_posed stage (49 x 16 B: odd)
constexpr int BROW = 164;          // LDS row of the blend input (41 x 16 B: odd)
// geometry the blend loop's two shortcuts rely on (mano_heads_kernel): the last chunk's 3 * (778 - 12 * 64) = 30 outputs fit row
// tiles 0-1 (tile tt > 0 of a wave is skipped there), and the contraction is 10 shape + 135 pose + 1 (v_template) = 146 of the
// packed row's 160 columns, so the last k-step of four holds k = 144, 145 in .x .y and zero padding in .z .w
constexpr int BLEND_K = 10 + 135 + 1, BLEND_KPAD = 160;
static_assert(3 * (NV - (NCHUNK - 1) * CHUNK_V) <= 32, "last chunk must fit the first two 16-row tiles");
static_assert(BLEND_K == 146 && BLEND_KPAD == 160 && BLEND_KPAD - BLEND_K >= 14 && (BLEND_K - 1) / 16 == 9 && (BLEND_K - 1) % 16 < 4 * 4,
              "k = 145 must be the last non-zero column, in the last of ten 16-wide steps");
static_assert(CHUNK_M == 3 * CHUNK_V && CHUNK_M == 12 * 16, "a chunk is 12 row tiles of 16 outputs");

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool AA_INPUT>
__global__ void __launch_bounds__(256, 2) mano_heads_kernel(ManoHeadsArgs a) {
  // phase-1 scratch (joint rotations, rest joints, global transforms) is dead once A is built: it shares its
  // bytes with the v_posed stage of phases 2-3
  __shared__ __attribute__((aligned(16))) float sScratch[MH * NJ * 15 > MH * VROW ? MH * NJ * 15 : MH * VROW];
  float (*sJ)[NJ][3] = reinterpret_cast<float (*)[NJ][3]>(sScratch);
  float (*sG)[NJ][12] = reinterpret_cast<float (*)[NJ][12]>(sScratch + MH * NJ * 3);
  float (*stage)[VROW] = reinterpret_cast<float (*)[VROW]>(sScratch);
  __shared__ __attribute__((aligned(16))) float sAT[MH][4][12][4];
  __shared__ __attribute__((aligned(16))) float sBin[MH][BROW];
  __shared__ float sCam[MH][3];
  __shared__ float sTip[MH][5][3];
  const ManoSideArgs& S = a.side[blockIdx.y];
  const hands_mano_consts& c = S.c;
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * MH;
  const int B = a.B;

  // ---- phase 1: pose, joints, forward kinematics: thread = (hand h or h + 16, joint j) ----------------------
#if defined(HANDS_MANO_ABL) && HANDS_MANO_ABL == 1      // timing-only ablation (tools/build_variant.sh): no pose / FK
  for (int e = tid; e < MH * BROW; e += 256) (&sBin[0][0])[e] = 0.f;
  for (int e = tid; e < MH * NJ * 12; e += 256) (&sAT[0][0][0][0])[e] = 0.f;
  if (tid < MH * 3) (&sCam[0][0])[tid] = 0.f;
  __syncthreads();
#else
  {
    const int j = tid & 15;
    float Rloc[MH / 16][9];                          // this thread's local joint rotations (one per hand it owns)
#pragma unroll
    for (int rep = 0; rep < MH / 16; ++rep) {
      const int h = (tid >> 4) + 16 * rep;
      const int b = b0 + h;
      if (b < B) {
        float aa[3];
        float* R = Rloc[rep];
        if constexpr (AA_INPUT) {
          const float* src = S.rot + ((long long)b * NJ + j) * 3;
          aa[0] = src[0]; aa[1] = src[1]; aa[2] = src[2];
        } else {
          hands::matrix_to_axis_angle(S.rot + ((long long)b * NJ + j) * 9, aa);
        }
        aa[0] += c.pose_mean[3 * j + 0];
        aa[1] += c.pose_mean[3 * j + 1];
        aa[2] += c.pose_mean[3 * j + 2];
        hands::rodrigues(aa, R);
        const float* be = S.betas + (long long)b * a.ld_betas;
        float* row = sBin[h];
        if (j >= 1) {
#pragma unroll
          for (int e = 0; e < 9; ++e) row[10 + (j - 1) * 9 + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        }
        if (j < 10) row[j] = be[j];
        if (j < 15) row[145 + j] = j == 0 ? 1.f : 0.f;    // column 145: the constant that multiplies v_template (below)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          float acc = c.J_template[3 * j + cc];
          const float* js = c.J_shapedirs + (3 * j + cc) * 10;
#pragma unroll
          for (int k = 0; k < 10; ++k) acc += js[k] * be[k];
          sJ[h][j][cc] = acc;
        }
        if (j == 0) {
          // weak_perspective_to_perspective_torch (camera.py:456-474) with focal = (K00 + K11)/2
          const float* Kb = a.K + (long long)b * 9;
          const float f = (Kb[0] + Kb[4]) / 2.0f;
          const float s = fmaxf(S.cam_wp[b * 3 + 0], a.min_s);
          sCam[h][0] = S.cam_wp[b * 3 + 1];
          sCam[h][1] = S.cam_wp[b * 3 + 2];
          sCam[h][2] = 2.0f * f / (a.img_res * s + 1e-9f);
        }
      } else {
        for (int e = j; e < 160; e += NJ) sBin[h][e] = 0.f;   // dead rows feed zeros to the matrix cores
      }
    }
    __syncthreads();
    const int dj = depth_of(j), pj = parent_of(j);
    for (int level = 0; level < 4; ++level) {
#pragma unroll
      for (int rep = 0; rep < MH / 16; ++rep) {
        const int h = (tid >> 4) + 16 * rep;
        if (b0 + h < B && dj == level) {
          float* g = sG[h][j];
          const float* r = Rloc[rep];
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
      }
      __syncthreads();
    }
#pragma unroll
    for (int rep = 0; rep < MH / 16; ++rep) {
      const int h = (tid >> 4) + 16 * rep;
      const int b = b0 + h;
      if (b < B) {
        const float* g = sG[h][j];
        float am[12];
        const float j0 = sJ[h][j][0], j1 = sJ[h][j][1], j2 = sJ[h][j][2];
        const float cx = sCam[h][0], cy = sCam[h][1], cz = sCam[h][2];
        float p[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          am[rr * 4 + 0] = g[rr * 4 + 0]; am[rr * 4 + 1] = g[rr * 4 + 1]; am[rr * 4 + 2] = g[rr * 4 + 2];
          am[rr * 4 + 3] = g[rr * 4 + 3] - (g[rr * 4 + 0] * j0 + g[rr * 4 + 1] * j1 + g[rr * 4 + 2] * j2);
          p[rr] = g[rr * 4 + 3];
        }
        {
          float (*at)[12][4] = sAT[h];
#pragma unroll
          for (int e = 0; e < 12; ++e) at[j & 3][e][j >> 2] = am[e];
        }
        if (blockIdx.z == 0) {       // the 16 posed joints + camera outputs: written by the first vertex-range block
          float* dj3 = S.o.joints3d + ((long long)b * 21 + j) * 3;
          dj3[0] = p[0]; dj3[1] = p[1]; dj3[2] = p[2];
          const float px = p[0] + cx, py = p[1] + cy, pz = p[2] + cz;
          float* dc = S.o.j3d_cam + ((long long)b * 21 + j) * 3;
          dc[0] = px; dc[1] = py; dc[2] = pz;
          const float* Kb = a.K + (long long)b * 9;
          const float hx = Kb[0] * px + Kb[1] * py + Kb[2] * pz;
          const float hy = Kb[3] * px + Kb[4] * py + Kb[5] * pz;
          const float hz = Kb[6] * px + Kb[7] * py + Kb[8] * pz;
          float* d2 = S.o.j2d_norm + ((long long)b * 21 + j) * 2;
          d2[0] = 2.0f * (hx / hz) / a.img_res - 1.0f;
          d2[1] = 2.0f * (hy / hz) / a.img_res - 1.0f;
          if (j < 3) S.o.cam_t[b * 3 + j] = sCam[h][j];
        }
      }
    }
    __syncthreads();      // sA complete; the scratch (sR, sJ, sG) is free for the v_posed stage from here on
  }
#endif

  // ---- phase 2 + 3 per chunk --------------------------------------------------------------------------------
  const int lane = tid & 63, wave = tid >> 6;
  const int mh = lane & 15, g = lane >> 4;           // MFMA column (hand within a 16-hand group) and k-group
  float4 bfrag[MH / 16][10];                         // this lane's blend-input operands: k = 16*s + 4*g + e
#pragma unroll
  for (int nt = 0; nt < MH / 16; ++nt)
#pragma unroll
    for (int s4 = 0; s4 < 10; ++s4) bfrag[nt][s4] = *reinterpret_cast<const float4*>(&sBin[nt * 16 + mh][16 * s4 + 4 * g]);

  const int c0 = (int)((long long)blockIdx.z * NCHUNK / a.vsplit);
  const int c1 = (int)((long long)(blockIdx.z + 1) * NCHUNK / a.vsplit);
  // skin assignment: wave w owns the chunk's vertex tile w (16 vertices, MFMA column = lane & 15); lane group g
  // (= lane >> 4) ends up with row g of each vertex's 3x4 skinning transform, i.e. computes output coordinate g
  int tip[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) tip[t] = c.tip_ids[t];
  const int e12 = lane & 15;                         // MFMA row of the transform GEMM: entry e of A_u (12 used)

  for (int ch = c0; ch < c1; ++ch) {
#if defined(HANDS_MANO_ABL) && HANDS_MANO_ABL == 3      // timing-only ablation: no blend product
    for (int e = tid; e < MH * VROW; e += 256) (&stage[0][0])[e] = 1.f;
    __syncthreads();
#else
    // -- blend GEMM of this chunk: wave w owns row tiles w, w+4, w+8.  A weight fragment (16 rows x 4 k per
    //    instruction, straight from L2 in operand order) feeds one MFMA per 16-hand group, so the 1.5 MB blend
    //    matrix is streamed once per MH hands; the groups' accumulation chains are independent and interleaved
    float4 wf[10];
    {
      int m0 = ch * CHUNK_M + wave * 16;
      m0 = m0 < 2432 ? m0 : 2432 - 16;                 // past the packed matrix: recompute its (all-zero) last tile
      const float* wr = S.blend_w + (size_t)(m0 + mh) * 160 + 4 * g;
#pragma unroll
      for (int s4 = 0; s4 < 10; ++s4) wf[s4] = *reinterpret_cast<const float4*>(wr + 16 * s4);
    }
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      if (tt > 0 && ch == NCHUNK - 1) break;           // chunk 12 holds vertices in its first two row tiles only (2304 ... 2335)
      int m0 = ch * CHUNK_M + (wave + 4 * tt) * 16;
      m0 = m0 < 2432 ? m0 : 2432 - 16;
      // v_template is column 145 of the packed matrix (hands_pack_mano_f32) against the constant 1 of the input row: the
      // accumulators start at zero and no tile waits for a bias load before its first MFMA
      f32x4 acc[MH / 16];
#pragma unroll
      for (int nt = 0; nt < MH / 16; ++nt) { acc[nt][0] = 0.f; acc[nt][1] = 0.f; acc[nt][2] = 0.f; acc[nt][3] = 0.f; }
      float4 wn[10];                                   // next tile's fragments, requested under this tile's MFMAs
      if (tt < 2) {
        int m1 = ch * CHUNK_M + (wave + 4 * (tt + 1)) * 16;
        m1 = m1 < 2432 ? m1 : 2432 - 16;
        const float* wr = S.blend_w + (size_t)(m1 + mh) * 160 + 4 * g;
#pragma unroll
        for (int s4 = 0; s4 < 10; ++s4) wn[s4] = *reinterpret_cast<const float4*>(wr + 16 * s4);
      }
#pragma unroll
      for (int s4 = 0; s4 < 10; ++s4) {
#pragma unroll
        for (int nt = 0; nt < MH / 16; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s4].x, bfrag[nt][s4].x, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < MH / 16; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s4].y, bfrag[nt][s4].y, acc[nt], 0, 0, 0);
        if (s4 == 9) continue;       // .y of the last step is k = 145 (1 x v_template), .z .w are k = 146 ... 159: zero padding
#pragma unroll
        for (int nt = 0; nt < MH / 16; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s4].z, bfrag[nt][s4].z, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < MH / 16; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s4].w, bfrag[nt][s4].w, acc[nt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < MH / 16; ++nt)
        *reinterpret_cast<float4*>(&stage[nt * 16 + mh][(wave + 4 * tt) * 16 + 4 * g]) =
            make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
      if (tt < 2) {
#pragma unroll
        for (int s4 = 0; s4 < 10; ++s4) wf[s4] = wn[s4];
      }
    }
    __syncthreads();
#endif
#if !(defined(HANDS_MANO_ABL) && HANDS_MANO_ABL == 2)   // timing-only ablation 2: no skinning product
    // -- skinning: T[v][h] = sum_u w[v][u] A_h[u] as a (12 x 16 joints) x (16 joints x 16 vertices) product per hand
    //    on the matrix cores; lane (vertex, g < 3) then holds row g of T and applies it to the posed vertex
    const int vl = wave * 16 + (lane & 15);
    const int v = ch * CHUNK_V + vl;
    const bool v_ok = v < NV;
    float wreg[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) wreg[s4] = v_ok ? c.lbs_weights[v * NJ + 4 * s4 + g] : 0.f;
    int tip_slot = -1;
#pragma unroll
    for (int t = 0; t < 5; ++t) tip_slot = (v == tip[t]) ? t : tip_slot;
    for (int h0 = 0; h0 < MH; h0 += 4) {
      f32x4 T[4];
      float av[4][4];
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        const float4 a4 = e12 < 12 ? *reinterpret_cast<const float4*>(sAT[h0 + hh][g][e12]) : make_float4(0.f, 0.f, 0.f, 0.f);
        av[hh][0] = a4.x; av[hh][1] = a4.y; av[hh][2] = a4.z; av[hh][3] = a4.w;
        T[hh][0] = 0.f; T[hh][1] = 0.f; T[hh][2] = 0.f; T[hh][3] = 0.f;
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) T[hh] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[hh][s4], wreg[s4], T[hh], 0, 0, 0);
      }
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        const int h = h0 + hh;
        const int b = b0 + h;
        if (v_ok && g < 3 && b < B) {
          const float x = stage[h][3 * vl + 0], y = stage[h][3 * vl + 1], z = stage[h][3 * vl + 2];
          const float o = T[hh][0] * x + T[hh][1] * y + T[hh][2] * z + T[hh][3];
          stage[h][3 * vl + g] = o;                  // in place: every lane of the wave has read x, y, z by now
          if (tip_slot >= 0) sTip[h][tip_slot][g] = o;
        }
      }
    }
#else
    const int vl = wave * 16 + (lane & 15);
#endif
#if !(defined(HANDS_MANO_ABL) && HANDS_MANO_ABL == 4)   // timing-only ablation 4: no vertex write-back
    // the wave's 16 vertices x 3 coordinates of every hand are 48 consecutive floats both in the stage and in
    // the output arrays: write them back 8 bytes per lane (hand rows are 9336 B apart: 8-byte aligned)
#pragma unroll
    for (int it = 0; it < MH * 24 / 64; ++it) {
      const int idx = it * 64 + lane;
      const int h = idx / 24, k2 = (idx - 24 * h) * 2;
      const int m = ch * CHUNK_M + wave * 48 + k2;   // first of the two output floats (vertex-major, 3 per vertex)
      const int b = b0 + h;
      if (b < B && m < NV * 3) {
        const float2 o2 = *reinterpret_cast<const float2*>(&stage[h][wave * 48 + k2]);
        const int c0i = m % 3, c1i = (m + 1) % 3;
        *reinterpret_cast<float2*>(S.o.vertices + (long long)b * (NV * 3) + m) = o2;
        *reinterpret_cast<float2*>(S.o.v3d_cam + (long long)b * (NV * 3) + m) =
            make_float2(o2.x + sCam[h][c0i], o2.y + sCam[h][c1i]);
      }
    }
#endif
    __syncthreads();      // the stage is rewritten by the next chunk's GEMM
  }
  // ---- fingertip joints 16..20 of the tips whose vertex this block skinned: joints3d, camera space, projection
  if (tid < MH * 5) {
    const int h = tid / 5, t = tid - 5 * h;
    const int b = b0 + h;
    const int tch = tip[t] / CHUNK_V;
    if (b < B && tch >= c0 && tch < c1) {
      const float ox = sTip[h][t][0], oy = sTip[h][t][1], oz = sTip[h][t][2];
      const float cx = sCam[h][0], cy = sCam[h][1], cz = sCam[h][2];
      float* dj3 = S.o.joints3d + ((long long)b * 21 + 16 + t) * 3;
      dj3[0] = ox; dj3[1] = oy; dj3[2] = oz;
      const float px = ox + cx, py = oy + cy, pz = oz + cz;
      float* dcj = S.o.j3d_cam + ((long long)b * 21 + 16 + t) * 3;
      dcj[0] = px; dcj[1] = py; dcj[2] = pz;
      const float* Kb = a.K + (long long)b * 9;
      const float hx = Kb[0] * px + Kb[1] * py + Kb[2] * pz;
      const float hy = Kb[3] * px + Kb[4] * py + Kb[5] * pz;
      const float hz = Kb[6] * px + Kb[7] * py + Kb[8] * pz;
      float* d2 = S.o.j2d_norm + ((long long)b * 21 + 16 + t) * 2;
      d2[0] = 2.0f * (hx / hz) / a.img_res - 1.0f;
      d2[1] = 2.0f * (hy / hz) / a.img_res - 1.0f;
    }
  }
}

}  // namespace

extern "C" {

int hands_mano_heads_f32(const hands_mano_side* sides, int n_sides, const float* K, int ld_betas, float img_res,
                         float min_s, int B, int axis_angle_input, hands_stream_t stream) {
  if (!sides || n_sides < 1 || n_sides > 2 || !K || B <= 0 || ld_betas < 10) return HANDS_EINVAL;
  ManoHeadsArgs a;
  for (int s = 0; s < n_sides; ++s) {
    const hands_mano_side& h = sides[s];
    if (!h.consts.pose_mean || !h.consts.J_template || !h.consts.J_shapedirs || !h.consts.lbs_weights ||
        !h.consts.tip_ids || !h.blend_w || !h.blend_bias || !h.rot || !h.betas || !h.cam_wp || !h.out.vertices ||
        !h.out.joints3d || !h.out.v3d_cam || !h.out.j3d_cam || !h.out.j2d_norm || !h.out.cam_t)
      return HANDS_EINVAL;
    a.side[s].c = h.consts; a.side[s].blend_w = h.blend_w; a.side[s].blend_bias = h.blend_bias; a.side[s].o = h.out;
    a.side[s].rot = h.rot; a.side[s].betas = h.betas; a.side[s].cam_wp = h.cam_wp;
  }
  if (n_sides == 1) a.side[1] = a.side[0];
  a.K = K; a.ld_betas = ld_betas; a.B = B; a.img_res = img_res; a.min_s = min_s;
  const int nb = (B + MH - 1) / MH;
  // split the 13 vertex chunks over blockIdx.z until ~800 blocks (3 per CU) exist; only the splits that lower the
  // largest per-block chunk count are candidates (results do not depend on the split).  Measured at 2048 hands:
  // 4 -> 55 us, 7 -> 49 us, 13 -> 55 us; at 512 hands 13 is best (22 us)
  const int base = nb * n_sides;
  const int want = (800 + base - 1) / base;
  int vs = NCHUNK;
  const int cand[7] = {1, 2, 3, 4, 5, 7, 13};
  for (int i = 6; i >= 0; --i)
    if (cand[i] >= want) vs = cand[i];
  a.vsplit = vs;
  dim3 grid((unsigned)nb, (unsigned)n_sides, (unsigned)a.vsplit);
  if (axis_angle_input) hipLaunchKernelGGL(mano_heads_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(mano_heads_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  HANDS_LAUNCH_CHECK();
}

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

int hands_stream_is_capturing(hands_stream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing((hipStream_t)stream, &st);
  if (e != hipSuccess) return -(int)e;
  return st == hipStreamCaptureStatusNone ? 0 : 1;
}

const char* hands_error_string(int code) {
  if (code == 0) return "ok";
  if (code == HANDS_EINVAL) return "hands: invalid argument / descriptor";
  return hipGetErrorString((hipError_t)code);
}

}  // extern "C"
