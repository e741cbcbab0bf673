// rot_device.h -- rotation conversions shared by the MANO and flip kernels (fp32, device).
// Each function states the reference site whose arithmetic it follows.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace hands {

// common/rot.py:118-177 matrix_to_quaternion: four candidates sqrt(max(0, 1 +- m00 +- m11 +- m22)),
// keep the one with the largest denominator (first maximum), divide by 2*max(q_abs, 0.1).
__device__ __forceinline__ void matrix_to_quaternion(const float* m, float* q) {
  const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6],
              m21 = m[7], m22 = m[8];
  float t[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22,
                1.0f - m00 - m11 + m22};
  float qa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) qa[i] = t[i] > 0.f ? sqrtf(t[i]) : 0.f;
  int idx = 0;
  float best = qa[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (qa[i] > best) { best = qa[i]; idx = i; }
  float c[4];
  if (idx == 0) { c[0] = qa[0] * qa[0]; c[1] = m21 - m12; c[2] = m02 - m20; c[3] = m10 - m01; }
  else if (idx == 1) { c[0] = m21 - m12; c[1] = qa[1] * qa[1]; c[2] = m10 + m01; c[3] = m02 + m20; }
  else if (idx == 2) { c[0] = m02 - m20; c[1] = m10 + m01; c[2] = qa[2] * qa[2]; c[3] = m12 + m21; }
  else { c[0] = m10 - m01; c[1] = m20 + m02; c[2] = m21 + m12; c[3] = qa[3] * qa[3]; }
  const float den = 2.0f * fmaxf(best, 0.1f);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = c[i] / den;
}

// common/rot.py:55-83 quaternion_to_axis_angle (and :754-782 for the reverse): sin(x/2)/x with the
// series 0.5 - x^2/48 when abs(x) < 1e-6.
__device__ __forceinline__ float sin_half_over_angle(float half, float ang) {
  return fabsf(ang) < 1e-6f ? 0.5f - (ang * ang) / 48.0f : sinf(half) / ang;
}

__device__ __forceinline__ void quaternion_to_axis_angle(const float* q, float* aa) {
  const float n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float half = atan2f(n, q[0]);
  const float ang = 2.0f * half;
  const float s = sin_half_over_angle(half, ang);
  aa[0] = q[1] / s; aa[1] = q[2] / s; aa[2] = q[3] / s;
}

__device__ __forceinline__ void matrix_to_axis_angle(const float* m, float* aa) {
  float q[4];
  matrix_to_quaternion(m, q);
  quaternion_to_axis_angle(q, aa);
}

// pytorch3d axis_angle_to_matrix = quaternion_to_matrix(axis_angle_to_quaternion(.)), both vendored
// at common/rot.py:754-782 and :86-115.  Used by the is_flipped branch only.
__device__ __forceinline__ void axis_angle_to_matrix(const float* aa, float* o) {
  const float ang = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
  const float half = ang * 0.5f;
  const float s = sin_half_over_angle(half, ang);
  const float r = cosf(half), i = aa[0] * s, j = aa[1] * s, k = aa[2] * s;
  const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
  o[0] = 1 - two_s * (j * j + k * k); o[1] = two_s * (i * j - k * r); o[2] = two_s * (i * k + j * r);
  o[3] = two_s * (i * j + k * r); o[4] = 1 - two_s * (i * i + k * k); o[5] = two_s * (j * k - i * r);
  o[6] = two_s * (i * k - j * r); o[7] = two_s * (j * k + i * r); o[8] = 1 - two_s * (i * i + j * j);
}

// smplx.lbs.batch_rodrigues: theta = ||r + 1e-8||, axis = r / theta, R = I + sin K + (1-cos) K K.
__device__ __forceinline__ void rodrigues(const float* r, float* R) {
  const float e0 = r[0] + 1e-8f, e1 = r[1] + 1e-8f, e2 = r[2] + 1e-8f;
  const float ang = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  const float rx = r[0] / ang, ry = r[1] / ang, rz = r[2] / ang;
  const float s = sinf(ang), c1 = 1.0f - cosf(ang);
  // K = [[0,-rz,ry],[rz,0,-rx],[-ry,rx,0]];  K@K computed entry-wise as bmm would
  const float k00 = -rz * rz - ry * ry, k01 = ry * rx, k02 = rz * rx;
  const float k10 = rx * ry, k11 = -rz * rz - rx * rx, k12 = rz * ry;
  const float k20 = rx * rz, k21 = ry * rz, k22 = -ry * ry - rx * rx;
  R[0] = 1.0f + c1 * k00;           R[1] = s * (-rz) + c1 * k01;  R[2] = s * ry + c1 * k02;
  R[3] = s * rz + c1 * k10;         R[4] = 1.0f + c1 * k11;       R[5] = s * (-rx) + c1 * k12;
  R[6] = s * (-ry) + c1 * k20;      R[7] = s * rx + c1 * k21;     R[8] = 1.0f + c1 * k22;
}

}  // namespace hands
