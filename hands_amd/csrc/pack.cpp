// pack.cpp -- HOST-side, one-time conversion of reference-layout parameters (PyTorch state_dict tensors,
// MANO asset arrays) into the layouts the gfx950 kernels read.  Plain C ABI, host pointers only, no GPU
// call: a non-Python host can produce every packed buffer from include/hands_hip.h alone.
//
// Replaces nothing on the reference's hot path (the reference applies conv and eval BatchNorm2d
// separately, src/nets/backbone/resnet.py:137-149); it is what makes the fused kernels drop-in for a
// reference checkpoint.  All folds run in fp64 and round once to fp32.
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "hands_hip.h"

namespace {
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
}  // namespace

extern "C" {

int hands_pack_conv_dims(int Cout, int Cin, int KH, int KW, int cin_pad_to, hands_packed_dims* dims) {
  if (!dims || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return HANDS_EINVAL;
  const int Cin_k = cin_pad_to > 0 ? cin_pad_to : Cin;
  if (Cin_k < Cin) return HANDS_EINVAL;
  dims->Cin = Cin_k;
  dims->Cout = round_up(Cout, 4);
  dims->Cout_pad = round_up(Cout, 128);
  dims->Kpad = round_up(KH * KW * Cin_k, 16);
  return 0;
}

// eval-mode BatchNorm2d folded into the preceding convolution:
//   scale = gamma / sqrt(var + eps);  w' = w * scale[cout];  b' = beta - mean * scale      (all fp64)
int hands_fold_bn_f32(int Cout, long long per_out, const float* w, const float* gamma, const float* beta,
                      const float* mean, const float* var, double eps, double* w_folded, double* bias_folded) {
  if (!w || !gamma || !beta || !mean || !var || !w_folded || !bias_folded || Cout <= 0 || per_out <= 0)
    return HANDS_EINVAL;
  for (int n = 0; n < Cout; ++n) {
    const double scale = (double)gamma[n] / sqrt((double)var[n] + eps);
    const float* src = w + (long long)n * per_out;
    double* dst = w_folded + (long long)n * per_out;
    for (long long i = 0; i < per_out; ++i) dst[i] = (double)src[i] * scale;
    const double ms = (double)mean[n] * scale;
    bias_folded[n] = (double)beta[n] - ms;
  }
  return 0;
}

// (Cout, Cin, KH, KW) -> [Cout_pad][Kpad], k = (kh, kw, cin) with cin padded to cin_pad_to; zero fill.
int hands_pack_conv_f64(int Cout, int Cin, int KH, int KW, int cin_pad_to, const double* w_oihw, const double* bias,
                        float* w_packed, float* bias_packed) {
  hands_packed_dims d;
  if (!w_oihw || !w_packed || !bias_packed || hands_pack_conv_dims(Cout, Cin, KH, KW, cin_pad_to, &d)) return HANDS_EINVAL;
  memset(w_packed, 0, sizeof(float) * (size_t)d.Cout_pad * d.Kpad);
  memset(bias_packed, 0, sizeof(float) * (size_t)d.Cout_pad);
  const long long hw = (long long)KH * KW;
  for (int n = 0; n < Cout; ++n) {
    float* row = w_packed + (size_t)n * d.Kpad;
    const double* src = w_oihw + (long long)n * Cin * hw;
    for (int c = 0; c < Cin; ++c)
      for (long long t = 0; t < hw; ++t) row[t * d.Cin + c] = (float)src[c * hw + t];
    if (bias) bias_packed[n] = (float)bias[n];
  }
  return 0;
}

// nn.Linear weight (N, K) as a 1x1 convolution.  col_index[k] (or k) is the packed column of reference
// input column k in a row of k_total columns; row_index[n] (or n) the packed row of reference output n
// among n_total stored outputs.  Kpad = round_up(k_total, 16), Cout_pad = round_up(n_total, 128).
int hands_pack_linear_f64(int N, int K, const double* w, const double* bias, const int32_t* col_index, int k_total,
                          const int32_t* row_index, int n_total, float* w_packed, float* bias_packed) {
  if (!w || !w_packed || !bias_packed || N <= 0 || K <= 0) return HANDS_EINVAL;
  const int kt = k_total > 0 ? k_total : K, nt = n_total > 0 ? n_total : N;
  if (kt < K || nt < N) return HANDS_EINVAL;
  const int Kpad = round_up(kt, 16), Cout_pad = round_up(nt, 128);
  for (int k = 0; k < K; ++k)
    if (col_index && (col_index[k] < 0 || col_index[k] >= kt)) return HANDS_EINVAL;
  for (int n = 0; n < N; ++n)
    if (row_index && (row_index[n] < 0 || row_index[n] >= nt)) return HANDS_EINVAL;
  memset(w_packed, 0, sizeof(float) * (size_t)Cout_pad * Kpad);
  memset(bias_packed, 0, sizeof(float) * (size_t)Cout_pad);
  for (int n = 0; n < N; ++n) {
    const int r = row_index ? row_index[n] : n;
    float* row = w_packed + (size_t)r * Kpad;
    const double* src = w + (long long)n * K;
    if (col_index) for (int k = 0; k < K; ++k) row[col_index[k]] = (float)src[k];
    else for (int k = 0; k < K; ++k) row[k] = (float)src[k];
    if (bias) bias_packed[r] = (float)bias[n];
  }
  return 0;
}

// conv3 + downsample of a bottleneck's first block as ONE two-source 1x1 GEMM (hands_conv1x1_dual_nhwc_f32):
// packed row [W0 (K0) | W1 (K1)], bias = b0 + b1 summed in fp64.  Inputs are the fp64 folds of both branches.
int hands_pack_conv1x1_dual_f64(int Cout, int K0, int K1, const double* w0, const double* b0, const double* w1,
                                const double* b1, float* w_packed, float* bias_packed) {
  if (!w0 || !w1 || !b0 || !b1 || !w_packed || !bias_packed || Cout <= 0 || K0 <= 0 || K1 <= 0) return HANDS_EINVAL;
  const int Kpad = round_up(K0 + K1, 16), Cout_pad = round_up(Cout, 128);
  memset(w_packed, 0, sizeof(float) * (size_t)Cout_pad * Kpad);
  memset(bias_packed, 0, sizeof(float) * (size_t)Cout_pad);
  for (int n = 0; n < Cout; ++n) {
    float* row = w_packed + (size_t)n * Kpad;
    for (int k = 0; k < K0; ++k) row[k] = (float)w0[(long long)n * K0 + k];
    for (int k = 0; k < K1; ++k) row[K0 + k] = (float)w1[(long long)n * K1 + k];
    bias_packed[n] = (float)(b0[n] + b1[n]);
  }
  return 0;
}

// MANO constants for hands_mano_pose_f32 / the blend GEMM / hands_mano_skin_f32 from the asset arrays
// smplx registers as buffers (common/body_models.py:92-99): v_template (778,3), shapedirs (778,3,10),
// posedirs (135, 2334), J_regressor (16,778) dense, hands_mean (45).
//   pose_mean (48) = zeros(3) ++ hands_mean;  J_template (16,3) = J_regressor @ v_template (fp64);
//   J_shapedirs (48,10) = J_regressor @ shapedirs (fp64);
//   blend weight [2432][160]: row m = 3*vertex + coord, columns [shapedirs[m, 0:10] | posedirs[0:135, m] | 0],
//   blend bias [2432] = v_template.flatten().
int hands_pack_mano_f32(const float* v_template, const float* shapedirs, const float* posedirs,
                        const float* J_regressor, const float* hands_mean, float* pose_mean, float* J_template,
                        float* J_shapedirs, float* blend_w_packed, float* blend_bias_packed) {
  if (!v_template || !shapedirs || !posedirs || !J_regressor || !hands_mean || !pose_mean || !J_template ||
      !J_shapedirs || !blend_w_packed || !blend_bias_packed)
    return HANDS_EINVAL;
  constexpr int NV = 778, NJ = 16, NB = 10, NP = 135, M = NV * 3, KP = 160, NPAD = 2432;
  pose_mean[0] = pose_mean[1] = pose_mean[2] = 0.f;
  for (int i = 0; i < 45; ++i) pose_mean[3 + i] = hands_mean[i];
  for (int j = 0; j < NJ; ++j) {
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int v = 0; v < NV; ++v) acc += (double)J_regressor[j * NV + v] * (double)v_template[v * 3 + c];
      J_template[j * 3 + c] = (float)acc;
      for (int k = 0; k < NB; ++k) {
        double a2 = 0.0;
        for (int v = 0; v < NV; ++v) a2 += (double)J_regressor[j * NV + v] * (double)shapedirs[(v * 3 + c) * NB + k];
        J_shapedirs[(j * 3 + c) * NB + k] = (float)a2;
      }
    }
  }
  memset(blend_w_packed, 0, sizeof(float) * (size_t)NPAD * KP);
  memset(blend_bias_packed, 0, sizeof(float) * (size_t)NPAD);
  for (int m = 0; m < M; ++m) {
    float* row = blend_w_packed + (size_t)m * KP;
    for (int k = 0; k < NB; ++k) row[k] = shapedirs[m * NB + k];
    for (int p = 0; p < NP; ++p) row[NB + p] = posedirs[(size_t)p * M + m];
    blend_bias_packed[m] = v_template[m];
    row[NB + NP] = v_template[m];      // column 145: see include/hands_hip.h
  }
  return 0;
}

// Winograd F(2x2, 3x3) weights for hands_conv3x3_winograd_f32: U = G g G^T per (cout, cin) in fp64, one rounding to fp32,
// stored in MFMA-A operand order [Cout/32][Cin/8][xi 4][nu 4][lane 64][4]: lane l of frequency (xi, nu) holds output
// channel 32 nb + (l & 31), input channels 8 c8 + 4 (l >> 5) + 0..3.  w_oihw (Cout, Cin, 3, 3) is the BN-folded weight.
//   G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]]
long long hands_pack_conv3x3_winograd_floats(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || Cout % 32 || Cin % 16) return 0;
  return 16LL * Cout * Cin;
}

int hands_pack_conv3x3_winograd_f64(int Cout, int Cin, const double* w_oihw, float* u_packed) {
  if (!w_oihw || !u_packed || Cout <= 0 || Cin <= 0 || Cout % 32 || Cin % 16) return HANDS_EINVAL;
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  const int nc8 = Cin / 8;
  for (int o = 0; o < Cout; ++o) {
    const int nb = o / 32, ol = o % 32;
    for (int c = 0; c < Cin; ++c) {
      const double* g = w_oihw + ((long long)o * Cin + c) * 9;
      double t[4][3];
      for (int x = 0; x < 4; ++x)
        for (int j = 0; j < 3; ++j) t[x][j] = G[x][0] * g[0 * 3 + j] + G[x][1] * g[1 * 3 + j] + G[x][2] * g[2 * 3 + j];
      const int c8 = c / 8, h = (c % 8) / 4, e = c % 4;
      for (int x = 0; x < 4; ++x)
        for (int n = 0; n < 4; ++n) {
          const double u = t[x][0] * G[n][0] + t[x][1] * G[n][1] + t[x][2] * G[n][2];
          const long long idx = (((((long long)nb * nc8 + c8) * 4 + x) * 4 + n) * 64 + (h * 32 + ol)) * 4 + e;
          u_packed[idx] = (float)u;
        }
    }
  }
  return 0;
}

// Winograd F(4x4, 3x3) weights for hands_conv3x3_winograd4_f32: U = G g G^T (6x6 per (cout, cin)) in fp64, one rounding to
// fp32, MFMA-A operand order [Cout/32][Cin/8][f = 6 xi + nu][lane 64][4]: lane l of frequency f holds output channel
// 32 nb + (l & 31), input channels 8 c8 + 4 (l >> 5) + 0..3.
//   G = [[1/4, 0, 0], [-1/6, -1/6, -1/6], [-1/6, 1/6, -1/6], [1/24, 1/12, 1/6], [1/24, -1/12, 1/6], [0, 0, 1]]
long long hands_pack_conv3x3_winograd4_floats(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || Cout % 32 || Cin % 8) return 0;
  return 36LL * Cout * Cin;
}

int hands_pack_conv3x3_winograd4_f64(int Cout, int Cin, const double* w_oihw, float* u_packed) {
  if (!w_oihw || !u_packed || Cout <= 0 || Cin <= 0 || Cout % 32 || Cin % 8) return HANDS_EINVAL;
  static const double G[6][3] = {{1.0 / 4, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  const int nc8 = Cin / 8;
  for (int o = 0; o < Cout; ++o) {
    const int nb = o / 32, ol = o % 32;
    for (int c = 0; c < Cin; ++c) {
      const double* g = w_oihw + ((long long)o * Cin + c) * 9;
      double t[6][3];
      for (int x = 0; x < 6; ++x)
        for (int j = 0; j < 3; ++j) t[x][j] = G[x][0] * g[0 * 3 + j] + G[x][1] * g[1 * 3 + j] + G[x][2] * g[2 * 3 + j];
      const int c8 = c / 8, h = (c % 8) / 4, e = c % 4;
      for (int x = 0; x < 6; ++x)
        for (int n = 0; n < 6; ++n) {
          const double u = t[x][0] * G[n][0] + t[x][1] * G[n][1] + t[x][2] * G[n][2];
          const long long idx = ((((long long)nb * nc8 + c8) * 36 + x * 6 + n) * 64 + (h * 32 + ol)) * 4 + e;
          u_packed[idx] = (float)u;
        }
    }
  }
  return 0;
}

}  // extern "C"
