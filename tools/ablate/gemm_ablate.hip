// Ablation harness for the conv_igemm main loop (dev tool, not part of the library).
// Plain GEMM D[n][m] = sum_k W[n][k] X[m][k] with the same tiling; VARIANT selects what is removed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float f4e(const float4& v, int t) { return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }

// VARIANT 0: full; 1: no global loads/ds_write in loop (LDS content reused); 2: no LDS reads (register operands);
// 3: full but only 2 blocks/CU (extra LDS); 4: MFMA only (no loads, no LDS, no barrier)
template <int VARIANT>
__global__ void __launch_bounds__(256, 2) gemm_kernel(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ D,
                                                      int M, int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 16, LR = 20;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LR + (VARIANT == 3 ? 12000 : 0)];
  float* sX = lds; float* sW = lds + 2 * BM * LR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int nbn = N / BN;
  const int m0 = (blockIdx.x / nbn) * BM, n0 = (blockIdx.x % nbn) * BN;
  const int srow = tid >> 2, chunk = tid & 3;
  const float* xp0 = X + (size_t)(m0 + srow) * K + chunk * 4; const float* xp1 = xp0 + (size_t)64 * K;
  const float* wp0 = W + (size_t)(n0 + srow) * K + chunk * 4; const float* wp1 = wp0 + (size_t)64 * K;
  float4 xr0, xr1, wr0, wr1;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
#define LOADT(kt) { xr0 = *(const float4*)(xp0 + (kt) * BK); xr1 = *(const float4*)(xp1 + (kt) * BK); wr0 = *(const float4*)(wp0 + (kt) * BK); wr1 = *(const float4*)(wp1 + (kt) * BK); }
#define STORET(buf) { float* dx = sX + (buf) * BM * LR + srow * LR + chunk * 4; float* dw = sW + (buf) * BN * LR + srow * LR + chunk * 4; \
    *(float4*)dx = xr0; *(float4*)(dx + 64 * LR) = xr1; *(float4*)dw = wr0; *(float4*)(dw + 64 * LR) = wr1; }
  LOADT(0); STORET(0); __syncthreads();
  const int frag = (lane & 31) * LR + (lane >> 5) * 4;
  const float* fw = sW + wn * 64 * LR + frag; const float* fx = sX + wm * 64 * LR + frag;
  float4 wf[2][2], xf[2][2];
  if (VARIANT == 2 || VARIANT == 4) for (int kk = 0; kk < 2; ++kk) for (int i = 0; i < 2; ++i) { wf[i][kk] = *(const float4*)(fw + i * 32 * LR + kk * 8); xf[i][kk] = *(const float4*)(fx + i * 32 * LR + kk * 8); }
  if (VARIANT == 6) {
    // direct global->LDS (global_load_lds_dwordx4): unpadded 64-B rows, 16-B chunk XOR-swizzled by (row>>2)&3
    // on the SOURCE side (lane picks which k-chunk it fetches) and on the fragment reads.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    float* sX6 = lds; float* sW6 = lds + 2 * BM * 16;
    const int kc = (tid & 3) ^ ((tid >> 4) & 3);
    const float* gx0 = X + (size_t)(m0 + srow) * K + kc * 4; const float* gx1 = gx0 + (size_t)64 * K;
    const float* gw0 = W + (size_t)(n0 + srow) * K + kc * 4; const float* gw1 = gw0 + (size_t)64 * K;
#define GLDS(kt, buf) { \
    __builtin_amdgcn_global_load_lds((gbl_void*)(gx0 + (kt) * BK), (lds_void*)(sX6 + (buf) * BM * 16 + (wave * 16) * 16), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void*)(gx1 + (kt) * BK), (lds_void*)(sX6 + (buf) * BM * 16 + (64 + wave * 16) * 16), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void*)(gw0 + (kt) * BK), (lds_void*)(sW6 + (buf) * BN * 16 + (wave * 16) * 16), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void*)(gw1 + (kt) * BK), (lds_void*)(sW6 + (buf) * BN * 16 + (64 + wave * 16) * 16), 16, 0, 0); }
    __syncthreads();   // the register-staged prologue above wrote the padded layout; start clean
    GLDS(0, 0);
    __syncthreads();
    const int sw = (lane >> 2) & 3, h = lane >> 5;
    const int o0 = (lane & 31) * 16 + ((h ^ sw) << 2), o1 = (lane & 31) * 16 + (((2 + h) ^ sw) << 2);
    const float* fw6 = sW6 + wn * 64 * 16; const float* fx6 = sX6 + wm * 64 * 16;
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) GLDS(kt + 1, buf ^ 1);
      for (int i = 0; i < 2; ++i) {
        wf[i][0] = *(const float4*)(fw6 + buf * BN * 16 + i * 32 * 16 + o0); wf[i][1] = *(const float4*)(fw6 + buf * BN * 16 + i * 32 * 16 + o1);
        xf[i][0] = *(const float4*)(fx6 + buf * BM * 16 + i * 32 * 16 + o0); xf[i][1] = *(const float4*)(fx6 + buf * BM * 16 + i * 32 * 16 + o1); }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf[i][kk], t), f4e(xf[j][kk], t), acc[i][j], 0, 0, 0);
      __syncthreads();
    }
  } else
  if (VARIANT == 7) {
    // direct global->LDS (global_load_lds_dwordx4): unpadded 64-B rows, 16-B chunk XOR-swizzled by (row>>2)&3
    // on the SOURCE side (lane picks which k-chunk it fetches) and on the fragment reads.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    float* sX6 = lds; float* sW6 = lds + 2 * BM * 16;
    const int kc = (tid & 3) ^ ((tid >> 4) & 3);
    const float* gx0 = X + (size_t)(m0 + srow) * K + kc * 4; const float* gx1 = gx0 + (size_t)64 * K;
    const float* gw0 = W + (size_t)(n0 + srow) * K + kc * 4; const float* gw1 = gw0 + (size_t)64 * K;
    const unsigned lx = (unsigned)(size_t)(lds_void*)sX6 + __builtin_amdgcn_readfirstlane(wave) * 1024u;
    const unsigned lw = (unsigned)(size_t)(lds_void*)sW6 + __builtin_amdgcn_readfirstlane(wave) * 1024u;
#define GLDS1(gsrc, ldst) { unsigned keep; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory"); }
#define GLDS7(kt, buf) { GLDS1(gx0 + (kt) * BK, lx + (buf) * (BM * 64)); GLDS1(gx1 + (kt) * BK, lx + (buf) * (BM * 64) + 4096u); \
    GLDS1(gw0 + (kt) * BK, lw + (buf) * (BN * 64)); GLDS1(gw1 + (kt) * BK, lw + (buf) * (BN * 64) + 4096u); }
    __syncthreads();   // the register-staged prologue above wrote the padded layout; start clean
    GLDS7(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int sw = (lane >> 2) & 3, h = lane >> 5;
    const int o0 = (lane & 31) * 16 + ((h ^ sw) << 2), o1 = (lane & 31) * 16 + (((2 + h) ^ sw) << 2);
    const float* fw6 = sW6 + wn * 64 * 16; const float* fx6 = sX6 + wm * 64 * 16;
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) GLDS7(kt + 1, buf ^ 1);
      for (int i = 0; i < 2; ++i) {
        wf[i][0] = *(const float4*)(fw6 + buf * BN * 16 + i * 32 * 16 + o0); wf[i][1] = *(const float4*)(fw6 + buf * BN * 16 + i * 32 * 16 + o1);
        xf[i][0] = *(const float4*)(fx6 + buf * BM * 16 + i * 32 * 16 + o0); xf[i][1] = *(const float4*)(fx6 + buf * BM * 16 + i * 32 * 16 + o1); }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf[i][kk], t), f4e(xf[j][kk], t), acc[i][j], 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else
  if (VARIANT == 12) {
    // weights straight from L2 in MFMA operand order (W is read as if packed [n / 32][k16][kk][lane 64][4]: 1 KB per wave
    // instruction), one k-step ahead in registers; only the activations go through LDS
    float4 wq[2][2][2];
    const float* wbase = W + ((size_t)(n0 / 32 + wn * 2) * nk) * 512 + lane * 4;
#define LOADW12(SET, kt) { for (int i = 0; i < 2; ++i) for (int kk = 0; kk < 2; ++kk) wq[SET][i][kk] = *(const float4*)(wbase + ((size_t)i * nk + (kt)) * 512 + kk * 256); }
#define LOADX12(kt) { xr0 = *(const float4*)(xp0 + (kt) * BK); xr1 = *(const float4*)(xp1 + (kt) * BK); }
#define STOREX12(buf) { float* dx = sX + (buf) * BM * LR + srow * LR + chunk * 4; *(float4*)dx = xr0; *(float4*)(dx + 64 * LR) = xr1; }
#define STEP12(SET, buf) { \
      for (int kk = 0; kk < 2; ++kk) for (int i = 0; i < 2; ++i) xf[i][kk] = *(const float4*)(fx + (buf) * BM * LR + i * 32 * LR + kk * 8); \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int i = 0; i < 2; ++i) \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wq[SET][i][kk], t), f4e(xf[j][kk], t), acc[i][j], 0, 0, 0); }
    LOADW12(0, 0);
    for (int kt = 0; kt + 1 < nk; kt += 2) {
      LOADX12(kt + 1); LOADW12(1, kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      STEP12(0, 0);
      STOREX12(1);
      __syncthreads();
      { const int k2 = kt + 2 < nk ? kt + 2 : kt; LOADX12(k2); LOADW12(0, k2); }
      __builtin_amdgcn_sched_barrier(0);
      STEP12(1, 1);
      STOREX12(0);
      __syncthreads();
    }
  } else
  if (VARIANT == 5) {
    LOADT(1);
    for (int kt = 0; kt < nk - 1; ++kt) {
      const int buf = kt & 1;
      STORET(buf ^ 1);                        // tile kt+1 (loaded one iteration ago) -> other buffer
      { const int k2 = kt + 2 < nk ? kt + 2 : nk - 1; LOADT(k2); }
      for (int kk = 0; kk < 2; ++kk) for (int i = 0; i < 2; ++i) {
        wf[i][kk] = *(const float4*)(fw + buf * BN * LR + i * 32 * LR + kk * 8);
        xf[i][kk] = *(const float4*)(fx + buf * BM * LR + i * 32 * LR + kk * 8); }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf[i][kk], t), f4e(xf[j][kk], t), acc[i][j], 0, 0, 0);
      __syncthreads();
    }
  } else
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const int buf = (VARIANT == 1 || VARIANT == 8) ? 0 : (kt & 1);
    if (VARIANT == 0 || VARIANT == 2 || VARIANT == 3 || VARIANT == 8) LOADT(kt + 1);
    if (VARIANT != 2 && VARIANT != 4)
      for (int kk = 0; kk < 2; ++kk) for (int i = 0; i < 2; ++i) {
        wf[i][kk] = *(const float4*)(fw + buf * BN * LR + i * 32 * LR + kk * 8);
        xf[i][kk] = *(const float4*)(fx + buf * BM * LR + i * 32 * LR + kk * 8); }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf[i][kk], t), f4e(xf[j][kk], t), acc[i][j], 0, 0, 0);
    if (VARIANT == 0 || VARIANT == 2 || VARIANT == 3 || VARIANT == 9) STORET(buf ^ 1);
    if (VARIANT == 8) { asm volatile("" :: "v"(xr0.x), "v"(xr0.y), "v"(xr0.z), "v"(xr0.w), "v"(xr1.x), "v"(xr1.y), "v"(xr1.z), "v"(xr1.w));
                        asm volatile("" :: "v"(wr0.x), "v"(wr0.y), "v"(wr0.z), "v"(wr0.w), "v"(wr1.x), "v"(wr1.y), "v"(wr1.z), "v"(wr1.w)); }
    if (VARIANT != 4) __syncthreads();
  }
  const int half = lane >> 5;
  for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int q = 0; q < 4; ++q) {
    const int m = m0 + wm * 64 + j * 32 + (lane & 31), n = n0 + wn * 64 + i * 32 + q * 8 + half * 4;
    *(float4*)(D + (size_t)m * N + n) = make_float4(acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
  }
}

// VARIANT 10: 256x128 block tile, 8 waves (512 threads), 3 staging loads per thread per k-step
__global__ void __launch_bounds__(512, 2) gemm_kernel_big(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ D,
                                                          int M, int N, int K) {
  constexpr int BM = 256, BN = 128, BK = 16, LR = 20;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LR];
  float* sX = lds; float* sW = lds + 2 * BM * LR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int nbn = N / BN;
  const int m0 = (blockIdx.x / nbn) * BM, n0 = (blockIdx.x % nbn) * BN;
  const int srow = tid >> 2, chunk = tid & 3;
  const float* xp0 = X + (size_t)(m0 + srow) * K + chunk * 4; const float* xp1 = xp0 + (size_t)128 * K;
  const float* wp0 = W + (size_t)(n0 + srow) * K + chunk * 4;
  float4 xr0, xr1, wr0;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
#define LOADB(kt) { xr0 = *(const float4*)(xp0 + (kt) * BK); xr1 = *(const float4*)(xp1 + (kt) * BK); wr0 = *(const float4*)(wp0 + (kt) * BK); }
#define STOREB(buf) { float* dx = sX + (buf) * BM * LR + srow * LR + chunk * 4; float* dw = sW + (buf) * BN * LR + srow * LR + chunk * 4; \
    *(float4*)dx = xr0; *(float4*)(dx + 128 * LR) = xr1; *(float4*)dw = wr0; }
  LOADB(0); STOREB(0); __syncthreads();
  const int frag = (lane & 31) * LR + (lane >> 5) * 4;
  const float* fw = sW + wn * 64 * LR + frag; const float* fx = sX + wm * 64 * LR + frag;
  float4 wf[2][2], xf[2][2];
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const int buf = kt & 1;
    LOADB(kt + 1);
    for (int kk = 0; kk < 2; ++kk) for (int i = 0; i < 2; ++i) {
      wf[i][kk] = *(const float4*)(fw + buf * BN * LR + i * 32 * LR + kk * 8);
      xf[i][kk] = *(const float4*)(fx + buf * BM * LR + i * 32 * LR + kk * 8); }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf[i][kk], t), f4e(xf[j][kk], t), acc[i][j], 0, 0, 0);
    STOREB(buf ^ 1);
    __syncthreads();
  }
  const int half = lane >> 5;
  for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int q = 0; q < 4; ++q) {
    const int m = m0 + wm * 64 + j * 32 + (lane & 31), n = n0 + wn * 64 + i * 32 + q * 8 + half * 4;
    *(float4*)(D + (size_t)m * N + n) = make_float4(acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
  }
}
void run_big(const float* X, const float* W, float* D, int M, int N, int K) {
  dim3 g((M / 256) * (N / 128));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_kernel_big, g, dim3(512), 0, 0, X, W, D, M, N, K);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); const int R = 20;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(gemm_kernel_big, g, dim3(512), 0, 0, X, W, D, M, N, K);
  hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= R;
  printf("%-34s %8.1f us  %6.1f TF/s\n", "10 tile 256x128, 8 waves", ms * 1e3, 2.0 * M * N * K / ms / 1e9);
}

// VARIANT 11: model of a patch-based conv k-loop: X fragments from a resident LDS tile (no staging),
// W fragments straight from global / L2 in operand order (no LDS), one barrier every 9 k-steps.
__global__ void __launch_bounds__(256, 2) gemm_kernel_patch(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ D,
                                                            int M, int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 16, LR = 20;
  __shared__ __attribute__((aligned(16))) float lds[2 * BM * LR];
  float* sX = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int nbn = N / BN;
  const int m0 = (blockIdx.x / nbn) * BM, n0 = (blockIdx.x % nbn) * BN;
  for (int i = tid; i < 2 * BM * LR; i += 256) lds[i] = X[(size_t)m0 * K + i % 4096];
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  const int frag = (lane & 31) * LR + (lane >> 5) * 4;
  const float* fx = sX + wm * 64 * LR + frag;
  const float* wg = W + (size_t)(n0 + wn * 64 + (lane & 31)) * K + (lane >> 5) * 4;
  float4 wf[2][2], xf[2][2];
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    for (int kk = 0; kk < 2; ++kk) for (int i = 0; i < 2; ++i) {
      wf[i][kk] = *(const float4*)(wg + (size_t)i * 32 * K + kt * BK + kk * 8);
      xf[i][kk] = *(const float4*)(fx + buf * BM * LR + i * 32 * LR + kk * 8); }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(wf[i][kk], t), f4e(xf[j][kk], t), acc[i][j], 0, 0, 0);
    if (kt % 9 == 8) __syncthreads();
  }
  const int half = lane >> 5;
  for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int q = 0; q < 4; ++q) {
    const int m = m0 + wm * 64 + j * 32 + (lane & 31), n = n0 + wn * 64 + i * 32 + q * 8 + half * 4;
    *(float4*)(D + (size_t)m * N + n) = make_float4(acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
  }
}
void run_patch(const float* X, const float* W, float* D, int M, int N, int K) {
  dim3 g((M / 128) * (N / 128));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_kernel_patch, g, dim3(256), 0, 0, X, W, D, M, N, K);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); const int R = 20;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(gemm_kernel_patch, g, dim3(256), 0, 0, X, W, D, M, N, K);
  hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= R;
  printf("%-34s %8.1f us  %6.1f TF/s\n", "11 patch model: X in LDS, W from L2", ms * 1e3, 2.0 * M * N * K / ms / 1e9);
}
template <int V> void run(const float* X, const float* W, float* D, int M, int N, int K, const char* name) {
  dim3 g((M / 128) * (N / 128));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_kernel<V>, g, dim3(256), 0, 0, X, W, D, M, N, K);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); const int R = 20;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(gemm_kernel<V>, g, dim3(256), 0, 0, X, W, D, M, N, K);
  hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= R;
  printf("%-34s %8.1f us  %6.1f TF/s\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 24576, N = argc > 2 ? atoi(argv[2]) : 3840, K = argc > 3 ? atoi(argv[3]) : 1280;
  float *X, *W, *D; hipMalloc(&X, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&D, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2 - 1; hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  h.resize((size_t)N * K); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2 - 1; hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("M=%d N=%d K=%d\n", M, N, K);
  for (int round = 0; round < 3; ++round) {
    printf("-- round %d\n", round);
    run<0>(X, W, D, M, N, K, "0 full");
    run<5>(X, W, D, M, N, K, "5 store-at-top, load 2 ahead");
    run<1>(X, W, D, M, N, K, "1 no global loads / ds_write");
    run<2>(X, W, D, M, N, K, "2 no ds_read (reg operands)");
    run<3>(X, W, D, M, N, K, "3 full, 2 blocks/CU");
    run<4>(X, W, D, M, N, K, "4 MFMA only");
    run_big(X, W, D, M, N, K);
    run_patch(X, W, D, M, N, K);
    run<8>(X, W, D, M, N, K, "8 global loads, no ds_write");
    run<9>(X, W, D, M, N, K, "9 ds_write, no global loads");
    run<6>(X, W, D, M, N, K, "6 global_load_lds direct");
    run<7>(X, W, D, M, N, K, "7 global_load_lds asm, own waits");
    run<12>(X, W, D, M, N, K, "12 W operand-ordered from L2, X via LDS");
  }
  for (int var = 6; var <= 7; ++var) {  // spot-check variants 6, 7 against a host dot product
    hipMemset(D, 0, (size_t)M * N * 4);
    if (var == 6) hipLaunchKernelGGL(gemm_kernel<6>, dim3((M / 128) * (N / 128)), dim3(256), 0, 0, X, W, D, M, N, K);
    else hipLaunchKernelGGL(gemm_kernel<7>, dim3((M / 128) * (N / 128)), dim3(256), 0, 0, X, W, D, M, N, K);
    hipDeviceSynchronize();
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hd((size_t)M * N);
    hipMemcpy(hx.data(), X, hx.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hd.data(), D, hd.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0; for (int s = 0; s < 4000; ++s) { const int m = rand() % M, n = rand() % N; double acc = 0;
      for (int k = 0; k < K; ++k) acc += (double)hx[(size_t)m * K + k] * hw[(size_t)n * K + k];
      const double e = fabs(acc - hd[(size_t)m * N + n]); if (e > worst) worst = e; }
    printf("variant %d spot check: max abs err %.3e (K=%d)\n", var, worst, K);
  }
  return 0;
}
