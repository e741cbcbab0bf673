// How many independent v_mfma_f32_16x16x4_f32 accumulation chains does a SIMD need?  (MANO kernel sizing.)
// Build: hipcc -O3 --offload-arch=gfx950 mfma16.hip -o mfma16 ; run: ./mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH>
__global__ void __launch_bounds__(256) chain_kernel(float* out, int iters) {
  f32x4 acc[NCH];
  for (int i = 0; i < NCH; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32 / NCH; ++u)
#pragma unroll
      for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NCH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH> void run(float* o, int blocks_per_cu) {
  const int iters = 4000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(chain_kernel<NCH>, dim3(grid), dim3(256), 0, 0, o, iters);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(chain_kernel<NCH>, dim3(grid), dim3(256), 0, 0, o, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)grid * 4 * iters * 32 * 2048.0;
  const double mfma_per_simd = (double)blocks_per_cu * iters * 32;      // one wave of each block per SIMD
  printf("chains/wave %d, waves/SIMD %d: %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", NCH, blocks_per_cu,
         flop / ms / 1e9, ms * 1e-3 * 2.4e9 / mfma_per_simd);
}

int main() {
  float* o; hipMalloc(&o, 256 * 4096 * 4);
  for (int w = 1; w <= 4; w *= 2) { run<1>(o, w); run<2>(o, w); run<4>(o, w); }
  return 0;
}
