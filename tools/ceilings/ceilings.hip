// On-box chip ceilings: a pure fp32-MFMA loop and streaming HBM kernels.
// Build: hipcc -O3 --offload-arch=gfx950 ceilings.hip -o ceilings ; run: ./ceilings
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ void __launch_bounds__(256) read_kernel(const float4* __restrict__ in, float* out, size_t n) {
  float4 s = make_float4(0, 0, 0, 0);
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = in[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}
__global__ void __launch_bounds__(256) write_kernel(float4* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// non-temporal variants (nt bit on the instruction): do streaming stores / loads get a better rate past the L2?
__global__ void __launch_bounds__(256) copy_nt_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n, int mode) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const f32x4 v = (mode & 1) ? __builtin_nontemporal_load(&in[i]) : in[i];
    if (mode & 2) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
  }
}
__global__ void __launch_bounds__(256) write_nt_kernel(f32x4* __restrict__ out, size_t n) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, &out[i]);
}

template <class F> float time_ms(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
  float* o; hipMalloc(&o, 256 * 4096 * 4);
  // warm the clocks
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(mfma_loop, dim3(4096), dim3(256), 0, 0, o, 2000);
  hipDeviceSynchronize();
  for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
    const int blocks = 256 * waves_per_simd * 4;   // 4 waves per block = one per SIMD
    const int iters = 4000;
    const float ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, o, iters); }, 5);
    const double flop = (double)blocks * 4 /*waves*/ * iters * 32 /*mfma*/ * (2.0 * 32 * 32 * 2);
    printf("mfma_f32_32x32x2 loop, %d wave(s)/SIMD: %.1f TFLOP/s (datasheet 157.3)\n", waves_per_simd, flop / ms / 1e9);
  }
  const size_t bytes = (size_t)4 << 30;   // 4 GiB buffers: far beyond the 256 MB MALL
  float4 *x, *y; hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
  const size_t n = bytes / 16;
  for (int grid : {2048, 8192, 32768}) {
    const float mc = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, x, y, n); }, 5);
    const float mr = time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, x, o, n); }, 5);
    const float mw = time_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, y, n); }, 5);
    printf("HBM stream, grid %5d: copy %.2f TB/s (read+write), read-only %.2f TB/s, write-only %.2f TB/s (datasheet 8.0)\n",
           grid, 2.0 * bytes / mc / 1e9, (double)bytes / mr / 1e9, (double)bytes / mw / 1e9);
  }
  for (int mode = 0; mode < 4; ++mode) {
    const float mc = time_ms([&] { hipLaunchKernelGGL(copy_nt_kernel, dim3(8192), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, n, mode); }, 5);
    printf("copy, grid 8192, %s loads, %s stores: %.2f TB/s (read+write)\n", (mode & 1) ? "nt" : "plain", (mode & 2) ? "nt" : "plain",
           2.0 * bytes / mc / 1e9);
  }
  {
    const float mw = time_ms([&] { hipLaunchKernelGGL(write_nt_kernel, dim3(8192), dim3(256), 0, 0, (f32x4*)y, n); }, 5);
    printf("write-only, nt stores: %.2f TB/s\n", (double)bytes / mw / 1e9);
  }
  return 0;
}
