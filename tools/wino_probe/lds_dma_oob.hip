// Probe: does `buffer_load_dwordx4 ... offen lds` write ZEROS into LDS for lanes whose offset is out of range?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* in, float* out, int soff) {
  __shared__ float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -7.f;     // poison
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)0x80000000u, 0x00020000);
  int vo = (threadIdx.x % 3 == 1) ? (int)0x80000000u : (int)threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, vo, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 1.0f + i;
  float *din, *dout;
  hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 256 * 4);
  hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, dout, 64);
  std::vector<float> o(256);
  hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
  int bad = 0, zeros = 0, poison = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      float v = o[l * 4 + e];
      if (l % 3 == 1) { if (v == 0.f) ++zeros; else if (v == -7.f) ++poison; else ++bad; }
      else if (v != 1.0f + 16 + l * 4 + e) ++bad;
    }
  printf("lds_dma_oob: in-range mismatches+other=%d oob_zero=%d oob_untouched=%d (of %d oob elems)\n", bad, zeros, poison, 21 * 4 + 4);
  return 0;
}
