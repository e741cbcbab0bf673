// common.h -- shared helpers for the gfx950 kernels of libhands_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define HANDS_LAUNCH_CHECK() return (int)hipGetLastError()

static inline int hands_grid_1d(long long work, int block, int cap = 256 * 8) {
  long long g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
