// A user-written likelihood as a GPU kernel (MCX_VL_DEVICE): the VLFunc contract of
// src/vlfunc.hh:9-12 on device memory.  Rosenbrock1 for d = 8 with the operation order of
// MCX arithmetic v1, so that the run can be compared with the built-in bit for bit.
#include <hip/hip_runtime.h>

extern "C" __global__ void user_rosenbrock8(int npset, const float *x, float *y)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npset) return;
  const float *p = x + (size_t)j * 8;
  float part[2];
  for (int q = 0; q < 2; ++q) {
    float acc = 0.0f;
    for (int k = 4 * q; k < 4 * q + 4; k += 2) {
      const float t1 = 1.0f - p[k];
      const float t2 = __builtin_fmaf(-p[k], p[k], p[k + 1]);
      acc = acc + __builtin_fmaf(100.0f * t2, t2, t1 * t1);
    }
    part[q] = acc;
  }
  y[j] = 0.0f - (part[0] + part[1]);
}
