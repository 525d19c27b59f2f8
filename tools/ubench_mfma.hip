// mfma_k.hip -- cycles per matrix instruction on gfx950: v_mfma_f32_32x32x16_bf16 against the older K = 8 form
// (v_mfma_f32_32x32x8_bf16_1k), one wavefront per SIMD and four, dependent chains on one accumulator.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_mfma tools/ubench_mfma.hip ; run on the GPU box.
// Measured: 16.1-18.4 ns per instruction per SIMD for BOTH forms (= 32 cycles at the 1.95 GHz the chip holds under
// this load): a K of 8 costs what a K of 16 costs, and the dense bf16 peak one can sustain is 2.0 PFLOP/s, not 2.5.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITER 2048

__global__ void k16(float *out, float seed)
{
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i + threadIdx.x); b[i] = (__bf16)(seed - i); }
  f32x16 acc = {0};
  for (int i = 0; i < ITER; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k8(float *out, float seed)
{
  s16x4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = (short)(0x3f80 + i + threadIdx.x); b[i] = (short)(0x3f80 - i); }
  f32x16 acc = {0};
  for (int i = 0; i < ITER; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, acc, 0, 0, 0);
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + seed;
}

int main()
{
  float *out;
  hipMalloc(&out, 1 << 24);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int which = 0; which < 2; ++which) {
      const dim3 g(ncu * wps), b(256);  // 4 wavefronts per workgroup = one per SIMD
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k16, g, b, 0, 0, out, 1.0f);
        else hipLaunchKernelGGL(k8, g, b, 0, 0, out, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double per = ms * 1e6 / (double)ITER / wps;  // ns per instruction per SIMD
      printf("%s, %d wavefront(s) per SIMD: %.2f ns per instruction per SIMD (%.1f cycles at 2.4 GHz)\n",
             which == 0 ? "32x32x16_bf16   " : "32x32x8_bf16_1k ", wps, per, per * 2.4);
    }
  }
  return 0;
}
