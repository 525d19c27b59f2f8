// occupy.hip -- test helper (tests/test_gpu_coresidency.py): a "foreign" kernel that holds CUs for a while.
// nblocks workgroups of `threads` threads, each holding `lds_bytes` of LDS, spin on the 100 MHz wall clock for `ms`
// milliseconds (bounded: every wavefront leaves when the time is up) on a stream of their own.  Built at test time with
// hipcc -shared; lives in the test process next to libmcx.so, which knows nothing of it.
#include <hip/hip_runtime.h>
#include <cstdint>

extern "C" {

__global__ void k_occupy(unsigned long long ticks, unsigned *sink)
{
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;  // (the LDS is really used: the allocation is what matters)
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = lds[(threadIdx.x * 7u) % blockDim.x];
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    acc += 1u;
  }
  if (acc == 0xffffffffu) sink[0] = acc;
}

struct occupy_t {
  hipStream_t st;
  unsigned *sink;
};

int occupy_start(int nblocks, int threads, int lds_bytes, double ms, void **handle)
{
  if (nblocks < 1 || threads < 64 || threads > 1024 || lds_bytes < 4 * threads || lds_bytes > 160 * 1024 || ms <= 0.0 || ms > 2000.0 || !handle) return -1;
  occupy_t *o = new occupy_t{};
  if (hipStreamCreateWithFlags(&o->st, hipStreamNonBlocking) != hipSuccess) return -2;
  if (hipMalloc(&o->sink, 64) != hipSuccess) return -3;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -4;
  hipLaunchKernelGGL(k_occupy, dim3(nblocks), dim3(threads), lds_bytes, o->st, (unsigned long long)(ms * 1e5), o->sink);
  if (hipGetLastError() != hipSuccess) return -5;
  *handle = o;
  return 0;
}

// 1 when the kernel has finished, 0 when it is still running
int occupy_done(void *handle)
{
  occupy_t *o = static_cast<occupy_t *>(handle);
  return hipStreamQuery(o->st) == hipSuccess ? 1 : 0;
}

int occupy_wait(void *handle)
{
  occupy_t *o = static_cast<occupy_t *>(handle);
  const hipError_t e = hipStreamSynchronize(o->st);
  (void)hipFree(o->sink);
  (void)hipStreamDestroy(o->st);
  delete o;
  return e == hipSuccess ? 0 : -1;
}
}
