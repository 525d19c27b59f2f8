// sync_latency_probe.hip -- how long does the host take to learn that a lone kernel on an idle queue has finished?
// The small-n job (one launch of ~0.39 ms) pays launch + completion once per run: 25-30 us of its 0.418 ms (EXPERIMENTS.md).
// Ways to wait, timed around the same ~100 us kernel (kernel time by events subtracted):
//   sync    hipStreamSynchronize
//   copy    a 64-byte hipMemcpyAsync device -> pinned host behind the kernel, then hipStreamSynchronize (what mcx_run does)
//   event   spin on hipEventQuery of an event recorded behind the kernel
//   value   hipStreamWriteValue32 of a serial number into pinned host memory behind the kernel, host spins on the word
//   kernel  the kernel itself stores the serial to pinned host memory (system scope) as its last act, host spins
// build: hipcc -O2 --offload-arch=gfx950 tools/sync_latency_probe.hip -o /tmp/sync_latency_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long ticks, volatile unsigned *host_flag, unsigned serial, unsigned *dev_count)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (host_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(dev_count, 1u) == gridDim.x - 1u) {  // the grid's last workgroup
        *dev_count = 0;
        __hip_atomic_store(const_cast<unsigned *>(host_flag), serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned *flag = nullptr, *dcount = nullptr;
  CHK(hipHostMalloc(&flag, 64, hipHostMallocMapped));
  CHK(hipMalloc(&dcount, 4));
  CHK(hipMemset(dcount, 0, 4));
  *flag = 0;
  hipEvent_t ea, eb;
  CHK(hipEventCreate(&ea));
  CHK(hipEventCreate(&eb));
  const unsigned long long ticks = 10000;  // 100 us at 100 MHz
  const int grid = 256, reps = 200;
  unsigned serial = 0;
  const char *names[] = {"sync", "event", "value", "kernel", "copy"};
  unsigned long long *dctr = nullptr, *hctr = nullptr;
  CHK(hipMalloc(&dctr, 64));
  CHK(hipMemset(dctr, 0, 64));
  CHK(hipHostMalloc(&hctr, 64, hipHostMallocDefault));
  for (int mode = 0; mode < 5; ++mode) {
    std::vector<double> over;
    for (int r = 0; r < reps + 20; ++r) {
      ++serial;
      const double t0 = now_us();
      CHK(hipEventRecord(ea, st));
      hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, st, ticks, mode == 3 ? flag : nullptr, serial, dcount);
      CHK(hipEventRecord(eb, st));
      if (mode == 0) {
        CHK(hipStreamSynchronize(st));
      } else if (mode == 4) {
        CHK(hipMemcpyAsync(hctr, dctr, 64, hipMemcpyDeviceToHost, st));
        CHK(hipStreamSynchronize(st));
      } else if (mode == 1) {
        while (hipEventQuery(eb) == hipErrorNotReady) {}
      } else if (mode == 2) {
        CHK(hipStreamWriteValue32(st, flag, serial, 0));
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != serial) __builtin_ia32_pause();
      } else {
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != serial) __builtin_ia32_pause();
      }
      const double t1 = now_us();
      CHK(hipStreamSynchronize(st));
      float ms = 0;
      CHK(hipEventElapsedTime(&ms, ea, eb));
      if (r >= 20) over.push_back((t1 - t0) - ms * 1e3);
    }
    std::sort(over.begin(), over.end());
    printf("%-6s host wall - kernel time: median %.1f us, p10 %.1f, p90 %.1f\n", names[mode], over[over.size() / 2], over[over.size() / 10],
           over[over.size() * 9 / 10]);
  }
  return 0;
}
