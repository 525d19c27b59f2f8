// mcx_k_persist.hip -- instantiations and launcher of k_run_small<LPC, LIK> (mcx_persist.hpp)
#include "mcx_launch.hpp"
#include "mcx_persist.hpp"

#include <algorithm>
#include <cstdlib>

using namespace mcx;

// recorders (one per owner) where the owner's latency is the bound, not the generators' throughput: measured
// faster with one or two owner wavefronts per workgroup, equal with three, slower with four
bool mcxk_persist_recorders(int own) { return own <= 2; }

static size_t lds_for(int lpc, int own, int rec, int K)
{
  return (size_t)2 * (1 + rec) * K * own * 64 * sizeof(float4) + (size_t)2 * (1 + rec) * K * own * (64 / lpc) * sizeof(float);
}

// Steps per phase.  The default is the generators' count (16 - owners - recorders); with 1-3 owners per workgroup
// longer phases are faster (fewer phase changes for the latency-bound owners), by a table measured on one box with
// every even length the LDS buffers hold: 8-D x 4096 chains (1 owner; recorders not filling) 14 steps 0.290 ms per
// launch, 20 -> 0.308, 24 -> 0.274, 28 -> 0.269, 32 -> 0.268; 16-D x 8192 (2 owners) 12 -> 0.407,
// 14 -> 0.395, 16 -> 0.382; 16-D x 12 288 (3 owners) 13 -> 0.548, 16 -> 0.547, 18 -> 0.508, 20 -> 0.501, 22 -> 0.522;
// 16-D x 16 384 (4 owners) 8 -> 0.750, 10 -> 0.725, 12 -> 0.661, 14 -> 0.663, 16 -> 0.655.  The pattern is not monotonic: how the
// phase's items (a two-step item per owner and step pair, one acceptance item per owner) deal out over the filling
// wavefronts matters as much as the phase count.
int mcxk_persist_ksteps(int lpc, int own)
{
  const int rec = mcxk_persist_recorders(own) ? 1 : 0;
  int k = own == 1 ? 32 : (own == 2 ? 16 : (own == 3 ? 20 : (own == 4 ? 16 : PWAVES - own - rec * own)));
  while (k > 2 && (k > PKMAX || lds_for(lpc, own, rec, k) > MCXK_PERSIST_LDS_LIMIT)) k -= 2;
  return k;
}

size_t mcxk_persist_lds_bytes(int lpc, int own)
{
  return lds_for(lpc, own, mcxk_persist_recorders(own) ? 1 : 0, mcxk_persist_ksteps(lpc, own));
}

template <int LPC, int LIK, bool REC>
static hipError_t go2(const RunArgs &a, hipStream_t st)
{
  static bool attr_set = false;  // per instantiation; the value is the largest the kernel can be launched with
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_run_small<LPC, LIK, REC>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)MCXK_PERSIST_LDS_LIMIT);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const unsigned nwg = (unsigned)((a.nown + a.own - 1) / a.own);
  const size_t lds = mcxk_persist_lds_bytes(LPC, a.own);
  if (a.nburn > 0) {
    // tuner meetings inside: every workgroup of the grid must be resident at once.  What the occupancy
    // calculator and the device's CU count promise is checked here; what they cannot see (a CU mask, a foreign
    // kernel holding LDS) is caught by the meetings' own timeout.
    static size_t asked_lds = ~(size_t)0;  // (per instantiation: the answer depends on the LDS request only)
    static int asked_per_cu = 0;
    int per_cu = asked_per_cu, ncu = 0, dev = 0;
    hipError_t e = hipSuccess;
    if (asked_lds != lds) {
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_run_small<LPC, LIK, REC>), PBLOCK, lds);
      if (e == hipSuccess) { asked_per_cu = per_cu; asked_lds = lds; }
    }
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if ((long long)per_cu * ncu < (long long)nwg) return hipErrorCooperativeLaunchTooLarge;
  }
  hipLaunchKernelGGL((k_run_small<LPC, LIK, REC>), dim3(nwg), dim3(PBLOCK), lds, st, a);
  return hipGetLastError();
}

template <int LPC, int LIK>
static hipError_t go(const RunArgs &a, hipStream_t st)
{
  return mcxk_persist_recorders(a.own) ? go2<LPC, LIK, true>(a, st) : go2<LPC, LIK, false>(a, st);
}

template <int LPC>
static hipError_t by_lik(int lik, const RunArgs &a, hipStream_t st)
{
  switch (lik) {
  case LIK_ROSEN1: return go<LPC, LIK_ROSEN1>(a, st);
  case LIK_GAUSS: return go<LPC, LIK_GAUSS>(a, st);
  case LIK_MIX: return go<LPC, LIK_MIX>(a, st);
  default: return hipErrorInvalidValue;
  }
}

hipError_t mcxk_launch_persist(int lpc, int lik, const RunArgs &a, hipStream_t st)
{
  switch (lpc) {
  case 1: return by_lik<1>(lik, a, st);
  case 2: return by_lik<2>(lik, a, st);
  case 4: return by_lik<4>(lik, a, st);
  case 8: return by_lik<8>(lik, a, st);
  default: return hipErrorInvalidValue;
  }
}
