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

// steps per phase: the generators' count (16 - owners - recorders).  Measured on 8-D x 4096 chains (one owner,
// 14 generators): 14 steps per phase 0.315 ms per job, 28 (the LDS double buffers would hold it) 0.355, 10 -> 0.364,
// 7 -> 0.443: a phase must be long enough for one wavefront to finish a two-step generator item behind it.
int mcxk_persist_ksteps(int lpc, int own)
{
  (void)lpc;
  const int rec = mcxk_persist_recorders(own) ? 1 : 0;
  return PWAVES - own - rec * own;
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
  hipLaunchKernelGGL((k_run_small<LPC, LIK, REC>), dim3(nwg), dim3(PBLOCK), mcxk_persist_lds_bytes(LPC, a.own), st, a);
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
