// mcx_k_fastb_full.hip -- k_fused_fastb<LPC2, 2, MAIN, LIK, FULL = true> (mcx_fastb.hpp): full-covariance proposals with two
// mirrored blocks per lane -- the first blocks' multiply-adds above the factor's diagonal are never issued
#include "mcx_fastb.hpp"
#include "mcx_launch.hpp"

using namespace mcx;

template <int LPC2, int LIK>
static hipError_t go(bool main, const SegArgs &a, hipStream_t st)
{
  const dim3 grid((unsigned)(((size_t)a.n * LPC2 + BLOCK - 1) / BLOCK)), block(BLOCK);
  if (main) hipLaunchKernelGGL((k_fused_fastb<LPC2, 2, true, LIK, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_fused_fastb<LPC2, 2, false, LIK, true>), grid, block, 0, st, a);
  return hipGetLastError();
}

template <int LPC2>
static hipError_t by_lik(int lik, bool main, const SegArgs &a, hipStream_t st)
{
  switch (lik) {
  case LIK_ROSEN1: return go<LPC2, LIK_ROSEN1>(main, a, st);
  case LIK_GAUSS: return go<LPC2, LIK_GAUSS>(main, a, st);
  case LIK_MIX: return go<LPC2, LIK_MIX>(main, a, st);
  default: return hipErrorInvalidValue;
  }
}

// lpc = blocks per chain: 4 (np = 16) or 8 (np = 32)
hipError_t mcxk_launch_fastb_full(int lpc, int lik, bool main, const SegArgs &a, hipStream_t st)
{
  switch (lpc) {
  case 4: return by_lik<2>(lik, main, a, st);
  case 8: return by_lik<4>(lik, main, a, st);
  default: return hipErrorInvalidValue;
  }
}
