// mcx_k_fast.hip -- instantiations of the hot-path kernel k_fused_fast<LPC, MAIN, LIK> (mcx_device.hpp)
#include "mcx_launch.hpp"

using namespace mcx;

template <int LPC, int LIK>
static hipError_t go(bool main, const SegArgs &a, hipStream_t st)
{
  const dim3 grid((unsigned)(((size_t)a.n * LPC + BLOCK - 1) / BLOCK)), block(BLOCK);
  if (main) hipLaunchKernelGGL((k_fused_fast<LPC, true, LIK>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_fused_fast<LPC, false, LIK>), grid, block, 0, st, a);
  return hipGetLastError();
}

template <int LPC>
static hipError_t by_lik(int lik, bool main, const SegArgs &a, hipStream_t st)
{
  switch (lik) {
  case LIK_ROSEN1: return go<LPC, LIK_ROSEN1>(main, a, st);
  case LIK_GAUSS: return go<LPC, LIK_GAUSS>(main, a, st);
  case LIK_MIX: return go<LPC, LIK_MIX>(main, a, st);
  case LIK_ROSEN2F: return go<LPC, LIK_ROSEN2F>(main, a, st);
  default: return hipErrorInvalidValue;
  }
}

hipError_t mcxk_launch_fast(int lpc, int lik, bool main, const SegArgs &a, hipStream_t st)
{
  switch (lpc) {
  case 1: return by_lik<1>(lik, main, a, st);
  case 2: return by_lik<2>(lik, main, a, st);
  case 4: return by_lik<4>(lik, main, a, st);
  case 8: return by_lik<8>(lik, main, a, st);
  default: return hipErrorInvalidValue;
  }
}
