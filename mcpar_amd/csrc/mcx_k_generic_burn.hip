// mcx_k_generic_burn.hip -- instantiations of the generic fused kernel k_fused_steps<LPC, LIK, false>
// (mcx_device.hpp): any np <= 256, full covariance, accept-mask recording, every fusable likelihood
#include "mcx_launch.hpp"

using namespace mcx;

template <int LPC>
static hipError_t by_lik(int lik, const SegArgs &a, hipStream_t st)
{
  const dim3 grid((unsigned)(((size_t)a.n * LPC + BLOCK - 1) / BLOCK)), block(BLOCK);
  switch (lik) {
  case LIK_ROSEN1: hipLaunchKernelGGL((k_fused_steps<LPC, LIK_ROSEN1, false>), grid, block, 0, st, a); break;
  case LIK_GAUSS: hipLaunchKernelGGL((k_fused_steps<LPC, LIK_GAUSS, false>), grid, block, 0, st, a); break;
  case LIK_MIX: hipLaunchKernelGGL((k_fused_steps<LPC, LIK_MIX, false>), grid, block, 0, st, a); break;
  case LIK_ROSEN2F: hipLaunchKernelGGL((k_fused_steps<LPC, LIK_ROSEN2F, false>), grid, block, 0, st, a); break;
  default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t mcxk_launch_generic_burn(int lpc, int lik, const SegArgs &a, hipStream_t st)
{
  switch (lpc) {
  case 1: return by_lik<1>(lik, a, st);
  case 2: return by_lik<2>(lik, a, st);
  case 4: return by_lik<4>(lik, a, st);
  case 8: return by_lik<8>(lik, a, st);
  case 16: return by_lik<16>(lik, a, st);
  case 32: return by_lik<32>(lik, a, st);
  case 64: return by_lik<64>(lik, a, st);
  default: return hipErrorInvalidValue;
  }
}
