// mcx_k_fastb.hip -- instantiations of k_fused_fastb<LPC2, BPL, MAIN, LIK> (mcx_fastb.hpp): the hot-path kernel with
// two or four 4-parameter blocks per lane
#include "mcx_fastb.hpp"
#include "mcx_launch.hpp"

using namespace mcx;

template <int LPC2, int BPL, int LIK>
static hipError_t go(bool main, const SegArgs &a, hipStream_t st)
{
  const dim3 grid((unsigned)(((size_t)a.n * LPC2 + BLOCK - 1) / BLOCK)), block(BLOCK);
  if (main) hipLaunchKernelGGL((k_fused_fastb<LPC2, BPL, true, LIK>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_fused_fastb<LPC2, BPL, false, LIK>), grid, block, 0, st, a);
  return hipGetLastError();
}

template <int LPC2, int BPL>
static hipError_t by_lik(int lik, bool main, const SegArgs &a, hipStream_t st)
{
  switch (lik) {
  case LIK_ROSEN1: return go<LPC2, BPL, LIK_ROSEN1>(main, a, st);
  case LIK_GAUSS: return go<LPC2, BPL, LIK_GAUSS>(main, a, st);
  case LIK_MIX: return go<LPC2, BPL, LIK_MIX>(main, a, st);
  default: return hipErrorInvalidValue;
  }
}

// lpc = blocks per chain (next power of two of ceil(d / 4)), bpl = blocks per lane (2 or 4, <= lpc)
hipError_t mcxk_launch_fastb(int lpc, int bpl, int lik, bool main, const SegArgs &a, hipStream_t st)
{
  if (bpl == 2) {
    switch (lpc) {
    case 2: return by_lik<1, 2>(lik, main, a, st);
    case 4: return by_lik<2, 2>(lik, main, a, st);
    case 8: return by_lik<4, 2>(lik, main, a, st);
    default: return hipErrorInvalidValue;
    }
  }
  if (bpl == 4) {
    switch (lpc) {
    case 4: return by_lik<1, 4>(lik, main, a, st);
    case 8: return by_lik<2, 4>(lik, main, a, st);
    default: return hipErrorInvalidValue;
    }
  }
  return hipErrorInvalidValue;
}
