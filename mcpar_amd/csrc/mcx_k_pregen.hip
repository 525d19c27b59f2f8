// mcx_k_pregen.hip -- small-n mode: k_gen_normals<LPC> and k_fused_fast<LPC, MAIN, LIK, PREGEN = true>
#include "mcx_launch.hpp"

using namespace mcx;

template <int LPC, int LIK>
static hipError_t go(bool main, const SegArgs &a, hipStream_t st)
{
  const dim3 grid((unsigned)(((size_t)a.n * LPC + BLOCK - 1) / BLOCK)), block(BLOCK);
  if (main) hipLaunchKernelGGL((k_fused_fast<LPC, true, LIK, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_fused_fast<LPC, false, LIK, true>), grid, block, 0, st, a);
  return hipGetLastError();
}

template <int LPC>
static hipError_t by_lik(int lik, bool main, const SegArgs &a, hipStream_t st)
{
  switch (lik) {
  case LIK_ROSEN1: return go<LPC, LIK_ROSEN1>(main, a, st);
  case LIK_GAUSS: return go<LPC, LIK_GAUSS>(main, a, st);
  case LIK_MIX: return go<LPC, LIK_MIX>(main, a, st);
  default: return hipErrorInvalidValue;
  }
}

hipError_t mcxk_launch_fast_pregen(int lpc, int lik, bool main, const SegArgs &a, hipStream_t st)
{
  switch (lpc) {
  case 1: return by_lik<1>(lik, main, a, st);
  case 2: return by_lik<2>(lik, main, a, st);
  case 4: return by_lik<4>(lik, main, a, st);
  case 8: return by_lik<8>(lik, main, a, st);
  default: return hipErrorInvalidValue;
  }
}

template <int LPC>
static hipError_t gen(float *Z, float *U, int n, int d, int nsteps, uint32_t t0, uint32_t g0, uint32_t seed, hipStream_t st)
{
  const size_t lanes = (size_t)nsteps * n * LPC;
  hipLaunchKernelGGL((k_gen_normals<LPC>), dim3((unsigned)((lanes + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, Z, U, n, d,
                     nsteps, t0, g0, seed);
  return hipGetLastError();
}

hipError_t mcxk_launch_gen(int lpc, float *Z, float *U, int n, int d, int nsteps, uint32_t t0, uint32_t g0,
                           uint32_t seed, hipStream_t st)
{
  switch (lpc) {
  case 1: return gen<1>(Z, U, n, d, nsteps, t0, g0, seed, st);
  case 2: return gen<2>(Z, U, n, d, nsteps, t0, g0, seed, st);
  case 4: return gen<4>(Z, U, n, d, nsteps, t0, g0, seed, st);
  case 8: return gen<8>(Z, U, n, d, nsteps, t0, g0, seed, st);
  default: return hipErrorInvalidValue;
  }
}
