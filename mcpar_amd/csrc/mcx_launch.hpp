// mcx_launch.hpp -- launchers of the two fused-kernel families, each compiled in its own translation
// unit so that libmcx.so builds in parallel.  Return hipErrorInvalidValue for a (lanes per chain,
// likelihood) pair that has no instantiation, otherwise the launch status.
#pragma once
#include "mcx_device.hpp"

hipError_t mcxk_launch_fast(int lpc, int lik, bool main, const mcx::SegArgs &a, hipStream_t st);   // mcx_k_fast.hip
hipError_t mcxk_launch_generic_burn(int lpc, int lik, const mcx::SegArgs &a, hipStream_t st);      // mcx_k_generic_burn.hip
hipError_t mcxk_launch_generic_main(int lpc, int lik, const mcx::SegArgs &a, hipStream_t st);      // mcx_k_generic_main.hip
