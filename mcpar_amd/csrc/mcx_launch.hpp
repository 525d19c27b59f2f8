// mcx_launch.hpp -- launchers of the two fused-kernel families, each compiled in its own translation
// unit so that libmcx.so builds in parallel.  Return hipErrorInvalidValue for a (lanes per chain,
// likelihood) pair that has no instantiation, otherwise the launch status.
#pragma once
#include "mcx_device.hpp"

hipError_t mcxk_launch_fast(int lpc, int lik, bool main, const mcx::SegArgs &a, hipStream_t st);   // mcx_k_fast.hip
// the same with bpl = 2 or 4 consecutive blocks per lane (mcx_fastb.hpp), bpl <= lpc
hipError_t mcxk_launch_fastb(int lpc, int bpl, int lik, bool main, const mcx::SegArgs &a, hipStream_t st);  // mcx_k_fastb.hip
// full lower-triangular factor a.T (a.diag == 0), np <= 32, np % 4 == 0
hipError_t mcxk_launch_fast_full(int lpc, int lik, bool main, const mcx::SegArgs &a, hipStream_t st);  // mcx_k_fast_full.hip
// the same with two mirrored blocks per lane (mcx_fastb.hpp, FULL): lpc = 4 or 8
hipError_t mcxk_launch_fastb_full(int lpc, int lik, bool main, const mcx::SegArgs &a, hipStream_t st);  // mcx_k_fastb_full.hip
// small-n mode: a.zpre / a.upre must hold the output of mcxk_launch_gen for the same (t0, nsteps)
hipError_t mcxk_launch_fast_pregen(int lpc, int lik, bool main, const mcx::SegArgs &a, hipStream_t st);  // mcx_k_pregen.hip
hipError_t mcxk_launch_gen(int lpc, float *Z, float *U, int n, int d, int nsteps, uint32_t t0, uint32_t g0,
                           uint32_t seed, hipStream_t st);                                             // mcx_k_pregen.hip
hipError_t mcxk_launch_generic_burn(int lpc, int lik, const mcx::SegArgs &a, hipStream_t st);      // mcx_k_generic_burn.hip
hipError_t mcxk_launch_generic_main(int lpc, int lik, const mcx::SegArgs &a, hipStream_t st);      // mcx_k_generic_main.hip
// small-n mode, one launch per stretch of local steps (mcx_persist.hpp); every workgroup must be resident:
// ceil(a.nown / a.own) <= number of CUs, 1 <= a.own <= POWN_MAX
namespace mcx { struct RunArgs; }
// lpc = 4-parameter blocks per chain, bpl of them per lane (mcxk_persist_bpl's choice: 1, 2 or 4), lpc2 = lpc / bpl
hipError_t mcxk_launch_persist(int lpc, int bpl, int lik, const mcx::RunArgs &a, hipStream_t st);        // mcx_k_persist.hip
int mcxk_persist_bpl(int lpc, int d, int n, int ncu, int opt);
size_t mcxk_persist_lds_bytes(int lpc2, int bpl, int own);
int mcxk_persist_ksteps(int lpc2, int bpl, int own);
bool mcxk_persist_recorders(int own, int bpl);
// RunArgs::deal for a launch with `own` owner wavefronts per workgroup, K steps per phase: tab[3 * 16 * 12]
// (false: some wavefront's list overflowed -- mcxk_persist_ksteps never returns such a K)
bool mcxk_persist_deal(int lpc2, int bpl, int own, int rec, int K, uint32_t *tab);
constexpr int MCXK_PERSIST_DEAL_WORDS = 3 * 16 * 24;
constexpr size_t MCXK_PERSIST_LDS_LIMIT = (size_t)152 << 10;  // dynamic LDS a launch may ask for (160 KB per CU less the static part)
