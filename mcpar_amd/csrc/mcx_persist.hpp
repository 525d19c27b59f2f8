// mcx_persist.hpp -- small-n mode, one launch per run: k_run_small<LPC2, BPL, LIK, REC>.
//
// With few chains (fewer wavefronts than SIMDs) a Metropolis step is bound by the latency of ONE wave's
// instruction stream, and every kernel boundary (tuner checks, chunk changes) costs more than the steps it
// separates.  This kernel runs a whole stretch of the schedule -- burn-in with the acceptance-rate tuner
// (src/mcpar.cc:55-97), the start of the main loop (:99-104) and the local main-loop steps (:152-209) -- in one
// launch of one 1024-thread workgroup per CU:
//
//   * OWN (1..4) "owner" wavefronts per workgroup hold the chains (x, ly) in registers, in the lane layout of
//     k_fused_fast, and do only what the NEXT step depends on: proposal = x + T z, likelihood,
//     log u < ly' - ly.  A lone wave issues one instruction per 5.7 cycles whatever the instruction (tools/ubench.hip), so
//     the owner's loop is kept to the bare dependent chain -- 21 vector instructions per burn-in step since round 5 (the
//     wavefront's index a scalar: uniform control flow; two steps per iteration: no register moves between "next" and
//     "this" step's numbers; S = -ly kept instead of ly); with the moments and the sample emission in it a step took
//     ~85 instructions and 690 cycles;
//   * OWN "recorder" wavefronts, one per owner, follow one phase behind: they read the owner's post-step
//     (x, ly) from LDS and do the Welford moments (src/mcpar.cc:184-209), the exchange snapshot and the sample
//     emission (:176-182) -- everything that consumes the state without feeding the next step;
//   * the remaining 16 - 2 OWN "generator" wavefronts (and the recorders when they have nothing to record)
//     produce the owners' random numbers -- Philox4x32-10 + Box-Muller normals and the logs of the acceptance
//     draws, two thirds of a step's instructions, none of which depend on the chain state -- K = 16 - 2 OWN
//     steps at a time into a double buffer in the CU's own LDS (items pulled from an LDS counter; one
//     s_barrier per K steps).  Nothing random ever touches L2 or HBM, and the SIMDs the owners leave idle
//     do the bulk of the arithmetic;
//   * the tuner's decision needs the accept count of ALL chains: at its check steps the workgroups meet at a
//     counter in global memory (one 64-bit atomic per workgroup: arrivals << 40 | accepts), every owner then
//     takes the same decision from the same integers and rescales its own copy of the Cholesky diagonal.
//
// Same functions, same bits as the per-segment kernels (tests/test_gpu_run_parity.py, test_gpu_fuzz.py).
// Every workgroup of a launch that contains tuner events must be resident at once: the grid is at most one
// workgroup per CU (checked against the kernel's occupancy at launch), and the host holds a per-GPU lock while
// such a launch is in flight (mcx_engine.hip, meet_lock_open), so no second kernel of this kind can take part of
// the CUs and leave two kernels waiting for each other.  That is a courtesy, not the guarantee: a CU mask, a
// partitioned device or a foreign kernel holding LDS can still keep workgroups out, so a meeting that is not
// complete after RunArgs::meet_timeout ticks of the 100 MHz wall clock is ABANDONED -- the waiting owner marks
// the meeting's leaves, every owner that sees the mark goes on with a count of zero and skips the meetings that
// follow, the launch runs to its end without its EPILOGUE (chain state x / ly / T, moments mu / psum2, accept counts,
// tuner counters, ntrace stay as they were), ctr[5] tells the host, and mcx_run repeats the run on the per-segment
// kernels.  What an abandoned launch DOES dirty, because the recording of a phase is not predicated on the flag (the
// owner / recorder wavefronts are the latency-critical ones): the sample rows samp_x / samp_ly of the steps it got to,
// the shard's musigall slot and sig (the snapshot / final publish), trace[].  All of them are rewritten in full by the
// repeated run before anything may look at them, and NOTHING may look before meet_release() has read ctr[5]: every
// host path that hands results out -- getters, the output hook, the sink, exchange begin / publish -- goes through it
// (mcx_engine.hip: meet_release callers).  A new consumer of those buffers must do the same.
// Launches of main-loop steps only have no meetings.
#pragma once
#include "mcx_device.hpp"

// How an abandoned tuner meeting is handled inside the step loop (an A/B switch for tools/meet_ab.sh; 2 ships):
//   0  meetings wait without a bound (round 2; not safe -- measurement only)
//   2  bounded wait; after an abandoned meeting the owner goes on to the end of its phase with a zero count
#ifndef MCX_MEET_VARIANT
#define MCX_MEET_VARIANT 2
#endif

namespace mcx {

constexpr int PBLOCK = 1024;          // 16 wavefronts: 4 per SIMD of one CU
constexpr int PWAVES = PBLOCK / 64;
constexpr int POWN_MAX = 8;           // owner wavefronts per workgroup: measured 1.2-3x faster than the fused kernels up to 6, equal at 7-8 (tools/persist_sweep.py)
constexpr int PKMAX = 32;             // most steps per phase (LDS double buffers hold 2 phases)
constexpr int PTRASH = 16;           // floats of RunArgs::trash per thread of the grid (a float4 per block of the lane)
constexpr int PTRACE_WG = 4, PTRACE_PH = 128;
constexpr int PDEAL = 24;             // most generator items one wavefront is dealt per phase (RunArgs::deal)
constexpr uint32_t PDEAL_END = 0xffffffffu;
constexpr int PLEAVES = 16;           // words a tuner meeting's arrivals are spread over (same-address atomics cost ~10 ns each)
constexpr int PEVENTS = 64;           // tuner events (steps 51, 101, ... and the end of the burn-in) one launch can hold

struct RunArgs {
  float *x, *ly, *mu, *psum2, *sig;
  const float *x0;          // the run's first launch: initial state to start from (its likelihood is evaluated here,
                            // src/mcpar.cc:47-53) instead of (x, ly); null otherwise
  const float *T0;          // the factor as installed, when T has not been reset for this run yet; null otherwise
  int fresh;                // the run's first launch: counters start from zero (acc_cnt, ctr, ntrace are overwritten)
  uint32_t *acc_cnt;
  float *T;                 // [d][d] Cholesky factor, diagonal here; rescaled in place when the launch ends
  float *samp_x, *samp_ly;  // sample store of the run (row 0 = main step 0 / kept step 0), or null
  float *trash;             // PTRASH floats per thread of the grid: where lanes that own no parameters dump their stores
  int samp_stride;
  const float *lik;
  int ncomp;
  int n, d;
  uint32_t g0, t0, seed;    // t0 = RNG step index of this launch's first step
  int nburn;                // burn-in steps of this launch: the whole burn-in (steps 0 .. nburn-1) or none
  int nmain, isamp0;        // then nmain main-loop steps isamp0 .. isamp0+nmain-1
  int init_moments;         // the main part opens the main loop: mu = 0, psum2 = FPEPS (src/mcpar.cc:99-104)
  const float *winv;        // winv[i] = 1/(i+1)
  float *musig_own;         // this shard's musigall slot
  int snap_after;           // main step (relative to isamp0) after which the slot is snapshot, or -1
  int final_publish;        // 1: the launch ends the run of a single shard: slot and sig from the final moments
  float armin, armax, dfac, ifac;
  unsigned long long *ctr;  // [1] tuner naccept [2] tuner ntrial [3] burn-in accepts [4] main-loop accepts [5] a meeting was abandoned
  unsigned long long *bar;  // PLEAVES words per tuner event of this launch (<= PEVENTS events), zero at launch
  float *trace;
  int *ntrace;
  int nown;                 // owner wavefronts in the whole grid = ceil(n * LPC / 64)
  int own;                  // owner wavefronts per workgroup (1..POWN_MAX)
  int ksteps;               // steps per phase (mcxk_persist_ksteps: what the LDS double buffers hold, <= PKMAX)
  // Who generates what (mcxk_persist_deal): [3][PWAVES][PDEAL] item codes, PDEAL_END-terminated, one list per wavefront
  // for [0] the fill before the first phase (every wavefront but the owners), [1] phases whose concurrent phase is
  // burn-in, [2] main-loop (the recorders are at work too).  An item = kind << 14 | step or step pair << 4 | (owner, block):
  // kind 0 the normals of two consecutive steps, 3 of one step, 1 the acceptance logs of one owner, 2 the phase's 1/pwgt values.
  const uint32_t *deal;
  // builds with -DMCX_PERSIST_TRACE only (tools/persist_trace.py): [PTRACE_WG workgroups][PWAVES][PTRACE_PH phases][2] shader
  // clock after the wavefront's own work of the phase / after the barrier that ends it; null otherwise
  unsigned long long *trace_clk;
  unsigned long long meet_timeout;  // 100 MHz ticks a tuner meeting may take before the launch is abandoned
  int meet_expect_extra;    // debug (MCX_OPT_DEBUG_MEET): workgroups the meetings wait for beyond the grid's own
  // The launch that ends a run tells the host itself: its last workgroup copies ctr[0..6] to `report` (pinned, mapped host
  // memory) and then stores `report_serial` to report[7], which the host spins on -- no copy kernel behind the launch, no
  // interrupt (host wall - kernel time of a lone kernel: 16 us with the copy + hipStreamSynchronize, 4-6 this way,
  // tools/sync_latency_probe.hip).  `report_done` counts the workgroups that are through and is left zero.  An abandoned
  // launch reports like any other (ctr[5] says what happened).  report null: nothing of the kind.
  unsigned long long *report;
  unsigned *report_done;
  unsigned long long report_serial;
};

static_assert(PLEAVES == 16, "a meeting's leaves are polled by the 16 lanes of one DPP row");
constexpr unsigned long long MEET_ABORT_BIT = 1ull << 63;  // in a meeting word: a waiting owner gave up
constexpr unsigned long long MEET_ABORTED = ~0ull - 1ull;  // in lds_out / as owners_meet's result: abandon the launch

// all owners of the grid meet; returns the sum of `mine` over the workgroups, or MEET_ABORTED when the meeting
// was abandoned (some workgroup did not arrive within `timeout` wall-clock ticks).  The owner waves of a workgroup
// first add up in LDS; the last of them speaks for the workgroup: ONE atomic, on leaf blockIdx % PLEAVES of the
// event's PLEAVES words (256 arrivals on one word are 256 same-address atomics in a row, ~2.5 us of a 4 us meeting:
// round 4's trace), then lanes 0 .. PLEAVES - 1 of its wavefront poll one leaf each until the arrivals add up to the grid.
// leaf = abort << 63 | arrivals << 40 | sum (a meeting's sum is < 2^32: <= 50 steps of <= 2^17 chains).
// Called by whole wavefronts (wave-uniform arguments).
__device__ __forceinline__ unsigned long long owners_meet(unsigned long long *leaves, unsigned mine, int own, int nwg,
                                                         unsigned *lds_sum, unsigned *lds_cnt, unsigned long long *lds_out,
                                                         unsigned long long timeout)
{
  const unsigned lane = threadIdx.x & 63u;
  unsigned arrived = 0;
  if (lane == 0) {
    atomicAdd(lds_sum, mine);
    __threadfence_block();
    arrived = atomicAdd(lds_cnt, 1u) + 1u;
  }
  const bool speaker = __builtin_amdgcn_readfirstlane(arrived) == (unsigned)own;  // the last owner of this workgroup
  unsigned long long total = 0;
  if (speaker) {
    if (lane == 0) {
      const unsigned s = atomicExch(lds_sum, 0u);
      atomicExch(lds_cnt, 0u);
      // the leaves are the only thing the workgroups share: relaxed device-scope atomics, no cache maintenance
      // (the returned value is consumed: an atomic whose result is never read would stay "pending" for the
      // compiler's s_waitcnt bookkeeping and put vmcnt waits -- on the previous step's stores -- into the step loop)
      const unsigned long long add = (1ull << 40) | (unsigned long long)s;
      const unsigned long long old = __hip_atomic_fetch_add(leaves + (blockIdx.x % PLEAVES), add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(old));
    }
    // bounded wait: the constant-rate wall clock, looked at once per 64 polls (reading it is a scalar memory
    // operation of its own: once per poll it stretched every meeting of a healthy run)
    unsigned long long t_start = 0;
    unsigned polls = 0, sum = 0;
    bool abort = false;
    for (;;) {
      const unsigned long long v = lane < (unsigned)PLEAVES ? __hip_atomic_load(leaves + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      // arrivals over the 16 leaves = the 16 lanes of DPP row 0: four rotate-and-add steps, no LDS crossbar in the poll
      unsigned arr = (unsigned)((v & ~MEET_ABORT_BIT) >> 40);
      sum = (unsigned)v;
      arr += (unsigned)__builtin_amdgcn_update_dpp(0, (int)arr, 0x128, 0xF, 0xF, true);  // row_ror:8
      arr += (unsigned)__builtin_amdgcn_update_dpp(0, (int)arr, 0x124, 0xF, 0xF, true);  // row_ror:4
      arr += (unsigned)__builtin_amdgcn_update_dpp(0, (int)arr, 0x122, 0xF, 0xF, true);  // row_ror:2
      arr += (unsigned)__builtin_amdgcn_update_dpp(0, (int)arr, 0x121, 0xF, 0xF, true);  // row_ror:1
      arr = __builtin_amdgcn_readfirstlane(arr);
      abort = __ballot((v & MEET_ABORT_BIT) != 0ull) != 0ull;
      if ((int)arr >= nwg || abort) break;
#if MCX_MEET_VARIANT != 0
      if ((++polls & 63u) == 0u) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t_start == 0) t_start = now;
        else if (now - t_start > timeout) {  // someone never arrived: tell everybody
          if (lane == 0) {
            const unsigned long long o2 = __hip_atomic_fetch_or(leaves, MEET_ABORT_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(o2));
          }
          abort = true;
          break;
        }
      }
#endif
      __builtin_amdgcn_s_sleep(2);
    }
    sum += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sum, 0x128, 0xF, 0xF, true);
    sum += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sum, 0x124, 0xF, 0xF, true);
    sum += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sum, 0x122, 0xF, 0xF, true);
    sum += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sum, 0x121, 0xF, 0xF, true);
    total = abort ? MEET_ABORTED : (unsigned long long)__builtin_amdgcn_readfirstlane(sum);
    if (lane == 0) __hip_atomic_store(lds_out, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    if (lane == 0) {
      for (;;) {
        total = __hip_atomic_load(lds_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (total != ~0ull) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    // broadcast lane 0's value to the wave
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)total), hi = __builtin_amdgcn_readfirstlane((unsigned)(total >> 32));
    total = ((unsigned long long)hi << 32) | lo;
  }
  return total;
}

// LPC2 lanes per chain, each holding BPL consecutive 4-parameter blocks (LPC = LPC2 * BPL blocks per chain).  BPL = 2
// (d % 8 == 0) halves the owner wavefronts: the per-chain part of a step (acceptance test, selects, counters, ballot) is
// paid once per two blocks, the first stage of the butterfly over the block index is an in-lane add --
// the same pair, the same sum -- and Philox counters go by block index whichever lane holds the block: same bits.
template <int LPC2, int BPL, int LIK, bool REC>
__device__ __forceinline__ void run_small_body(const RunArgs &a);

template <int LPC2, int BPL, int LIK, bool REC>
__global__ __launch_bounds__(PBLOCK) void k_run_small(const RunArgs a)
{
  run_small_body<LPC2, BPL, LIK, REC>(a);
}

// (the body apart from the kernel: mcx_user.hip compiles it at run time around a user's likelihood source, LIK_USER)
template <int LPC2, int BPL, int LIK, bool REC>
__device__ __forceinline__ void run_small_body(const RunArgs &a)
{
  static_assert(LIK == LIK_ROSEN1 || LIK == LIK_GAUSS || LIK == LIK_MIX || LIK == LIK_USER, "hot-path likelihoods (or a user's source in block form)");
  static_assert(BPL == 1 || BPL == 2 || BPL == 4, "one, two or four blocks per lane");
#ifdef MCX_PERSIST_TRACE
  const unsigned long long trace_t_entry = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ __attribute__((aligned(16))) float lds_means[LIK == LIK_MIX ? 8 * MAXD_LDS : 4];
  __shared__ float lds_logw[8];
  __shared__ unsigned lds_sum, lds_cnt, lds_abort;
  __shared__ unsigned long long lds_out[PEVENTS];  // one result word per tuner event (never reused within a launch)
  __shared__ float wbuf[4 * PKMAX];  // 1/pwgt of a phase's main-loop steps, by phase % 4 (written 1 ahead, read 1 behind)
  if (LIK == LIK_MIX) {
    const int kd = a.ncomp * a.d;
    for (int i = threadIdx.x; i < kd; i += PBLOCK) lds_means[i] = a.lik[i];
    if (threadIdx.x < (unsigned)a.ncomp) lds_logw[threadIdx.x] = a.lik[kd + threadIdx.x];
  }
  if (threadIdx.x == 0) { lds_sum = 0; lds_cnt = 0; lds_abort = 0; }
  if (threadIdx.x < PEVENTS) lds_out[threadIdx.x] = ~0ull;
  constexpr bool NEGSUM = LIK == LIK_ROSEN1 || LIK == LIK_GAUSS;  // (see loglike)
  const int OWN = a.own, K = a.ksteps;  // steps per phase: a multiple of the generator count
  const int T = a.nburn + a.nmain;
  const int nphase = (T + K - 1) / K;
  // (the wavefront's index as a SCALAR: what it decides -- owner, recorder, generator, working -- is then uniform control
  // flow to the compiler too: loop counters in SGPRs and s_cbranch instead of exec masks and VGPR counters in the owner's
  // step loop, which a single wavefront issues at one instruction per 5.7 cycles: tools/ubench.hip)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
  const bool owner = wv < OWN, recorder = REC && !owner && wv < 2 * OWN;
  // LDS double buffers, by phase parity: zbuf / xbuf [2][K][OWN][BPL][64] float4, ubuf / lbuf [2][K][OWN][64/LPC2] float
  constexpr int CPW = 64 / LPC2;  // chains per owner wavefront
  const int OB = OWN * BPL;       // (owner, block of the lane) pairs of the workgroup
  const size_t nz = (size_t)2 * K * OB * 64, nu = (size_t)2 * K * OWN * CPW;
  float4 *zbuf = reinterpret_cast<float4 *>(lds_raw);
  float4 *xbuf = zbuf + nz;  // (REC only)
  float *ubuf = reinterpret_cast<float *>(zbuf + (REC ? 2 : 1) * nz);
  float *lbuf = ubuf + nu;   // (REC only)
  const int d = a.d;

  // ---- generator side -------------------------------------------------------------------------------------
  // The work of one phase is a list of items dealt round-robin to the `nfill` waves that fill in this
  // iteration (a counter in LDS was measured slower: ~40 same-address atomics per phase serialise).
  const int fq = lane % LPC2, fcl = lane / LPC2;  // this lane's place in its chain and its chain within an owner wavefront
  const __attribute__((address_space(4))) uint32_t *deal_tab = (const __attribute__((address_space(4))) uint32_t *)a.deal;
  auto fill = [&](int phase, int table) {
    const int buf = phase & 1, tau0 = phase * K;
    const int ns = T - tau0 < K ? T - tau0 : K;  // steps [tau0, tau0 + ns)
    const __attribute__((address_space(4))) uint32_t *mylist = deal_tab + ((size_t)table * PWAVES + wv) * PDEAL;
    for (int j = 0; j < PDEAL; ++j) {
      const uint32_t item = mylist[j];  // (wave-uniform: a scalar load)
      if (item == PDEAL_END) break;
      const int kind = (int)(item >> 14);
      if (kind == 3) {
        // normals of ONE step of one (owner, block) set: what evens the wavefronts' shares out when the pairs do not
        const int g0s = (int)((item >> 4) & 0x3ffu), ob = (int)(item & 15u);
        if (g0s >= ns) continue;  // (a last, shorter phase)
        const int o = ob / BPL, b = ob % BPL;
        const int chain = ((int)blockIdx.x * OWN + o) * CPW + fcl;
        const int qb = fq * BPL + b;
        if (chain < a.n && 4 * qb < d) {
          f32x2 ze, zo;
          normal4_packed(philox4x32_10(a.t0 + (uint32_t)(tau0 + g0s), a.g0 + (uint32_t)chain, (uint32_t)qb, 0u, a.seed, ST_LOCAL), ze, zo);
          zbuf[((size_t)(buf * K + g0s) * OB + ob) * 64 + lane] = make_float4(ze.x, ze.y, zo.x, zo.y);
        }
      } else if (kind == 0) {
        // normals: one item = TWO consecutive steps of one block-per-lane set of one owner (two independent Philox /
        // Box-Muller chains per lane: the lone instruction streams of 3-4 waves do not fill a SIMD otherwise)
        const int gp = (int)((item >> 4) & 0x3ffu), ob = (int)(item & 15u);
        const int g0s = 2 * gp;
        if (g0s >= ns) continue;  // (a last, shorter phase)
        const int o = ob / BPL, b = ob % BPL;
        const int chain = ((int)blockIdx.x * OWN + o) * CPW + fcl;
        const int qb = fq * BPL + b;  // the block's index within the chain
        if (chain < a.n && 4 * qb < d) {
          const uint32_t t = a.t0 + (uint32_t)(tau0 + g0s), gch = a.g0 + (uint32_t)chain;
          f32x2 ze, zo, ye, yo;
          normal4_packed(philox4x32_10(t, gch, (uint32_t)qb, 0u, a.seed, ST_LOCAL), ze, zo);
          normal4_packed(philox4x32_10(t + 1u, gch, (uint32_t)qb, 0u, a.seed, ST_LOCAL), ye, yo);
          float4 *dst = zbuf + ((size_t)(buf * K + g0s) * OB + ob) * 64 + lane;
          dst[0] = make_float4(ze.x, ze.y, zo.x, zo.y);  // (even pair, odd pair)
          // Unconditionally: K is even, so the slot of step g0s + 1 exists even when a last, shorter phase does not use
          // it -- and a store under `if (g0s + 1 < ns)` made the compiler sink the whole second chain behind the first
          // (no interleaving: the very thing two steps per item are for).
          dst[(size_t)OB * 64] = make_float4(ye.x, ye.y, yo.x, yo.y);
        }
      } else if (kind == 1) {
        // the logs of the phase's acceptance draws of one owner (one Philox block of the ACCEPT stream serves 4 steps)
        const int oo = (int)(item & 15u);
        const uint32_t tf = a.t0 + (uint32_t)tau0, tl = tf + (uint32_t)ns - 1u;
        const uint32_t bf = tf >> 2, bl = tl >> 2;
        const int c = lane % CPW;
        const int chain = ((int)blockIdx.x * OWN + oo) * CPW + c;
        if (chain < a.n)
          for (uint32_t b = bf + (uint32_t)(lane / CPW); b <= bl; b += (uint32_t)LPC2) {
            const u32x4 aw = philox4x32_10(b, a.g0 + (uint32_t)chain, 0u, 0u, a.seed, ST_ACCEPT);
            const f32x2 l01 = accept_lu_x2(aw.x, aw.y), l23 = accept_lu_x2(aw.z, aw.w);
            const float l[4] = {l01.x, l01.y, l23.x, l23.y};
#pragma unroll
            for (uint32_t wi = 0; wi < 4u; ++wi) {
              const uint32_t t = (b << 2) + wi;
              if (t >= tf && t <= tl) ubuf[((size_t)(buf * K + (int)(t - tf)) * OWN + oo) * CPW + c] = l[wi];
            }
          }
      } else if (lane < ns) {  // 1/pwgt (src/mcpar.cc:186-187), from the host-built table
        const int im = tau0 + lane - a.nburn;
        wbuf[(phase & 3) * PKMAX + lane] = (im >= 0 && im < a.nmain) ? a.winv[a.isamp0 + im] : 1.0f;
      }
    }
  };

  // ---- this wave's chains (the lane layout of k_fused_fast / k_fused_fastb); recorder o + OWN mirrors owner o ---
  const int slot_o = owner ? wv : (recorder ? wv - OWN : 0);
  const size_t gid = ((size_t)blockIdx.x * OWN + slot_o) * 64 + lane;
  const size_t chain = gid / LPC2;
  const int q = (int)(gid % LPC2);
  const int k0 = 4 * BPL * q;  // first parameter of this lane; block b holds k0 + 4 b .. k0 + 4 b + 3 (d % (4 BPL) == 0)
  const bool mine = (owner || recorder) && chain < (size_t)a.n;
  const bool live = mine && k0 < d;
  const bool working = (owner || recorder) && (int)blockIdx.x * OWN + slot_o < a.nown;
  const size_t off = chain * (size_t)d + k0;
  f32x2 xe[BPL], xo[BPL], me[BPL], mo[BPL], se[BPL], so[BPL], te[BPL], to[BPL], gme[BPL], gmo[BPL];
  float gs[BPL][4];
  float ly = __builtin_inff();
#pragma unroll
  for (int b = 0; b < BPL; ++b) {
    xe[b] = xo[b] = me[b] = mo[b] = se[b] = so[b] = te[b] = to[b] = gme[b] = gmo[b] = f32x2{0, 0};
    gs[b][0] = gs[b][1] = gs[b][2] = gs[b][3] = 0.0f;
    const int kb = k0 + 4 * b;
    if (live && owner) {
      const float4 f = *reinterpret_cast<const float4 *>((a.x0 ? a.x0 : a.x) + off + 4 * b);
      xe[b] = f32x2{f.x, f.z}; xo[b] = f32x2{f.y, f.w};
      const float *Tsrc = a.T0 ? a.T0 : a.T;
      te[b] = f32x2{Tsrc[(kb + 0) * d + kb + 0], Tsrc[(kb + 2) * d + kb + 2]};
      to[b] = f32x2{Tsrc[(kb + 1) * d + kb + 1], Tsrc[(kb + 3) * d + kb + 3]};
      if (LIK == LIK_GAUSS) {
        gme[b] = f32x2{a.lik[kb + 0], a.lik[kb + 2]}; gmo[b] = f32x2{a.lik[kb + 1], a.lik[kb + 3]};
        gs[b][0] = a.lik[d + kb + 0]; gs[b][1] = a.lik[d + kb + 1]; gs[b][2] = a.lik[d + kb + 2]; gs[b][3] = a.lik[d + kb + 3];
      }
    }
    if (live && (REC ? recorder : owner) && a.nmain > 0 && !a.init_moments) {
      const float4 m = *reinterpret_cast<const float4 *>(a.mu + off + 4 * b);
      const float4 p = *reinterpret_cast<const float4 *>(a.psum2 + off + 4 * b);
      me[b] = f32x2{m.x, m.z}; mo[b] = f32x2{m.y, m.w};
      se[b] = f32x2{p.x, p.z}; so[b] = f32x2{p.y, p.w};
    }
  }
  // owner lanes that hold no chain never accept: log u < ly' - (+inf) is false for every ly'
  if (mine && owner && !a.x0) ly = a.ly[chain];
  // Every load of the chain state is awaited here, once, on every path (the compiler's s_waitcnt bookkeeping
  // is path-insensitive): inside the step loops the only vector-memory operations are stores, and no
  // s_waitcnt vmcnt may end up there -- it would wait for the previous step's stores.
#pragma unroll
  for (int b = 0; b < BPL; ++b)
    asm volatile("" ::"v"(xe[b]), "v"(xo[b]), "v"(te[b]), "v"(to[b]), "v"(me[b]), "v"(mo[b]), "v"(se[b]), "v"(so[b]), "v"(gme[b]),
                 "v"(gmo[b]), "v"(gs[b][0]), "v"(gs[b][1]), "v"(gs[b][2]), "v"(gs[b][3]));
  asm volatile("" ::"v"(ly));
  if (owner) __builtin_amdgcn_s_setprio(3);  // the owners' dependent instruction stream goes first on its SIMD
  else if (recorder) __builtin_amdgcn_s_setprio(1);
  uint32_t cnt = 0, cnt_mark = 0;  // accepted proposals of this lane's chain; its value at the last tuner event / end of the burn-in
  uint32_t btail = 0;              // its accepted burn-in proposals after the tuner's last decision (added up in the epilogue)
  // accepted proposals of the wave's chains since `mark`: the per-chain counters of the chains' first lanes, added up
  // when somebody asks -- at a tuner event, at the end of the launch -- not ballot by ballot in the step loop
  // (4 of its ~40 instructions)
  auto wave_sum_chains = [&](uint32_t per_chain) -> uint32_t {  // over the chains of the wave: their first lanes' values
    uint32_t v = (mine && q == 0) ? per_chain : 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
  };
  auto wave_accepts = [&](uint32_t mark) -> uint32_t { return wave_sum_chains(cnt - mark); };

  // the first log2(BPL) stages of the butterfly over the block index, inside the lane; the lane group does the rest
  auto blocks_sum = [&](const float p[BPL]) -> float {
    if (BPL == 1) return group_sum<LPC2>(p[0]);
    if (BPL == 2) return group_sum<LPC2>(p[0] + p[BPL > 1 ? 1 : 0]);
    return group_sum<LPC2>((p[0] + p[BPL > 1 ? 1 : 0]) + (p[BPL == 4 ? 2 : 0] + p[BPL == 4 ? 3 : 0]));
  };

  // likelihood of the proposal (pe, po), same arithmetic as k_fused_fast
  auto loglike = [&](const f32x2 pe[BPL], const f32x2 po[BPL]) -> float {
#ifdef MCX_USER_LIK
    if (LIK == LIK_USER) {  // a user's source, block form: per-block partials, the engine's butterfly, the user's finish
      float acc[BPL];
#pragma unroll
      for (int b = 0; b < BPL; ++b) {
        const float xb[4] = {pe[b].x, po[b].x, pe[b].y, po[b].y};
        acc[b] = live ? ::mcx_user_block(xb, 4, k0 + 4 * b, d, a.lik) : 0.0f;
      }
      return ::mcx_user_finish(blocks_sum(acc), d, a.lik);
    }
#endif
    if (LIK == LIK_MIX) {
      const int Kc = a.ncomp;
      float e[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        e[c] = 0.0f;
        if (c < Kc) {
          float s2[BPL];
#pragma unroll
          for (int b = 0; b < BPL; ++b) {
            s2[b] = 0.0f;
            if (live) {
              const float4 m = *reinterpret_cast<const float4 *>(&lds_means[c * d + k0 + 4 * b]);
              const f32x2 ae = pe[b] - f32x2{m.x, m.z}, ao = po[b] - f32x2{m.y, m.w};
              s2[b] = __builtin_fmaf(ae.x, ae.x, 0.0f);
              s2[b] = __builtin_fmaf(ao.x, ao.x, s2[b]);
              s2[b] = __builtin_fmaf(ae.y, ae.y, s2[b]);
              s2[b] = __builtin_fmaf(ao.y, ao.y, s2[b]);
            }
          }
          e[c] = __builtin_fmaf(-0.5f, blocks_sum(s2), lds_logw[c]);
        }
      }
      float emax = e[0];
#pragma unroll
      for (int c = 1; c < 8; ++c)
        if (c < Kc) emax = e[c] > emax ? e[c] : emax;
      float ssum = 0.0f;
#pragma unroll
      for (int c = 0; c < 8; c += 2) {
        if (c < Kc) {
          const f32x2 ex = expf_v2x2(f32x2{e[c] - emax, e[c + 1] - emax});
          ssum = ssum + ex.x;
          if (c + 1 < Kc) ssum = ssum + ex.y;
        }
      }
      return emax + logf_v1(ssum);
    }
    float acc[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
      acc[b] = 0.0f;
      if (LIK == LIK_ROSEN1) {
        const f32x2 t1 = splat2(1.0f) - pe[b];
        const f32x2 t2 = fma2(-pe[b], pe[b], po[b]);
        const f32x2 term = fma2(splat2(100.0f) * t2, t2, t1 * t1);
        // (the select stays: a lane without parameters is never dealt normals, its slots of zbuf hold whatever LDS held, and
        // 0 x inf is a NaN.  Parking such lanes at x = 1, T = 0 -- term +0 by itself -- needs the slots zeroed once per launch,
        // which costs 6 us of a 0.36 ms job: more than the select)
        if (live) acc[b] = term.x + term.y;
      } else if (LIK == LIK_GAUSS) {
        const f32x2 ae = pe[b] - gme[b], ao = po[b] - gmo[b];
        const f32x2 he = (splat2(0.5f) * ae) * ae, ho = (splat2(0.5f) * ao) * ao;
        if (live) {
          acc[b] = __builtin_fmaf(he.x, gs[b][0], 0.0f);
          acc[b] = __builtin_fmaf(ho.x, gs[b][1], acc[b]);
          acc[b] = __builtin_fmaf(he.y, gs[b][2], acc[b]);
          acc[b] = __builtin_fmaf(ho.y, gs[b][3], acc[b]);
        }
      }
    }
    // Rosenbrock1 / Gaussian (NEGSUM): the sum S itself, L = 0 - S.  The owners keep S_cur = -ly instead of ly: the test
    // log u < ly' - ly is log u < S_cur - S' -- the same difference, rounded once either way -- and the negation leaves
    // the step loop, where every instruction costs this lone wavefront 5.7 cycles: ly = 0 - S_cur where it is wanted
    if (NEGSUM) return blocks_sum(acc);
    return 0.0f - blocks_sum(acc);
  };

  __syncthreads();  // the mixture's means and log-weights are staged
  if (NEGSUM && owner) ly = 0.0f - ly;  // (a state handed over from an earlier launch: S_cur = -ly; -inf where no chain is)
  if (owner && a.x0) {  // L(pinit) (src/mcpar.cc:53); every lane of a chain gets the chain's value
    const float l0 = loglike(xe, xo);
    ly = mine ? l0 : (NEGSUM ? -__builtin_inff() : __builtin_inff());
  }
  // tuner state (src/mcpar.cc:77-96), identical in every owner wave
  unsigned long long tun_na = a.fresh ? 0ull : a.ctr[1], tun_nt = a.fresh ? 0ull : a.ctr[2], burn_acc = 0;
  int irate = 50, seg_start = 0, nevent = 0, ntrace_local = 0;
  const int nwg = (int)gridDim.x + a.meet_expect_extra;
  bool aborted = false;
  const int own_here = a.nown - (int)blockIdx.x * OWN < OWN ? a.nown - (int)blockIdx.x * OWN : OWN;
  int next_event = a.nburn > 0 ? (51 < a.nburn ? 51 : a.nburn - 1) : -1;  // burn-in step of the next tuner event

  // recorder: sample store through per-lane pointers that advance by one row per kept step; lanes that own
  // nothing store to a trash slot with stride 0, so the stores need no exec-mask regions
  const size_t rowx = (size_t)a.n * d, rowl = (size_t)a.n;
  const bool emit = a.samp_x != nullptr;
  const int sstride = a.samp_stride > 1 ? a.samp_stride : 1;
  float *sxv = nullptr, *slv = nullptr;
  size_t sxs = 0, sls = 0;
  int kmod = 0;  // isamp % samp_stride of the next main-loop step
  if (emit) {
    kmod = a.isamp0 % sstride;
    const size_t row0 = (size_t)((a.isamp0 + sstride - 1) / sstride);  // row of the first kept step >= isamp0
    float *dump = a.trash + PTRASH * ((size_t)blockIdx.x * PBLOCK + threadIdx.x);
    const bool rec = REC ? recorder : owner;
    sxv = (live && rec) ? a.samp_x + row0 * rowx + off : dump;
    slv = (mine && rec) ? a.samp_ly + row0 * rowl + chain : dump;
    sxs = (live && rec) ? rowx : 0;
    sls = (mine && rec) ? rowl : 0;
  }
  // Welford (src/mcpar.cc:184-209), the exchange snapshot (:202-208) and the sample emission (:177-182) of one main-loop
  // step, from the post-step state (ce, co, lyv): the recorder's work, or the owner's when there are no recorders
  auto record = [&](const f32x2 ce[BPL], const f32x2 co[BPL], float lyv, float w, bool snap) {
    const f32x2 w2 = splat2(w);  // 1/pwgt, src/mcpar.cc:186-187
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
      const f32x2 de = ce[b] - me[b], dO = co[b] - mo[b];  // src/mcpar.cc:199-202
      me[b] = fma2(de, w2, me[b]);
      mo[b] = fma2(dO, w2, mo[b]);
      se[b] = fma2(de, ce[b] - me[b], se[b]);
      so[b] = fma2(dO, co[b] - mo[b], so[b]);
    }
    if (snap && live) {  // snapshot for the next exchange (src/mcpar.cc:202-208)
#pragma unroll
      for (int b = 0; b < BPL; ++b) {
        const f32x2 ve = se[b] * w2, vo = so[b] * w2;
        float4 *slot = reinterpret_cast<float4 *>(a.musig_own + 2 * (off + 4 * b));
        slot[0] = make_float4(me[b].x, ve.x, mo[b].x, vo.x);
        slot[1] = make_float4(me[b].y, ve.y, mo[b].y, vo.y);
      }
    }
    if (emit) {  // src/mcpar.cc:177-182
      if (kmod == 0) {
#pragma unroll
        for (int b = 0; b < BPL; ++b) *reinterpret_cast<float4 *>(sxv + 4 * b) = make_float4(ce[b].x, co[b].x, ce[b].y, co[b].y);
        *slv = lyv;  // every lane of the chain stores the same value
        sxv += sxs;
        slv += sls;
      }
      kmod = kmod + 1 == sstride ? 0 : kmod + 1;
    }
  };

  __syncthreads();
#ifdef MCX_PERSIST_TRACE
  const unsigned long long trace_t_ready = __builtin_amdgcn_s_memtime();  // state loaded, L(pinit) evaluated
#endif
  if (!owner) fill(0, 0);  // (nobody has anything to record yet: the recorders fill too)
  __syncthreads();
#ifdef MCX_PERSIST_TRACE
  // (the last traced phase slot of wavefront 0 carries the launch's own milestones: entry / ready / first fill done / end)
  const unsigned long long trace_t_filled = __builtin_amdgcn_s_memtime();
#endif
  // iteration p: owners run phase p, recorders digest phase p - 1, everybody else fills phase p + 1
  for (int p = 0; p < nphase + (REC ? 1 : 0); ++p) {
    const int buf = p & 1;
    // (After an abandoned tuner meeting nobody leaves early: the owners go on with counts of zero, the other wavefronts
    // never learn of it, the launch runs to its end and writes nothing back -- an abandoned launch is rare and its time
    // is the host's to lose.  A look at the flag at the top of every phase cost BASELINE's C2 6 % of its kernel time:
    // round 4's A/B, 0.2708 -> 0.2541 ms.)
    if (owner) {
      if (working && p < nphase) {
        const int tau0 = p * K;
        const int ns = T - tau0 < K ? T - tau0 : K;
        const int nb = a.nburn - tau0 < 0 ? 0 : (a.nburn - tau0 < ns ? a.nburn - tau0 : ns);  // burn-in steps of this phase
        const float4 *zp = zbuf + ((size_t)(buf * K) * OB + wv * BPL) * 64 + lane;
        const float *up = ubuf + ((size_t)(buf * K) * OWN + wv) * CPW + lane / LPC2;
        float4 *xq = xbuf + ((size_t)(buf * K) * OB + wv * BPL) * 64 + lane;
        float *lq = lbuf + ((size_t)(buf * K) * OWN + wv) * CPW + lane / LPC2;
        float4 zn[BPL];
#pragma unroll
        for (int b = 0; b < BPL; ++b) zn[b] = zp[b * 64];
        float lun = *up;
        // proposal, likelihood, acceptance (src/mcpar.cc:302-312, 62-75); returns the chains of the wave that accepted.
        // The next step's numbers are fetched first: they are on their way while this step computes.
        auto core = [&](const float4 (&z)[BPL], const float lu) {
          f32x2 pe[BPL], po[BPL];
#pragma unroll
          for (int b = 0; b < BPL; ++b) {
            pe[b] = fma2(te[b], f32x2{z[b].x, z[b].y}, xe[b]);
            po[b] = fma2(to[b], f32x2{z[b].z, z[b].w}, xo[b]);
          }
          const float lyt = loglike(pe, po);
          const bool take = NEGSUM ? lu < ly - lyt : accept_local(lyt, ly, lu);  // (NEGSUM: ly is S_cur, lyt is S')
#pragma unroll
          for (int b = 0; b < BPL; ++b) {
            xe[b] = take ? pe[b] : xe[b];
            xo[b] = take ? po[b] : xo[b];
          }
          ly = take ? lyt : ly;
          {  // cnt += take: every lane of a chain counts its chain's accepted proposals -- ONE instruction, an add whose carry-in
             // is the comparison's mask (the compiler makes a select and an add of it)
            const unsigned long long tm = __builtin_amdgcn_ballot_w64(take);
            unsigned long long carry_out;
            asm("v_addc_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(cnt), "=s"(carry_out) : "s"(tm));
          }
        };
        auto fetch_next = [&](float4 (&z)[BPL], float &lu) {
          zp += (size_t)OB * 64;
          up += (size_t)OWN * CPW;
#pragma unroll
          for (int b = 0; b < BPL; ++b) z[b] = zp[b * 64];
          lu = *up;
        };
        auto metropolis = [&](int s) {
          float4 z[BPL];
#pragma unroll
          for (int b = 0; b < BPL; ++b) z[b] = zn[b];
          const float lu = lun;
          if (s + 1 < ns) fetch_next(zn, lun);
          core(z, lu);
        };
        // Steps s and s + 1 (s + 1 < ns): the second step's numbers go to registers of their own, the third's back to (zn, lun)
        // -- nothing is moved from "next" to "this" between two steps (3 of a step's 30 instructions, and this wavefront
        // issues one per 5.7 cycles whatever it is).  `between` is what follows each step (the main loop's hand-over).
        auto two_steps = [&](int s, auto &&between) {
          float4 z1[BPL];
          float lu1;
          fetch_next(z1, lu1);
          core(zn, lun);
          between();
          if (s + 2 < ns) fetch_next(zn, lun);
          core(z1, lu1);
          between();
        };
        // Two loops, not one with a branch: the main-loop steps write to LDS, and a wait shared by both kinds of
        // step would have to cover those writes (LDS operations retire in order) on every step.
        int s = 0;
        while (s < nb) {  // ---- burn-in steps (src/mcpar.cc:58-75), up to and including the next tuner event's step
          const int ev = next_event - tau0;  // (>= nb: the event lies in a later phase)
          const int send = ev < nb ? ev + 1 : nb;
          for (; s + 1 < send; s += 2) two_steps(s, [] {});  // the step loop proper: nothing in it but the steps
          for (; s < send; ++s) metropolis(s);
          if (ev < nb) {  // tuner event (src/mcpar.cc:77-96): the accept count of every chain of the shard
            const int last = next_event;
            const int steps = last - seg_start + 1, check = last > irate ? 1 : 0;
            if (!check) {
              // The event at the burn-in's last step when the tuner decides nothing there (src/mcpar.cc:78 is false):
              // nobody's next step depends on this count -- it only completes the burn-in's accept total -- so
              // nobody waits for it: every wave keeps its share and adds it to the total as the launch ends.
              btail += cnt - cnt_mark;
              cnt_mark = cnt;
              ++nevent;
              seg_start = last + 1;
              next_event = a.nburn - 1;
              continue;  // (s == nb: the burn-in's last step is behind us)
            }
            const uint32_t wacc = wave_accepts(cnt_mark);
            cnt_mark = cnt;
#if MCX_MEET_VARIANT == 0
            const unsigned long long seg = owners_meet(a.bar + (size_t)nevent * PLEAVES, wacc, own_here, nwg, &lds_sum, &lds_cnt, &lds_out[nevent],
                                                       a.meet_timeout);
#else
            // An abandoned meeting (wave-uniform) does not cut the phase short: the wave goes on with a count of zero,
            // skips later meetings, and the flag makes the launch skip its epilogue; what an abandoned launch may have
            // dirtied on the way (sample rows, the slot snapshot, trace[]) is listed at the top of this file.
            unsigned long long seg = 0;
            if (!aborted) {
              seg = owners_meet(a.bar + (size_t)nevent * PLEAVES, wacc, own_here, nwg, &lds_sum, &lds_cnt, &lds_out[nevent], a.meet_timeout);
              if (seg == MEET_ABORTED) {
                if (lane == 0) {
                  __hip_atomic_store(&lds_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  __hip_atomic_store(a.ctr + 5, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                aborted = true;
                seg = 0;
              }
            }
#endif
            ++nevent;
            tun_na += seg;
            tun_nt += (unsigned long long)steps * (unsigned long long)a.n;
            burn_acc += seg;
            {
              const float arate = (float)tun_na / (float)tun_nt;
              float f = 1.0f;
              if (arate < a.armin) { tun_na = tun_nt = 0; f = a.dfac; }
              else if (arate > a.armax) { tun_na = tun_nt = 0; f = a.ifac; }
              if (f != 1.0f) {
#pragma unroll
                for (int b = 0; b < BPL; ++b) { te[b] = te[b] * splat2(f); to[b] = to[b] * splat2(f); }
              }
              if (blockIdx.x == 0 && wv == 0 && lane == 0) {  // lane 0 of the grid holds T[0][0]
                const int kk = (a.fresh ? 0 : *a.ntrace) + ntrace_local;
                if (kk < 256) a.trace[kk] = te[0].x;
              }
              ++ntrace_local;
              irate += 50;
            }
            seg_start = last + 1;
            next_event = irate + 1 < a.nburn ? irate + 1 : a.nburn - 1;
          }
        }
        if (nb > 0 && tau0 + nb == a.nburn) cnt_mark = cnt;  // the burn-in ends here: the main loop's accepts count from now
        xq += (size_t)nb * OB * 64;
        lq += (size_t)nb * OWN * CPW;
        // ---- main-loop steps (src/mcpar.cc:152-209)
        auto hand_over = [&] {  // the state to the recorder (every lane of a chain writes the same ly)
#pragma unroll
          for (int b = 0; b < BPL; ++b) xq[b * 64] = make_float4(xe[b].x, xe[b].y, xo[b].x, xo[b].y);
          *lq = ly;
          xq += (size_t)OB * 64;
          lq += (size_t)OWN * CPW;
        };
        if (REC)
          for (; s + 1 < ns; s += 2) two_steps(s, hand_over);
        for (; s < ns; ++s) {
          metropolis(s);
          if (REC) {
            hand_over();
          } else {  // no recorders (the run is bound by the generators' throughput, not by this wave's latency)
            if (s == nb && tau0 + s == a.nburn && a.init_moments) {  // src/mcpar.cc:99-104
#pragma unroll
              for (int b = 0; b < BPL; ++b) {
                me[b] = mo[b] = splat2(0.0f);
                se[b] = so[b] = splat2(FPEPS);
              }
            }
            record(xe, xo, NEGSUM ? 0.0f - ly : ly, wbuf[(p & 3) * PKMAX + s], tau0 + s - a.nburn == a.snap_after);
          }
        }
      }
    } else {
      if (recorder && working && p >= 1) {  // Welford, snapshot, emit of phase p - 1 (src/mcpar.cc:176-209)
        const int pb = buf ^ 1, tau0 = (p - 1) * K;
        const int ns = T - tau0 < K ? T - tau0 : K;
        const int nb = a.nburn - tau0 < 0 ? 0 : (a.nburn - tau0 < ns ? a.nburn - tau0 : ns);
        if (nb < ns) {
          if (tau0 + nb == a.nburn && a.init_moments) {  // src/mcpar.cc:99-104
#pragma unroll
            for (int b = 0; b < BPL; ++b) {
              me[b] = mo[b] = splat2(0.0f);
              se[b] = so[b] = splat2(FPEPS);
            }
          }
          const float4 *xq = xbuf + ((size_t)(pb * K + nb) * OB + slot_o * BPL) * 64 + lane;
          const float *lq = lbuf + ((size_t)(pb * K + nb) * OWN + slot_o) * CPW + lane / LPC2;
          const float *wq = wbuf + ((p - 1) & 3) * PKMAX;
          for (int s = nb; s < ns; ++s) {
            f32x2 ce[BPL], co[BPL];
#pragma unroll
            for (int b = 0; b < BPL; ++b) {
              const float4 xv = xq[b * 64];
              ce[b] = f32x2{xv.x, xv.y};
              co[b] = f32x2{xv.z, xv.w};
            }
            const float lyv = NEGSUM ? 0.0f - *lq : *lq;  // (the owner hands over S_cur)
            const float w = wq[s];
            xq += (size_t)OB * 64;
            lq += (size_t)OWN * CPW;
            record(ce, co, lyv, w, tau0 + s - a.nburn == a.snap_after);
          }
        }
      }
      // (the recorders do not fill: they are the slowest wavefronts of the workgroup as it is -- with one owner, a
      // recorder that was dealt an item held every phase up: 0.283 -> 0.268 ms per launch on 8-D x 4096 chains)
      if (p + 1 < nphase && !recorder) fill(p + 1, p * K > a.nburn ? 2 : 1);  // (what runs beside this fill: phases p and p - 1)
    }
#ifdef MCX_PERSIST_TRACE
    unsigned long long trace_t1 = 0;
    if (a.trace_clk && blockIdx.x < PTRACE_WG && p < PTRACE_PH) trace_t1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef MCX_PERSIST_TRACE
    if (a.trace_clk && blockIdx.x < PTRACE_WG && p < PTRACE_PH && lane == 0) {
      unsigned long long *tc = a.trace_clk + ((((size_t)blockIdx.x * PWAVES + wv) * PTRACE_PH) + p) * 2;
      tc[0] = trace_t1;
      tc[1] = __builtin_amdgcn_s_memtime();
    }
#endif
  }

#ifdef MCX_PERSIST_TRACE
  if (a.trace_clk && blockIdx.x < PTRACE_WG && threadIdx.x == 0) {
    unsigned long long *tc = a.trace_clk + ((((size_t)blockIdx.x * PWAVES + 15) * PTRACE_PH) + (PTRACE_PH - 2)) * 2;
    tc[0] = trace_t_entry;
    tc[1] = trace_t_ready;
    tc[2] = trace_t_filled;
    tc[3] = __builtin_amdgcn_s_memtime();  // the phase loop is over
  }
#endif
  // ---- epilogue ---------------------------------------------------------------------------------------------
  // (every iteration ends with a barrier: a flag raised during the last phase is visible here)
  if (__hip_atomic_load(&lds_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) aborted = true;
  // abandoned launch: no state is written back, the host repeats the run (ctr[5] is set)
  if (!aborted) {
  if (owner) {
    if (live) {
#pragma unroll
      for (int b = 0; b < BPL; ++b) {
        const int kb = k0 + 4 * b;
        *reinterpret_cast<float4 *>(a.x + off + 4 * b) = make_float4(xe[b].x, xo[b].x, xe[b].y, xo[b].y);
        if ((a.nburn > 0 || a.T0) && chain == 0) {  // the (rescaled) diagonal; off-diagonal entries are zero on this path
          a.T[(kb + 0) * d + kb + 0] = te[b].x; a.T[(kb + 2) * d + kb + 2] = te[b].y;
          a.T[(kb + 1) * d + kb + 1] = to[b].x; a.T[(kb + 3) * d + kb + 3] = to[b].y;
        }
      }
    }
    if (mine && q == 0) {
      a.ly[chain] = NEGSUM ? 0.0f - ly : ly;
      a.acc_cnt[chain] = a.fresh ? cnt : a.acc_cnt[chain] + cnt;
    }
    const unsigned long long macc = wave_accepts(cnt_mark);  // accepted main-loop proposals of this wave
    if (lane == 0 && macc) atomicAdd(a.ctr + 4, macc);
    const unsigned long long bt = wave_sum_chains(btail);    // and its burn-in proposals that no meeting counted
    if (lane == 0 && bt) atomicAdd(a.ctr + 3, bt);
  }
  if ((REC ? recorder : owner) && live && a.nmain > 0) {
    const f32x2 w2 = splat2(a.winv[a.isamp0 + a.nmain - 1]);
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
      *reinterpret_cast<float4 *>(a.mu + off + 4 * b) = make_float4(me[b].x, mo[b].x, me[b].y, mo[b].y);
      *reinterpret_cast<float4 *>(a.psum2 + off + 4 * b) = make_float4(se[b].x, so[b].x, se[b].y, so[b].y);
      if (a.final_publish) {  // src/mcpar.cc:202-208 after the last step
        const f32x2 ve = se[b] * w2, vo = so[b] * w2;
        *reinterpret_cast<float4 *>(a.sig + off + 4 * b) = make_float4(ve.x, vo.x, ve.y, vo.y);
        float4 *slot = reinterpret_cast<float4 *>(a.musig_own + 2 * (off + 4 * b));
        slot[0] = make_float4(me[b].x, ve.x, mo[b].x, vo.x);
        slot[1] = make_float4(me[b].y, ve.y, mo[b].y, vo.y);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(a.ctr + 1, tun_na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (atomics: a.report)
    __hip_atomic_store(a.ctr + 2, tun_nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (burn_acc) atomicAdd(a.ctr + 3, burn_acc);  // (what the meetings counted; the waves add the rest; the run's block starts at zero)
    *a.ntrace = (a.fresh ? 0 : *a.ntrace) + ntrace_local;
  }
  }
  if (a.report) {  // (see RunArgs)
    // No fence on this path: a release at agent or system scope writes the L2's dirty lines back first -- 10 us and more
    // behind a launch that streamed sample rows -- which is what the host would then be waiting for (measured: nothing gained
    // over the copy).  Only the seven words matter here, and every access to them is an atomic at agent scope, performed at
    // the memory side: each wavefront waits for its own (s_waitcnt) before the workgroup's barrier, the ticket follows the
    // barrier, the last workgroup's loads follow its ticket.  Everything else the launch wrote is read through the stream,
    // behind the launch's own end-of-kernel release.
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (__hip_atomic_fetch_add(a.report_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {  // the grid's last workgroup
        __hip_atomic_store(a.report_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = __hip_atomic_load(a.ctr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 7; ++k) __hip_atomic_store(a.report + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);  // the words have arrived (host memory is written through) ...
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __hip_atomic_store(a.report + 7, a.report_serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // ... now the serial number
      }
    }
  }
}

}  // namespace mcx
