// mcx_engine_internal.hpp -- what the translation units of libmcx.so's host side share: the engine's state, the error
// and check macros, the device buffers, and the handful of functions that cross the seams between
//   mcx_engine.hip    the C ABI's create / destroy / options / run (the plan executor) / getters / standalone operators
//   mcx_plan.hip      the schedule of one run (pure host logic, exported as mcx_plan)
//   mcx_exchange.hip  the inter-shard exchange: begin / wait / publish, the RCCL shim, mcx_exchange_rccl_*
//   mcx_sink.hip      the streaming sample sink, its text side, mcx_samples_text / mcx_format_rows
//   mcx_murray.hip    genRemote on device buffers: draws, sweeps, decisions, the two exact screens
// Nothing here is part of the library's interface (include/mcx.h is); every cross-unit function has hidden visibility.
#pragma once
#include "../../include/mcx.h"
#include "mcx_device.hpp"
#include "mcx_launch.hpp"
#include "mcx_persist.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
// RCCL types only: librccl.so.1 is loaded at run time (mcx_rccl_*), never linked, and a build machine without the
// RCCL development headers gets the handful of declarations the dlopen shim needs
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
#endif

#include <algorithm>
#include <chrono>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace mcx;

#define MCXI __attribute__((visibility("hidden")))

// ---------------------------------------------------------------------------------------------
// error plumbing (mcx_engine.hip)
// ---------------------------------------------------------------------------------------------
MCXI int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
MCXI int need_device();

#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? MCX_ERR_NO_DEVICE \
                                                                        : MCX_ERR_HIP,      \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define MCXCHK(expr)          \
  do {                        \
    int s_ = (expr);          \
    if (s_ != MCX_OK) return s_; \
  } while (0)


// Test hook, not part of the public header: mcx_set_option(e, 12, k) makes the tuner meetings of the one-launch small-n
// kernel wait for k workgroups more than the grid has, i.e. they can never complete (tests/test_gpu_small_n_safety.py).
enum { MCX_OPT_DEBUG_MEET = 12 };

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
static inline int lpc_for(int d)
{
  const int nb = (d + 3) / 4;
  int l = 1;
  while (l < nb) l <<= 1;
  return l;
}
static inline int dmax_for(int d)
{
  int m = 2;
  while (m < d) m <<= 1;
  return m;
}
static inline unsigned nblocks(size_t threads) { return (unsigned)((threads + BLOCK - 1) / BLOCK); }

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  int alloc(size_t count)
  {
    if (count <= n && p) return MCX_OK;
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e != hipSuccess) {
      (void)hipGetLastError();  // clear the sticky error: later launch checks must not see it
      p = nullptr;
      n = 0;
      return fail(MCX_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    }
    n = count;
    return MCX_OK;
  }
  void release()
  {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// pinned host staging (host-callback likelihoods move nc*np floats out and nc floats in every step)
template <typename T>
struct PinBuf {
  T *p = nullptr;
  size_t n = 0;
  int alloc(size_t count)
  {
    if (count <= n && p) return MCX_OK;
    release();
    if (count == 0) count = 1;
    if (hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      return fail(MCX_ERR_ALLOC, "hipHostMalloc(%zu bytes) failed", count * sizeof(T));
    }
    n = count;
    return MCX_OK;
  }
  void release()
  {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
  }
};

// device-side likelihood descriptor built from an mcx_vlfunc
// mcx_user.hip: a user's likelihood source compiled into the step kernels at run time (MCX_VL_SOURCE)
struct UserLik;
int user_lik_get(const char *source, int np, std::shared_ptr<UserLik> *out);  // compiled once per (source, np)
int user_lik_launch_fused(const UserLik &u, bool main, const SegArgs &a, hipStream_t st);
int user_lik_launch_eval(const UserLik &u, const float *x, float *y, int n, int d, const float *par, int ncomp, hipStream_t st);
int user_lik_variant(int lpc, const SegArgs &a);  // 0 hot-path kernel, 1 its full-covariance form, 2 the generic kernel
double user_lik_compile_ms(const UserLik &u);
bool user_lik_small_ok(const UserLik &u);  // block form, <= 8 lanes per chain: the one-launch small-n kernel can be built for it
hipError_t user_lik_launch_small(UserLik &u, int bpl, const RunArgs &a, hipStream_t st);  // (mcxk_launch_persist's role)

struct LikDev {
  int kind = 0;  // LikKind, or MCX_VL_HOST / MCX_VL_DEVICE
  int ncomp = 0;
  DevBuf<float> params;
  std::vector<float> host;  // staging for the asynchronous upload (must outlive it)
  mcx_host_fn fn = nullptr;
  void *ctx = nullptr;
  std::shared_ptr<UserLik> user;  // LIK_USER
  bool fusable() const { return kind == LIK_ROSEN1 || kind == LIK_GAUSS || kind == LIK_MIX || kind == LIK_ROSEN2F || kind == LIK_USER; }
};


// ---------------------------------------------------------------------------------------------
// kernel dispatch on (LPC, likelihood)
// ---------------------------------------------------------------------------------------------
#define DISPATCH_LPC(lpc, CALL)                                   \
  switch (lpc) {                                                  \
  case 1: { constexpr int LPC_ = 1; CALL; } break;                \
  case 2: { constexpr int LPC_ = 2; CALL; } break;                \
  case 4: { constexpr int LPC_ = 4; CALL; } break;                \
  case 8: { constexpr int LPC_ = 8; CALL; } break;                \
  case 16: { constexpr int LPC_ = 16; CALL; } break;              \
  case 32: { constexpr int LPC_ = 32; CALL; } break;              \
  case 64: { constexpr int LPC_ = 64; CALL; } break;              \
  default: return fail(MCX_ERR_UNSUPPORTED, "np > 256 is not supported"); \
  }

// chains per lane of the Murray sweep when np == DMAX (mcx_device.hpp, sweep_rows2): two at 16-D and 32-D, where the
// LDS broadcast reads bind with one (measured on one box: R-murray jobs 3.5 % faster at 16-D, 5 % faster sweeps at 32-D;
// the coarser early-outs of 128 chains per wavefront cost less than the reads save)
#define SWEEP_CPL(DMAX) ((DMAX) == 16 || (DMAX) == 32 ? 2 : 1)
#define DISPATCH_DMAX(dm, CALL)                                   \
  switch (dm) {                                                   \
  case 2: { constexpr int DMAX_ = 2; CALL; } break;               \
  case 4: { constexpr int DMAX_ = 4; CALL; } break;               \
  case 8: { constexpr int DMAX_ = 8; CALL; } break;               \
  case 16: { constexpr int DMAX_ = 16; CALL; } break;             \
  case 32: { constexpr int DMAX_ = 32; CALL; } break;             \
  case 64: { constexpr int DMAX_ = 64; CALL; } break;             \
  default: return fail(MCX_ERR_UNSUPPORTED, "internal: register Murray kernels cover np <= 64"); \
  }

// ---------------------------------------------------------------------------------------------
// the engine
// ---------------------------------------------------------------------------------------------
// counters of one run: [0..7] tuner / accept totals, [8..] the tuner events' meeting words of k_run_small
constexpr int CTR_WORDS = 8 + PEVENTS * PLEAVES, CTR_RING = 16;

constexpr int SINK_RING = 4;  // blocks of the device ring in sink mode

struct EvPair {
  hipEvent_t a, b;
  int kind;
  uint64_t chain_steps;
};

struct mcx_engine {
  // problem (src/mcpar.hh:47-59)
  int nparam, nchain, ntot, ncov, size, rank, tchains;
  float PLOCAL, TGT_ARATE_MIN, TGT_ARATE_MAX, SCALE_DEC, SCALE_INC;
  int SYNCSTEP;
  uint32_t seed, tbase = 0;
  int lpc, vec4;
  int device = 0;  // the HIP device the engine lives on (current device at mcx_create)
  // device state (src/mcpar.hh:61-88)
  DevBuf<float> pvals, ptrial, mu, sig, psum2, mutrial, sigtrial, musigall, winvall;
  DevBuf<float> lylast, lytrial, cfac, cmax, cov, cov0, trace;  // cov0 = the factor as installed (cov is rescaled by the tuner)
  DevBuf<uint32_t> acc_cnt, acc_slots;
  int nslots = 0;
  DevBuf<unsigned long long> ctr;  // [0..3] tuner (k_tuner), [4] main-loop accepts
  DevBuf<int> active0, active1, nact, ntrace;  // nact: [0] survivors of the pass; as u64: [1 .. 1 + CULL_NCOUNT) / the next
                                               // CULL_NCOUNT cells: pairs kept by the exclusion tests of the min-arg sweep /
                                               // of the sum sweeps of this genRemote call (spread: same-address atomics are slow)
  // exact exclusion of far Gaussians in the Murray sweeps (mcx_remote.hpp, k_cull_*)
  DevBuf<unsigned> cull_keys, cull_hist;
  DevBuf<int> cull_sorted;
  DevBuf<unsigned long long> tun_cells;  // SegArgs::Tuner::cells
  DevBuf<unsigned long long> text_wg;    // mcx_samples_text: per-workgroup byte counts / offsets
  DevBuf<char> text_dev;                 // and the text itself
  DevBuf<float> cull_stats, cull_box, cull_lim;
  DevBuf<double> proj_acc, proj_p, proj_lohi;  // mcx_cull_proj.hpp: two power iterations' sums, e.x per chain, [lo, hi] per group
  DevBuf<unsigned long long> cull_excl;
  DevBuf<unsigned short> scr_a, scr_b;  // mcx_screen.hpp: A' per position of the sorted list, B' per Gaussian (bf16)
  DevBuf<float> scr_centre;
  DevBuf<float> cand;  // Murray passes over few chains: the next passes' proposals (p, mu, sig per candidate, then racpt)
  int opt_cull = -1;  // -1 auto (the per-pair bound: many chains, many Gaussians, np = 16 or 32), 0 off, whenever the kernels
                      // allow: 1 boxes, 2 one direction (mcx_cull_proj.hpp), 3 the per-pair bound (mcx_screen.hpp)
  int cull_skip[2] = {0, 0};  // auto mode: genRemote calls for which the min-arg / sum sweeps go without the test,
                              // because it excluded too little last time it was tried (then it is tried again)
  DevBuf<float> samp_x, samp_ly, winv_tab, psum, pmax, racpt, pinit_dev, zpre, upre, trash;
  bool pinit_staged = false;
  DevBuf<uint8_t> mask;
  // host staging
  PinBuf<float> h_ptrial, h_lytrial;
  PinBuf<unsigned long long> h_ctr;  // the run's counters, read back once at its end
  unsigned long long remote_serial = 0;  // Murray passes so far (what k_remote_decide signs its counters with)
  PinBuf<unsigned long long> h_nact;  // a Murray pass's survivor count (and the exclusion tests' counters)
  std::vector<float> h_cov, h_cov_dev, h_winv;  // h_cov_dev = what cov0 holds
  bool cov_pending = false;  // cov has not been reset to cov0 for the current run yet
  bool cov_offdiag = false;  // cov (device) may hold non-zero entries below the diagonal
  int ctr_set = 0;           // counter block of the current run (ring of CTR_RING blocks, zeroed when it wraps)
  int meet_fd = -1;          // lock file of this GPU: at most one kernel with grid-wide meetings in flight (see meet_lock_open)
  bool meet_held = false;    // this engine holds the lock: a launch with meetings may still be running
  bool meet_check = false;   // a launch with meetings is in flight: its "abandoned" word has not been looked at yet
  unsigned long long *meet_word = nullptr;  // that word (ctr[5] of the run's counter block)
  bool persist_broken = false;  // a meeting was abandoned once on this engine: the one-launch kernel is not used again
  int opt_meet_timeout_ms = 50, opt_debug_meet = 0;
  int opt_meet_under_gather = -1;  // may a launch with tuner meetings start under this engine's in-flight gather: -1 auto (= no), 0 no, 1 yes
  // run bookkeeping
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int opt_stride = 1;
  int opt_split = -1;  // small-n mode: -1 auto, 0 off, 1 on (when the hot-path kernel applies)
  int opt_bpl = 0;     // 4-parameter blocks per lane of the hot-path kernel: 0 auto, 1, 2, 4
  int opt_persist = -1;  // small-n mode, one launch per stretch of local steps (k_run_small): -1 auto, 0 off, 1 on
  int ncu = 0;           // compute units of the device (the persistent grid must be resident at once)
  int opt_samples = 1, opt_mask = 0, opt_fuse = 1, opt_maxseg = 256, opt_profile = 0, opt_eager = 0, opt_async_tail = 1, opt_sink_text = 0;
  int last_nsamp = 0, last_nburn = 0, samp_steps = 0;
  bool have_run = false, diag = true, xchg_pending = false;
  int published_steps = 0;  // main-loop steps reflected in this shard's musigall slot
  int tail_publish = 0;     // > 0: the run's last gather is still in flight; publish that many steps once it is done (finish_tail)
  LikDev lik;
  mcx_exchange_fn xfn = nullptr;
  void *xctx = nullptr;
  // native RCCL exchange (mcx_exchange_rccl_*): in-place ncclAllGather of the musigall slots on a side stream
  ncclComm_t xcomm = nullptr;
  bool xcomm_owned = false;
  // MCX_OPT_ASYNC_RUN: a run whose kernels are queued and whose end nobody has waited for yet (mcx_engine.hip: finish_pending)
  int opt_async_run = 0;
  int opt_reference_calls = 0;  // MCX_OPT_REFERENCE_CALLS: make the reference's discarded per-chain L(1, pvals_j) calls (host functors)
  struct PendingRun {
    bool active = false;
    int nsamp = 0, nburn = 0;
    uint32_t tbase0 = 0;                 // the RNG step index the run started from (a repeat starts there again)
    bool meet_check = false;             // its one-launch kernel had tuner meetings: word 5 of its counters says if one was abandoned
    unsigned long long *hctr = nullptr;  // its slot of the pinned counter ring
    bool host_pinit = false;             // it started from caller memory (kept in pinit_async for a repeat), not from the staged state
    int slot = 0;                        // its counter slot / events
    unsigned long long serial = 0;       // != 0: its last launch reports to the slot itself (RunArgs::report) and stores this last
  } pend;
  hipStream_t astream = nullptr;         // an asynchronous run's counters travel on it, beside the next run's kernels
  // Counter slots in pinned memory, one per run in turn.  FOUR: run k is queued once run k-2's kernels are over (two in flight),
  // run k-2's counters may leave only when run k-1 ends (their copy kernel finds no room beside a grid that fills the device),
  // and run k-3's are what the books look at meanwhile.
  static constexpr int HSLOTS = 4;
  hipEvent_t copy_ev[HSLOTS] = {nullptr, nullptr, nullptr, nullptr};
  bool copy_pending[HSLOTS] = {false, false, false, false};
  unsigned superseded_mask = 0;  // slots of runs nobody looked at before the next was queued (and that had tuner meetings): for the books only
  DevBuf<float> pinit_async;
  int hctr_slot = 0;
  hipEvent_t run_ev[HSLOTS] = {nullptr, nullptr, nullptr, nullptr};  // recorded behind each asynchronous run's last command, by counter slot
  bool run_queued[HSLOTS] = {false, false, false, false};
  // MCX_OPT_SELF_REPORT: a run that ends with a launch of the one-launch kernel has that launch write the counters to the slot
  // and its serial number behind them (RunArgs::report): nothing is queued behind the kernel, the host spins on the word
  int opt_self_report = 1;
  unsigned long long report_serial = 0;                    // serial numbers handed out so far
  double report_wait_us = 0.0;  // how long the last self-reporting run kept the host waiting: a run that takes longer than the
                                // spin allows is not spun for at all the next time (a core's 1.5 ms are somebody else's)
  unsigned long long slot_serial[HSLOTS] = {0, 0, 0, 0};   // the serial the slot's run will store (0: its counters come by copy)
  hipStream_t mstream = nullptr;   // Murray passes by column chunks: the sweeps' stream (mcx_murray.hip: screen_sweep_chunked)
  std::vector<hipEvent_t> mev;
  int opt_murray_overlap = 0;
  hipStream_t xstream = nullptr;
  hipEvent_t xready = nullptr, xdone = nullptr;
  mcx_output_fn ofn = nullptr;
  void *octx = nullptr;
  // streaming sample sink (mcx_set_sink): ring of SINK_RING blocks in samp_x / samp_ly, staged out on cstream
  mcx_sink_fn sfn = nullptr;
  mcx_text_sink_fn tfn = nullptr;  // mcx_set_text_sink: the blocks as text instead of rows (one of the two at most)
  DevBuf<unsigned long long> sink_text_wg[2];  // per staging buffer: the text kernels' byte counts / offsets
  PinBuf<unsigned long long> sink_text_total[2];
  DevBuf<char> sink_text_dev[2];   // per staging buffer: the block's text on the device ...
  PinBuf<char> sink_text_pin;      // ... and where it lands on the host (copied on tstream, under the next block's formatting)
  size_t sink_text_bytes[2] = {0, 0};
  bool sink_text_ok[2] = {false, false};
  int sink_text_written = 0, sink_text_copied = 0;  // blocks of this run whose text has been formatted / sent off
  bool run_sink_text = false;        // this run's row sink also gets every block's text (MCX_OPT_SINK_TEXT)
  const char *cb_text = nullptr;     // valid while a sink callback runs: mcx_sink_text
  size_t cb_text_bytes = 0;
  void *sctx = nullptr;
  int sink_block = 0;  // main-loop steps per block (0 = no sink: the whole run stays in HBM)
  bool run_sink = false;           // the current / last run streamed its samples
  int last_sink_total = 0;         // kept steps handed to the copy stream so far
  int run_sblock = 0, run_kb = 0;  // its block length in steps / in kept steps
  hipStream_t cstream = nullptr;
  hipEvent_t ev_steps[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
  hipStream_t tstream = nullptr;   // the text's copies to the host
  hipEvent_t ev_write[2] = {nullptr, nullptr}, ev_text = nullptr;
  DevBuf<float> sink_stage[2];
  PinBuf<float> sink_pin[2];
  DevBuf<float> best_row;            // running maximum-likelihood sample: [0] = log-likelihood, [1..np] = parameters
  DevBuf<unsigned long long> best_key;  // scratch of the arg-max reduction
  mcx_counters cnt{};
  DevBuf<unsigned long long> trace_clk;  // MCX_PERSIST_TRACE (debug builds): per-wavefront phase clocks of the last small-n launch
  DevBuf<uint32_t> deal_tab;         // RunArgs::deal of the one-launch small-n kernel, for the configuration in deal_key
  std::vector<uint32_t> h_deal;
  long long deal_key = -1;
  uint64_t meet_total = 0;           // runs repeated because a meeting was abandoned, over the engine's life
  int runs_since_broken = 0;         // runs on the per-segment kernels since then (the one-launch kernel is tried again)
  // time the step stream waits for gathers begun earlier (mcx_counters.exchange_wait_ns): event pairs around each wait
  std::chrono::steady_clock::time_point ht_mark[3];  // MCX_VERBOSE=2: first launch queued / everything queued / stream idle
  std::vector<std::pair<hipEvent_t, hipEvent_t>> xw_pool;
  size_t xw_used = 0;
  std::vector<EvPair> evs;
  mcx_profile prof{};
};

// every entry point may be called from a thread whose current device is another one; whatever an asynchronous run
// (MCX_OPT_ASYNC_RUN) left in flight is finished first -- waited for, its counters taken, a run whose tuner meeting was
// abandoned repeated -- so that no entry point ever sees a run half done (mcx_run itself queues behind it instead)
int finish_pending(mcx_engine *e);  // mcx_engine.hip
static inline int enter_raw(mcx_engine *e)
{
  if (!e) return fail(MCX_ERR_INVALID, "engine is NULL");
  HIPCHK(hipSetDevice(e->device));
  return MCX_OK;
}
static inline int enter(mcx_engine *e)
{
  MCXCHK(enter_raw(e));
  if (e->pend.active) MCXCHK(finish_pending(e));
  return MCX_OK;
}

struct ProfScope {
  mcx_engine *e;
  EvPair p{};
  bool on;
  ProfScope(mcx_engine *e_, int kind, uint64_t cs) : e(e_), on(e_->opt_profile != 0)
  {
    e->cnt.kernel_launches++;
    if (!on) return;
    p.kind = kind;
    p.chain_steps = cs;
    (void)hipEventCreate(&p.a);
    (void)hipEventCreate(&p.b);
    (void)hipEventRecord(p.a, e->stream);
  }
  ~ProfScope()
  {
    if (!on) return;
    (void)hipEventRecord(p.b, e->stream);
    e->evs.push_back(p);
  }
};


// ---------------------------------------------------------------------------------------------
// across the seams
// ---------------------------------------------------------------------------------------------
struct PlanCfg {
  int nsamp, nburn, sync;
  float pl;
  uint32_t seed, tbase;
  bool sharded, eager, fused, output_hook;
  int maxseg;
  int sink_block;  // > 0: cut the main loop into blocks of this many steps for the sample sink
};
MCXI std::vector<mcx_plan_item> build_plan(const PlanCfg &c);  // mcx_plan.hip

constexpr int MCX_INTERNAL_MEET_ABANDONED = 1000;  // run_once: a tuner meeting of the one-launch kernel was abandoned
MCXI int meet_release(mcx_engine *e, bool stream_is_idle);  // mcx_engine.hip

// mcx_exchange.hip
MCXI int exchange_begin(mcx_engine *e);
MCXI int exchange_wait(mcx_engine *e);
MCXI void xwait_collect(mcx_engine *e);
MCXI int publish(mcx_engine *e, int steps_done);
MCXI int finish_tail(mcx_engine *e);
MCXI bool exchange_tail_may_stay_in_flight(const mcx_engine *e);

// mcx_sink.hip
MCXI int sink_block_done(mcx_engine *e, int done, int nsteps, int seq);
MCXI int sink_drain(mcx_engine *e, int nblocks_done);
MCXI void samp_vbase(const mcx_engine *e, int isamp, float **px, float **pl);  // (mcx_engine.hip)

// cells of the Murray screens' "pairs kept" counters behind mcx_engine::nact (= CULL_NCOUNT of mcx_remote.hpp, which only
// mcx_murray.hip includes and checks)
constexpr int NACT_CULL_CELLS = 64;

// mcx_murray.hip: MCPar::genRemote on device buffers (src/mcpar.cc:315-451)
MCXI int remote_device(mcx_engine *e, uint32_t t, const float *pvals, const float *musigall, float *ptrial, float *cfac,
                       float *mutrial, float *sigtrial, int *npass_out);
