/*
 * mcx.h -- C ABI of the MI355X-native parallel Metropolis-Hastings engine (libmcx.so).
 *
 * This is the drop-in boundary for the chain-step hot path of rplzzz/mcpar.  The reference has
 * no C ABI of its own (its plug-in mechanism is a C++ abstract class linked into the same
 * executable), so each entry point below cites the C++ member of the reference it replaces; the
 * reference-compatible C++ classes (include/mcpar/{vlfunc,mcpar,mcout,rosenbrock}.hh) are thin
 * shims over these functions.  See INTEGRATION.md for the binding a maintainer would add.
 *
 * Plain pointers and sizes only.  Unless a parameter says "device", pointers are host memory.
 * All arithmetic is float32 ("MCX arithmetic v3"); accept decisions are bit-exact
 * against oracle/mcx_oracle.c for a fixed seed.
 *
 * Every function returns MCX_OK (0) or an mcx_status; mcx_last_error() gives the message.
 * There is no CPU fallback: without a gfx950 device every compute entry point fails with
 * MCX_ERR_NO_DEVICE.
 */
#ifndef MCX_H_
#define MCX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCX_ABI_VERSION 4

typedef enum mcx_status {
  MCX_OK = 0,
  MCX_ERR_INVALID = 1,     /* bad argument (the reference throws a string literal: src/mcpar.cc:268) */
  MCX_ERR_NO_DEVICE = 2,   /* no HIP device / not gfx950 */
  MCX_ERR_HIP = 3,         /* a HIP call failed (the reference abort()s on VSL errors: src/mcpar.hh:93) */
  MCX_ERR_UNSUPPORTED = 4, /* np > 256 */
  MCX_ERR_ALLOC = 5,       /* sample store does not fit (reference: exit(2), src/mcpar.cc:34-40) */
  MCX_ERR_EXCHANGE = 6,    /* exchange hook failed / missing (reference: MPI_Abort, src/mcpar.cc:133-137) */
  MCX_ERR_VLFUNC = 7       /* host likelihood callback missing */
} mcx_status;

/* ---- likelihood plug-in: replaces class VLFunc (src/vlfunc.hh:9-12) ---------------------- */
enum {
  MCX_VL_ROSENBROCK1 = 1, /* Rosenbrock1::operator()  src/rosenbrock.cc:4-21   (fused on device) */
  MCX_VL_ROSENBROCK2 = 2, /* Rosenbrock2::operator()  src/rosenbrock.cc:25-41  as written (device, unfused) */
  MCX_VL_GAUSSIAN = 3,    /* Gaussian::operator()     src/rosenbrock.cc:44-61  any d      (fused) */
  MCX_VL_DUALGAUSS = 4,   /* DualGaussian::operator() src/rosenbrock.cc:63-78             (fused) */
  MCX_VL_GAUSSMIX = 5,    /* N-D K-component unit-variance mixture (BASELINE config 5)    (fused) */
  MCX_VL_ROSENBROCK2_FIXED = 6, /* NOT reference behaviour, a flagged variant: the overlapping Rosenbrock function made
                             well-posed -- src/rosenbrock.cc:25-41 with '+' on the second term (the '-' at :38 leaves
                             log L unbounded above) and the loop kept inside each parameter set:
                             - sum_{k < d-1} (1 - x_k)^2 + 100 (x_{k+1} - x_k^2)^2                      (fused) */
  MCX_VL_HOST = 100,      /* any user VLFunc subclass: device -> host callback -> device  */
  MCX_VL_DEVICE = 101,    /* user likelihood as a GPU kernel: stays on the device (see mcx_vlfunc.ctx) */
  MCX_VL_SOURCE = 102     /* user likelihood as HIP SOURCE of device functions, compiled into the engine's fused step
                             kernels at run time (hiprtc): one launch per segment like the built-ins (see below) */
};

/* MCX_VL_SOURCE -- the VLFunc contract (src/vlfunc.hh:9-12) for ONE parameter set, as device code the engine inlines into
 * its own step kernels.  ctx = the NUL-terminated source text, params / ncomp = `ncomp` floats handed to the functions as
 * `par` (copied to the device by mcx_run).  The text defines, in the global namespace, EITHER
 *
 *     __device__ float mcx_user_loglike(const float *x, int d, const float *par);      // log L of x[0..d-1]
 *
 * (x points to on-chip memory; every call sees one whole parameter set) OR, for likelihoods that are a sum over
 * blocks of four consecutive parameters -- the engine's own decomposition, one GPU lane per block --
 *
 *     #define MCX_USER_BLOCK_FORM
 *     __device__ float mcx_user_block(const float xb[4], int nv, int k0, int d, const float *par);
 *         // partial of parameters k0 .. k0+nv-1 (nv = 4 except in the last block when d % 4 != 0)
 *     #define MCX_USER_FINISH                                                            // optional
 *     __device__ float mcx_user_finish(float sum, int d, const float *par);             // log L from the sum (default: sum)
 *
 * The partials are added in the order of the built-ins, so a restatement of a built-in returns
 * its bits.  MCX_USER_NP (= np) and MCX_USER_LPC (= the number of 4-parameter blocks, rounded up to a power of two) are
 * defined when the text is compiled: loops over them unroll, small arrays stay in registers.  In the whole-vector form a
 * chain of np <= 16 is held by ONE lane (four blocks per lane) and the function runs once per chain; above that by
 * np / 16 lanes, each of which evaluates it.  "mcx_numerics.hpp" is already included: mcx::logf_v1, mcx::expf_v2, ... are the engine's (and the CPU
 * oracle's) own transcendentals.  Compiled with -O3 -ffp-contract=off (write fma explicitly: __builtin_fmaf), once per
 * (text, np) per process; a text that does not compile fails mcx_run with MCX_ERR_VLFUNC and the compiler's messages
 * in mcx_last_error().  Without libhiprtc: MCX_ERR_UNSUPPORTED (MCX_VL_DEVICE and MCX_VL_HOST remain).  A text in block form
 * also gets the one-launch small-n kernel (few chains: burn-in, tuner meetings and main loop in ONE launch), built when a
 * run first qualifies for it. */
int mcx_user_source_available(void); /* 1 / 0 (mcx_last_error says why not) */
/* the compile step alone, for a chain of np parameters (needs no GPU): MCX_OK and the code object's size, or MCX_ERR_VLFUNC */
int mcx_debug_user_source_compile(const char *source, int np, size_t *code_bytes);
/* the same for the one-launch small-n kernel (few chains: the whole run in one launch), which takes the block form only and is
 * built when a run first wants it: bpl = 1 or 2 blocks per lane, rec = with recorder wavefronts */
int mcx_debug_user_source_compile_small(const char *source, int np, int bpl, int rec, size_t *code_bytes);
/* For MCX_VL_DEVICE without a compiler at hand: source of a whole kernel
 *   extern "C" __global__ void f(int npset, const float *x, float *y)   -> *function = its hipFunction_t (for mcx_vlfunc.ctx) */
int mcx_user_kernel_compile(const char *source, const char *symbol, void **function);

/* same contract as VLFunc::operator()(int npset, const float *x, float *restrict y): x is
 * row-major [npset][d], y is [npset]; the return code is ignored like the reference's
 * (doc/userguide.tex:146-149).  Called on the thread that called mcx_run. */
typedef int (*mcx_host_fn)(void *ctx, int npset, const float *x, float *y);

typedef struct mcx_vlfunc {
  int kind;            /* MCX_VL_* */
  int d;               /* parameters per set; must equal the engine's np */
  int ncomp;           /* GAUSSMIX: number of components K (<= 64); SOURCE: number of floats in params */
  const float *params; /* GAUSSIAN: mu[d], sig2[d] (NULL = standard normal); DUALGAUSS: w;
                          GAUSSMIX: means[K*d], weights[K]; SOURCE: the user's `par`.  Copied by mcx_run. */
  mcx_host_fn fn;      /* HOST */
  void *ctx;           /* HOST: passed to fn.  SOURCE: const char *, the source text.  DEVICE: a hipFunction_t (from the caller's own code object,
                          hipModuleGetFunction) of a kernel with the VLFunc contract on device memory,
                            extern "C" __global__ void f(int npset, const float *x, float *y);
                          launched with 256-thread blocks, ceil(npset / 256) blocks, on the engine's
                          stream; thread i evaluates parameter set i (or any mapping covering all sets). */
} mcx_vlfunc;

/* batched likelihood on host buffers: the VLFunc call itself, runs the device kernel */
int mcx_vlfunc_eval(const mcx_vlfunc *f, int npset, const float *x, float *y);

/* ---- engine: replaces class MCPar (src/mcpar.hh:10-91) ------------------------------------ */
typedef struct mcx_engine mcx_engine;

/* MCPar::MCPar(np, nc, mpisiz, mpirank, pl, armin, armax, dfac, ifac, sync) src/mcpar.hh:32-33,
 * src/mcpar.cc:216-272.  nshards/shard replace mpisiz/mpirank (one shard per GPU/process; this
 * shard owns global chains [shard*nc, (shard+1)*nc)).  seed replaces the literal 8675309 of
 * src/mcpar.cc:271.  Runs on the current HIP device of the calling thread. */
int mcx_create(mcx_engine **out, int np, int nc, int nshards, int shard, float pl, float armin,
               float armax, float dfac, float ifac, int sync, uint32_t seed);
/* MCPar::~MCPar  src/mcpar.cc:274-299 */
int mcx_destroy(mcx_engine *e);

/* MCPar::run(nsamp, nburn, pinit, L, outsamples, incov)  src/mcpar.hh:36-37, src/mcpar.cc:17-214.
 * pinit[nc*np] and incov[np*np] (or NULL = identity) are copied.  Samples go to the engine's
 * HBM-resident sample store (mcx_samples_*), which plays the role of MCout's vector. */
int mcx_run(mcx_engine *e, int nsamp, int nburn, const float *pinit, const mcx_vlfunc *L,
            const float *incov);
/* Stage the initial chain state pinit[nc*np] in HBM ahead of time; a later mcx_run(..., pinit = NULL,
 * ...) starts from the staged copy (device-to-device) instead of reading host memory. */
int mcx_stage_pinit(mcx_engine *e, const float *pinit);

/* MCPar::genLocal(pvals, ptrial, cfac)  src/mcpar.hh:40, src/mcpar.cc:302-312.  t is the RNG
 * step index; host buffers [nc*np], [nc*np], [nc].  Uses the engine's current
 * Cholesky factor (mcx_covar_setup / last run). */
int mcx_gen_local(mcx_engine *e, uint32_t t, const float *pvals, float *ptrial, float *cfac);
/* MCPar::genRemote(pvals, musigall, ptrial, cfac)  src/mcpar.hh:41-42, src/mcpar.cc:315-451.
 * musigall[nshards*nc*np*2] interleaved (mu, sig^2).  mutrial/sigtrial[nc*np] are the side
 * outputs the reference keeps in members (sigtrial returned squared, :447-448); npass = number
 * of rejection passes. */
int mcx_gen_remote(mcx_engine *e, uint32_t t, const float *pvals, const float *musigall,
                   float *ptrial, float *cfac, float *mutrial, float *sigtrial, int *npass);
/* MCPar::covar_setup(incov, cov)  src/mcpar.hh:38, src/mcpar.cc:454-484: cov[np*np] in/out,
 * identity if incov == NULL; also installs the factor in the engine. */
int mcx_covar_setup(mcx_engine *e, const float *incov, float *cov);

/* ---- inter-shard exchange: replaces MPI_Allgather(IN_PLACE, musigall) src/mcpar.cc:127-140 --
 * phase MCX_XCHG_BEGIN: start an in-place all-gather of musigall (device pointer, nshards slots
 * of slot_floats floats; this shard's slot is filled) ordered after the work already queued on
 * `stream` (a hipStream_t).  phase MCX_XCHG_WAIT: make `stream` wait for that gather.  A hook may
 * do all the work in BEGIN.  Required when nshards > 1. */
enum { MCX_XCHG_BEGIN = 0, MCX_XCHG_WAIT = 1 };
typedef int (*mcx_exchange_fn)(void *ctx, int phase, void *musigall_dev, size_t slot_floats,
                               int shard, int nshards, void *stream);
int mcx_set_exchange(mcx_engine *e, mcx_exchange_fn fn, void *ctx);

/* The exchange the library ships: an in-place ncclAllGather (RCCL over xGMI) of the musigall slots --
 * sendbuff = musigall + shard*2*ntot, 2*ntot floats per shard, the slot layout of src/mcpar.cc:206 -- on a
 * side stream, BEGIN = enqueue behind the engine's stream, WAIT = the engine's stream waits for it.
 * One process (or thread) per GPU, communicator rank = shard.  librccl.so.1 is loaded on first use.
 *   rank 0:     mcx_rccl_unique_id(id);  ship the MCX_RCCL_ID_BYTES bytes to every rank (MPI_Bcast, ...)
 *   every rank: mcx_exchange_rccl_init(e, id);     collective, like MPI_Comm_dup at src/mcpar.cc:228
 * or hand over a communicator the caller already has (ncclComm_t, rank == shard, nranks == nshards). */
#define MCX_RCCL_ID_BYTES 128
int mcx_rccl_available(void); /* 1 / 0 (mcx_last_error says why not) */
int mcx_rccl_unique_id(void *id);
int mcx_exchange_rccl_init(mcx_engine *e, const void *id);
int mcx_exchange_rccl_adopt(mcx_engine *e, void *nccl_comm);
int mcx_exchange_rccl_destroy(mcx_engine *e); /* also done by mcx_destroy */
/* what the installed RCCL exchange sees: ncclCommCount / ncclCommUserRank of its communicator (a launcher's
 * start-up check that every rank really joined); MCX_ERR_EXCHANGE when no RCCL exchange is installed */
int mcx_exchange_rccl_info(mcx_engine *e, int *nranks, int *rank);
/* run the installed exchange hook once, now (BEGIN, WAIT, drain): start-up self-check / tests */
int mcx_debug_exchange(mcx_engine *e);
/* fill this shard's musigall slot with `value` (start-up self-check of an exchange: every shard fills its slot with
 * its own number, mcx_debug_exchange, then mcx_get_musigall must show slot r full of shard r's number) */
int mcx_debug_fill_slot(mcx_engine *e, float value);

/* called where the reference dumps output (isamp % outstep == 0 && isamp > 0, and after the
 * last step: src/mcpar.cc:110-119,212) with the number of main-loop steps completed. */
typedef int (*mcx_output_fn)(void *ctx, int steps_done);
int mcx_set_output_hook(mcx_engine *e, mcx_output_fn fn, void *ctx);

/* Streaming sample sink: the role of MCout as the reference fills it (src/mcpar.cc:176-182) and dumps it every
 * outstep steps (:110-119), for runs whose samples should not all stay in HBM.  With a sink installed the
 * engine keeps only a ring of 4 blocks of `block_steps` main-loop steps on the device; when a block is complete
 * its rows -- MCout layout, np+1 columns, step-major then chain -- are interleaved on the device and copied to
 * pinned host memory on a second stream while the step stream goes on with the next blocks, and fn is called on
 * the thread that called mcx_run with rows valid for the duration of the call (first_step / nsteps count KEPT
 * steps, see MCX_OPT_SAMPLE_STRIDE).  nsamp is then bounded by nothing but the int32 range; mcx_samples_copy
 * serves only what the ring still holds, mcx_samples_maxlike keeps its running maximum on the device.
 * fn == NULL removes the sink.  block_steps is rounded up to a multiple of the sample stride. */
typedef int (*mcx_sink_fn)(void *ctx, int first_step, int nsteps, const float *rows);
int mcx_set_sink(mcx_engine *e, mcx_sink_fn fn, void *ctx, int block_steps);
/* The same sink, but every block arrives as the TEXT MCout::output prints for its rows (src/mcout.cc:41-45; see
 * mcx_samples_text), formatted on the device: nbytes characters, no terminating 0, valid during the call.  One of the
 * two sinks at a time: setting one removes the other. */
typedef int (*mcx_text_sink_fn)(void *ctx, int first_step, int nsteps, const char *text, size_t nbytes);
int mcx_set_text_sink(mcx_engine *e, mcx_text_sink_fn fn, void *ctx, int block_steps);
/* inside a row sink's callback of a run with MCX_OPT_SINK_TEXT: the same block as text (valid during the callback) */
int mcx_sink_text(mcx_engine *e, const char **text, size_t *nbytes);

/* ---- options ------------------------------------------------------------------------------ */
enum {
  MCX_OPT_SAMPLES = 1,     /* 0 none, 1 keep every (step, chain) row in HBM (reference semantics,
                              src/mcpar.cc:177-182) [default 1] */
  MCX_OPT_ACCEPT_MASK = 2, /* 1: record one byte per (step, chain), burn-in and main [default 0] */
  MCX_OPT_FUSE = 3,        /* 1: multi-step fused kernel for local steps [default 1]; 0: one
                              propose/eval/accept kernel triple per step (same bits) */
  MCX_OPT_MAX_SEGMENT = 4, /* upper bound on steps per fused launch [default 256] */
  MCX_OPT_PROFILE = 5,     /* 1: bracket every kernel launch with HIP events (slower) */
  MCX_OPT_STREAM = 6,      /* value = hipStream_t to run on (default: engine-owned stream) */
  MCX_OPT_SAMPLE_STRIDE = 8, /* k >= 1: keep only main-loop steps with isamp % k == 0 in the sample store
                              (thinning; the reference keeps every step, k = 1) [default 1] */
  MCX_OPT_SPLIT_RNG = 9,   /* small-n mode: random numbers of 32-256 steps at a time from a separate, fully parallel
                              kernel, streamed into the step kernel (same bits).  -1 auto [default: when the
                              chains fill fewer than 640 wavefronts], 0 off, 1 on */
  MCX_OPT_PERSIST = 10,    /* small-n mode, one launch per stretch of local steps: the burn-in with its tuner, the start of
                              the main loop and the local main-loop steps run in ONE kernel whose owner wavefronts keep
                              the chains while the other wavefronts of each CU generate their random numbers into LDS
                              (same bits).  -1 auto [default: when the chains fill at most 8 wavefronts per CU and
                              SPLIT_RNG is not 0], 0 off, 1 on */
  MCX_OPT_EAGER_EXCHANGE = 7, /* 0 [default]: gather the latest sync-point snapshot only when a Murray step (or
                              the end of the run) will read it -- bit-identical to 1: gather at every sync
                              point like the reference (src/mcpar.cc:127-140), overlapped with compute */
  MCX_OPT_MEET_TIMEOUT_MS = 11, /* small-n mode: how long a tuner meeting of the one-launch kernel may wait for a
                              workgroup that is not resident (CU mask, partitioned device, foreign kernel) before
                              the launch is abandoned and the run repeated on the per-segment kernels (same bits;
                              mcx_counters.meet_timeouts counts it, MCX_VERBOSE=1 in the environment prints it; the engine
                              keeps to the per-segment kernels for 16 runs, then tries again) [default 50: a healthy
                              launch is 0.3-1.3 ms long and its meetings wait microseconds] */
  MCX_OPT_CULL = 13,       /* Murray sweeps: exclude, exactly, the Gaussians that are too far from all 128 chains of a
                              wavefront to matter (the chains are sorted spatially first; same bits).  Three screens:
                              (3) a lower bound of arg for every (chain, Gaussian) PAIR on the matrix cores -- a bf16
                              product with its rounding bounded rigorously, reduced over the 128 chains
                              (mcx_screen.hpp): leaves 6 % of C3-murray's pairs and 54 % of C5's; (1) boxes of four
                              coordinates around the 128 chains (33 % / 100 %); (2) one direction -- the chains' first
                              principal axis -- with a Cauchy-Schwarz bound along it (for targets stretched along a
                              line; on BASELINE's shapes it excludes less than the boxes).  -1 auto [default: the
                              per-pair bound, with np = 16 or 32, >= 4096 chains still rejected, >= 4096 Gaussians],
                              0 off, 1 / 2 / 3: that screen whenever np allows */
  MCX_OPT_BLOCKS_PER_LANE = 14, /* hot-path kernel: consecutive 4-parameter blocks of a chain held by one lane -- 1: one
                              (np/4 lanes per chain), 2 or 4: fewer lanes per chain, the per-chain work (acceptance test,
                              selects, counters) paid once per 2 / 4 blocks (same bits).  0 auto [default] */
  MCX_OPT_ASYNC_TAIL = 15,   /* sharded runs on the library's RCCL exchange: mcx_run returns while the run's LAST all-gather
                              (which nothing inside the run reads) is still in flight on its side stream; whatever looks
                              at the gathered slots next -- mcx_get_musigall, the next run's first gather or publish,
                              mcx_synchronize, mcx_destroy -- waits for it.  1 [default]: on a communicator the library made
                              itself (mcx_exchange_rccl_init); not on an adopted one, where the caller's own collectives
                              could overtake the pending gather on some ranks.  0: mcx_run always waits itself.  2: also with
                              an adopted communicator or a caller's exchange hook (whose MCX_XCHG_WAIT call then comes after
                              mcx_run has returned): call mcx_synchronize before anything else touches the communicator */
  MCX_OPT_SINK_TEXT = 16,    /* a row sink (mcx_set_sink) also gets every block as text: inside the callback, mcx_sink_text
                              returns the characters MCout::output would print for the block's rows [default 0] */
  MCX_OPT_ASYNC_RUN = 19,    /* 1: mcx_run returns as soon as the run is QUEUED (one shard, no sink / output hook / host likelihood,
                              no Murray step, samples > 0; other runs are waited for as ever).  Every other entry point -- the
                              getters, mcx_samples_*, mcx_get_counters, mcx_synchronize, mcx_set_option, mcx_destroy -- first
                              finishes it (waits, takes its counters, repeats a run whose tuner meeting was abandoned), so results
                              are what a synchronous run gives.  Calling mcx_run again before looking queues the next run behind
                              it (at most two in flight): launch and completion latency of back-to-back small jobs overlap the
                              jobs; the overtaken run's results are never seen (the next run overwrites them, as ever) and its
                              counters are not reported.  pinit / incov / the vlfunc are copied during the call as ever.
                              [default 0] */
  MCX_OPT_REFERENCE_CALLS = 20, /* 1: after every main-loop step call a HOST functor once per chain on (1, pvals_j, &y) and discard the
                              result, as the reference does (src/mcpar.cc:177-182): for functors whose side effects -- a call
                              counter, a cache -- must see the calls they see there.  Results do not change.  [default 0] */
  MCX_OPT_SELF_REPORT = 21, /* a run that ends with a launch of the one-launch small-n kernel (few chains, one shard, no sink or
                              output hook): 1: that launch's last workgroup writes the run's counters and a serial number to
                              pinned host memory and mcx_run spins on the word (for at most 1.5 ms, then sleeps in
                              hipStreamSynchronize as ever) -- nothing is queued behind the kernel: 6 us less per job of
                              0.3-0.5 ms waited for.  0: the counters come by a copy behind the kernel.  Same results.  [default 1] */
  MCX_OPT_MURRAY_OVERLAP = 18, /* Murray passes over many chains (np = 16 or 32, the per-pair screen): cut the Gaussians into this
                              many column chunks and screen chunk c + 1 (matrix cores, step stream) while chunk c is swept
                              (vector units, a side stream).  Same bits.  0 / 1: one screen, then one sweep */
  MCX_OPT_MEET_UNDER_GATHER = 17 /* small-n mode, sharded runs: may a launch with tuner meetings -- whose workgroups must all
                              be resident at once -- start while this engine's own last gather is still in flight
                              (MCX_OPT_ASYNC_TAIL)?  0: no, the step stream waits for the gather first: no cycle of a
                              half-resident launch, its GPU's gather kernel and a peer's can form.  1: yes, the next
                              run's burn-in runs under the gather; MCX_OPT_MEET_TIMEOUT_MS is then the net.  -1 auto
                              [default] = 0 */
  /* (12 is taken by a test hook that is not part of this header) */
};
int mcx_set_option(mcx_engine *e, int opt, int64_t value);

/* ---- results ------------------------------------------------------------------------------ */
typedef struct mcx_counters {
  uint64_t naccept_burn, naccept_main; /* accepted proposals (the reference's float naccept, :43-44) */
  uint64_t nsteps_burn, nsteps_main;
  uint64_t remote_steps, remote_passes; /* genRemote calls / rejection passes */
  uint64_t exchanges;
  uint64_t kernel_launches;
  uint64_t remote_pairs; /* (chain, Q_i) pairs of the Murray sweeps as the reference loops over them: sum over
                            passes of n_active * N, plus n * N per genRemote call for the cfac numerator
                            (src/mcpar.cc:367-395, 421-437) */
  uint64_t remote_pairs_evaluated; /* the pairs whose arg the sweep kernels actually started to accumulate: the
                            rest were excluded by an exact bound (their Q_i is exactly 0, or cannot lower the running
                            minimum) before any per-pair work.  The late passes over a few hundred chains sweep the
                            next four passes' proposals at once (same results, same pass count): proposals a chain
                            then did not need are counted here and not in remote_pairs, so without a screen the
                            ratio of the two is a little above 1 */
  uint64_t meet_timeouts; /* runs repeated on the per-segment kernels because a tuner meeting of the one-launch
                            small-n kernel was abandoned (MCX_OPT_MEET_TIMEOUT_MS) */
  /* ABI 4 */
  uint64_t meet_timeouts_total; /* the same, over the engine's life (meet_timeouts is the last run's) */
  uint64_t small_n_launches;    /* launches of the one-launch small-n kernel in the last run (0: per-segment kernels) */
  uint64_t small_n_blocks_per_lane; /* 4-parameter blocks per lane those launches ran with (0 when there were none) */
  uint64_t exchange_waits;      /* times the step stream was made to wait for a gather (src/mcpar.cc:127-140) that had
                                   been begun earlier: the last run's waits, including a wait for the run before's tail */
  uint64_t exchange_wait_ns;    /* device time the step stream spent in those waits (HIP events around each wait) */
} mcx_counters;
int mcx_get_counters(mcx_engine *e, mcx_counters *c);

int mcx_get_state(mcx_engine *e, float *pvals);       /* [nc*np]  MCPar::pvals   */
int mcx_get_loglike(mcx_engine *e, float *ly);        /* [nc]     MCPar::lylast  */
int mcx_get_mean(mcx_engine *e, float *mu);           /* [nc*np]  MCPar::mu      */
int mcx_get_var(mcx_engine *e, float *sig);           /* [nc*np]  MCPar::sig (population variance) */
int mcx_get_musigall(mcx_engine *e, float *musigall); /* [nshards*nc*np*2] MCPar::musigall */
/* Wait for everything the last mcx_run left in flight on the device (MCX_OPT_ASYNC_TAIL: the last all-gather of a
 * sharded run and the slot's final publish behind it).  A launcher that times runs calls it before stopping the clock. */
int mcx_synchronize(mcx_engine *e);
int mcx_get_chol(mcx_engine *e, float *cov);          /* [np*np]  MCPar::cov after tuning */
int mcx_get_accept_counts(mcx_engine *e, uint32_t *counts); /* [nc] per chain, burn + main */
int mcx_get_accept_mask(mcx_engine *e, uint8_t *mask);      /* [(nburn+nsamp)*nc] of the last run */
int mcx_get_tuner_trace(mcx_engine *e, float *scales, int maxn, int *n); /* cov[0] after each check */

/* sample store of the last run, in MCout row format: (np+1) columns, step-major then chain
 * (src/mcout.cc:129-145).  rows = steps*nc. */
int mcx_samples_steps(mcx_engine *e, int *nsteps);
int mcx_samples_copy(mcx_engine *e, int first_step, int nsteps, float *rows);
/* The same rows as the text MCout::output prints for them (src/mcout.cc:41-45: every field as `ostream << float` with
 * the stream defaults, i.e. printf("%g"), two blanks behind it, a newline behind a row's last column), formatted on the
 * device -- that conversion is 65-87 % of the reference's wall time.  *nbytes = bytes of the text (no terminating 0);
 * text == NULL, capacity == 0: the size only.  A C3-sized block of 25 steps is ~270 MB of text: ask step ranges. */
int mcx_samples_text(mcx_engine *e, int first_step, int nsteps, char *text, size_t capacity, size_t *nbytes);
/* the same for any rows on the host (ncol columns each, e.g. what a sink was given): uploaded, formatted, copied back */
int mcx_format_rows(const float *rows, size_t nrows, int ncol, char *text, size_t capacity, size_t *nbytes);
/* running maximum-likelihood sample of this shard (MCout::add's maxlval, src/mcout.cc:140-144) */
int mcx_samples_maxlike(mcx_engine *e, float *lmax, float *params);

/* ---- the schedule of one run (host logic only, no device needed) --------------------------
 * mcx_run cuts MCPar::run's two loops (src/mcpar.cc:55-97, 99-210) into device launches between the
 * events it knows in advance: tuner checks, output dumps, exchanges and -- because the local/remote
 * coin is counter-based -- Murray steps.  mcx_plan returns exactly the item list mcx_run executes. */
enum {
  MCX_PLAN_BURN_SEGMENT = 1, /* first, nsteps: consecutive burn-in steps in one launch */
  MCX_PLAN_TUNER = 2,        /* first = last step of the segment, nsteps, aux = 1 if the tuner decides here */
  MCX_PLAN_INIT_MOMENTS = 3,
  MCX_PLAN_OUTPUT = 4,       /* first = main-loop steps completed (output hook) */
  MCX_PLAN_PUBLISH = 5,      /* first = steps completed: write this shard's (mu, sig^2) slot */
  MCX_PLAN_GATHER_BEGIN = 6, /* exchange hook, MCX_XCHG_BEGIN */
  MCX_PLAN_GATHER_WAIT = 7,  /* exchange hook, MCX_XCHG_WAIT */
  MCX_PLAN_REMOTE_STEP = 8,  /* first = isamp of a Murray (genRemote) step */
  MCX_PLAN_MAIN_SEGMENT = 9, /* first, nsteps: consecutive local main-loop steps in one launch;
                                aux = local step after which the kernel snapshots the slot, or -1 */
  MCX_PLAN_SINK = 10         /* first = main-loop steps completed, nsteps = steps of the block that just ended:
                                hand the block to the sample sink (mcx_set_sink) */
};
typedef struct mcx_plan_item {
  int kind, first, nsteps, aux;
} mcx_plan_item;
/* tbase = steps consumed by earlier runs of the engine (0 for a fresh one).  items may be NULL to
 * query the count. */
int mcx_plan(int nsamp, int nburn, int sync, float pl, uint32_t seed, uint32_t tbase, int nshards,
             int eager, int fused, int max_segment, int has_output_hook, int sink_block_steps,
             mcx_plan_item *items, int max_items, int *nitems);

/* ---- profiling (MCX_OPT_PROFILE) ---------------------------------------------------------- */
enum { MCX_K_FUSED_BURN = 0, MCX_K_FUSED_MAIN, MCX_K_PROPOSE, MCX_K_EVAL, MCX_K_ACCEPT,
       MCX_K_REMOTE,       /* a whole genRemote call: draws, sweeps, decisions and the host's survivor counts */
       MCX_K_TUNER, MCX_K_MISC,
       MCX_K_REMOTE_SWEEP, /* the all-pairs sweep kernels alone (inside MCX_K_REMOTE); chain_steps = pairs */
       MCX_K_GEN_NORMALS,  /* small-n mode: the random-number generator kernel (inside MCX_K_FUSED_*) */
       MCX_K_RUN_SMALL,    /* small-n mode: k_run_small, burn-in and main-loop steps of one launch together */
       MCX_K_REMOTE_SCREEN,/* the per-pair screen's matrix-core kernel alone (inside MCX_K_REMOTE); chain_steps = pairs */
       MCX_K_COUNT = 12 };
typedef struct mcx_profile {
  double ms[MCX_K_COUNT];
  uint64_t launches[MCX_K_COUNT];
  uint64_t chain_steps[MCX_K_COUNT];
} mcx_profile;
int mcx_get_profile(mcx_engine *e, mcx_profile *p);

/* ---- helpers for exchange / output hooks written in a host language ------------------------ */
/* copies ordered after the work queued on `stream` (NULL = the default stream); both block until done */
int mcx_copy_to_host(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int mcx_copy_to_device(void *dst_dev, const void *src_host, size_t bytes, void *stream);

/* ---- misc --------------------------------------------------------------------------------- */
const char *mcx_last_error(void);
int mcx_abi_version(void);
int mcx_device_info(char *name, size_t namelen, int *cu_count, size_t *hbm_bytes);
int mcx_set_device(int device);
int mcx_device_count(int *n);
/* PCI bus id of the calling thread's current device ("0000:05:00.0"): lets a multi-process launcher see
 * whether two ranks share a GPU (RCCL refuses such a communicator) */
int mcx_device_pci_bus_id(char *buf, size_t len);
/* measured device-copy bandwidth of the current device: `reps` device-to-device copies of `bytes` bytes, timed with
 * HIP events; *gbps = (bytes read + bytes written) / time in GB/s.  What bench.py reports the HBM figures against
 * next to the nominal 8 TB/s. */
int mcx_debug_copy_bandwidth(size_t bytes, int reps, double *gbps);
/* host logic of the one-launch small-n kernel, for tests (no device): recorders yes/no and steps per phase for `own` owner
 * wavefronts per workgroup of lpc2 lanes per chain x bpl blocks per lane, and who generates what: tab[3][16][24] item
 * codes (0xffffffff ends a wavefront's list; kind << 14 | step pair << 4 | (owner, block)) */
int mcx_debug_persist_deal(int lpc2, int bpl, int own, int *rec, int *ksteps, uint32_t *tab, int max_words);
/* the per-pair screen of the Murray sweeps alone (mcx_screen.hpp; np = 16 or 32), for tests: nact chains x[nact][d], in
 * groups of 128 as given, against N Gaussians musig[N][d][2] = (mu, sig2); sums != 0: the sum sweeps' bound (176), else
 * the min-arg sweep's (chain j's own Gaussian is own0 + j).  masks[(N + 63) / 64][(nact + 127) / 128]: bit b of word w
 * for group g set = Gaussian 64 w + b may matter to some chain of the group. */
int mcx_debug_murray_screen(int d, int nact, int N, const float *x, const float *musig, int own0, int sums,
                            unsigned long long *masks);
/* device evaluation of the arithmetic primitives for bit-exactness tests:
 * what = 0 logf(bits), 1 expf(bits), 2 sin(2 pi w/2^32), 3 cos(...), 4 u24, 5 uopen,
 * 6 philox word 0 of ctr=(w,0,0,0) key=(0,0), 7 the kernels' lean sqrt, 8 IEEE sqrtf,
 * 9/10 packed expf lane 0/1, 11 packed logf, 12/13 packed/scalar sincos hash, 14/15 scalar/packed log of
 * the acceptance draw */
int mcx_debug_numerics(int what, int n, const uint32_t *in, uint32_t *out_bits);
/* number of float bit patterns in [lo_bits, hi_bits) where the kernels' lean sqrt (valid for +-0 and
 * positive normal floats) differs from IEEE sqrtf, and the smallest such pattern */
int mcx_debug_sqrt_sweep(uint32_t lo_bits, uint32_t hi_bits, uint64_t *nbad, uint32_t *first_bad);
/* normals of stream `stream`, counter (t, g0+i, a, q) for i < n: out[n*4] */
int mcx_debug_normals(uint32_t seed, uint32_t stream, uint32_t t, uint32_t g0, uint32_t a,
                      uint32_t q, int n, float *out);

#ifdef __cplusplus
}
#endif
#endif /* MCX_H_ */
