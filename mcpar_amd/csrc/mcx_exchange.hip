// mcx_exchange.hip -- the inter-shard exchange of the (mu, sig^2) slots (src/mcpar.cc:127-140): begin / wait / publish
// as the plan executor calls them, the run's last gather left in flight (finish_tail), and the exchange the library
// ships: an in-place ncclAllGather over RCCL, loaded with dlopen on first use.
#include "mcx_engine_internal.hpp"


constexpr size_t XW_MAX = 256;  // waits of one run that are timed (a job has one per Murray step at most)

int exchange_wait(mcx_engine *e)
{
  if (!e->xchg_pending) return MCX_OK;
  MCXCHK(meet_release(e, false));
  e->xchg_pending = false;
  // the wait as the step stream sees it: an event on either side (collected by xwait_collect once the stream is idle)
  const bool timed = e->xw_used < XW_MAX;
  if (timed && e->xw_used == e->xw_pool.size()) {
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    e->xw_pool.emplace_back(a, b);
  }
  if (timed) HIPCHK(hipEventRecord(e->xw_pool[e->xw_used].first, e->stream));
  if (e->xfn(e->xctx, MCX_XCHG_WAIT, e->musigall.p, 2 * (size_t)e->ntot, e->rank, e->size, e->stream) != 0)
    return fail(MCX_ERR_EXCHANGE, "exchange hook failed in WAIT");
  if (timed) HIPCHK(hipEventRecord(e->xw_pool[e->xw_used++].second, e->stream));
  e->cnt.exchange_waits++;
  return MCX_OK;
}

// the step stream is idle: add up what the timed waits took
void xwait_collect(mcx_engine *e)
{
  for (size_t i = 0; i < e->xw_used; ++i) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, e->xw_pool[i].first, e->xw_pool[i].second) == hipSuccess && ms > 0.0f)
      e->cnt.exchange_wait_ns += (uint64_t)((double)ms * 1e6);
  }
  e->xw_used = 0;
}

// write this shard's slot from the resident moments after `steps_done` main-loop steps
int publish(mcx_engine *e, int steps_done)
{
  if (steps_done <= 0 || e->published_steps == steps_done) return MCX_OK;
  MCXCHK(exchange_wait(e));  // an in-flight gather still reads the slot
  ProfScope ps(e, MCX_K_MISC, 0);
  hipLaunchKernelGGL(k_publish, dim3(nblocks((size_t)e->ntot)), dim3(BLOCK), 0, e->stream, e->mu.p,
                     e->psum2.p, e->musigall.p + 2 * (size_t)e->rank * e->ntot, (size_t)e->ntot,
                     1.0f / (float)steps_done);
  HIPCHK(hipGetLastError());
  e->published_steps = steps_done;
  return MCX_OK;
}

// The last all-gather of a sharded run feeds nothing inside that run: it leaves the other shards' slots as of the
// last sync point for whoever looks at musigall next.  When the library's own RCCL exchange carries it (a side stream,
// nothing for the host to do), mcx_run does not wait for it: the gather runs on under the caller's next steps -- e.g.
// the burn-in of the next run, which never touches musigall -- and whatever does touch it (mcx_get_musigall, the next
// gather or publish, mcx_synchronize, mcx_destroy) waits first.  The slot's final publish (own moments after the last
// step) cannot precede the gather that still reads the slot, so it waits with it.
int finish_tail(mcx_engine *e)
{
  if (!e->tail_publish) return MCX_OK;
  const int steps = e->tail_publish;
  e->tail_publish = 0;
  MCXCHK(exchange_wait(e));
  MCXCHK(publish(e, steps));
  HIPCHK(hipStreamSynchronize(e->stream));
  xwait_collect(e);
  return MCX_OK;
}

int exchange_begin(mcx_engine *e)
{
  MCXCHK(meet_release(e, false));  // a hook may wait for other engines: never while holding the GPU's meeting lock
  MCXCHK(exchange_wait(e));
  if (e->xfn(e->xctx, MCX_XCHG_BEGIN, e->musigall.p, 2 * (size_t)e->ntot, e->rank, e->size, e->stream) != 0)
    return fail(MCX_ERR_EXCHANGE, "exchange hook failed in BEGIN");
  e->xchg_pending = true;
  e->cnt.exchanges++;
  return MCX_OK;
}

// ---------------------------------------------------------------------------------------------
// Native RCCL exchange: the MPI_Allgather(MPI_IN_PLACE, ..., musigall, 2*ntot, MPI_FLOAT) of
// src/mcpar.cc:127-140 as an in-place ncclAllGather over xGMI.  One process (or thread) per GPU, one
// communicator rank per shard, slot layout of src/mcpar.cc:206 (rank r owns floats [r*2*ntot, (r+1)*2*ntot)).
// BEGIN enqueues the collective on a side stream behind everything already queued on the engine's
// stream; WAIT makes the engine's stream wait for it -- so under the reference's own schedule
// (MCX_OPT_EAGER_EXCHANGE) the gather overlaps the next segment of local steps.  librccl.so.1 is loaded
// on first use: single-GPU users never need it.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
RcclApi g_rccl;

bool rccl_load()
{
  if (g_rccl.handle) return true;
  if (!g_rccl.why.empty()) return false;
  // if another RCCL is already in the process (e.g. the one PyTorch-ROCm ships) the soname resolves to it
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  // MCX_RCCL_LIB: this library and no other (a site's own RCCL build; tests/cpp/rccl_stub.hip) -- no quiet fallback to
  // the system's when it does not load
  const char *forced = std::getenv("MCX_RCCL_LIB");
  if (forced && *forced) {
    h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      const char *de = dlerror();
      g_rccl.why = std::string("MCX_RCCL_LIB=") + forced + " not loadable: " + (de ? de : "?");
      return false;
    }
  }
  for (const char *n : names)
    if (h || (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) {
    const char *de = dlerror();
    g_rccl.why = std::string("librccl.so.1 not loadable: ") + (de ? de : "?");
    return false;
  }
  bool ok = true;
  auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) { ok = false; g_rccl.why = std::string("missing RCCL symbol ") + n; } return p; };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
  g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))sym("ncclCommUserRank");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  if (!ok) return false;
  g_rccl.handle = h;
  return true;
}

#define NCCLCHK(expr)                                                                          \
  do {                                                                                         \
    ncclResult_t r_ = (expr);                                                                  \
    if (r_ != ncclSuccess)                                                                     \
      return fail(MCX_ERR_EXCHANGE, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));        \
  } while (0)

// the exchange hook itself: mcx_exchange_fn with ctx = the engine
int rccl_exchange(void *ctx, int phase, void *dev, size_t slot, int shard, int nshards, void *stream)
{
  mcx_engine *e = static_cast<mcx_engine *>(ctx);
  hipStream_t st = (hipStream_t)stream;
  (void)nshards;
  if (phase == MCX_XCHG_BEGIN) {
    HIPCHK(hipEventRecord(e->xready, st));
    HIPCHK(hipStreamWaitEvent(e->xstream, e->xready, 0));
    float *base = static_cast<float *>(dev);
    NCCLCHK(g_rccl.AllGather(base + slot * (size_t)shard, base, slot, ncclFloat, e->xcomm, e->xstream));  // in place
    HIPCHK(hipEventRecord(e->xdone, e->xstream));
  } else {
    HIPCHK(hipStreamWaitEvent(st, e->xdone, 0));
  }
  return 0;
}

int rccl_install(mcx_engine *e, ncclComm_t comm, bool owned)
{
  int cnt = 0, rk = -1;
  NCCLCHK(g_rccl.CommCount(comm, &cnt));
  NCCLCHK(g_rccl.CommUserRank(comm, &rk));
  if (cnt != e->size || rk != e->rank)
    return fail(MCX_ERR_INVALID, "communicator is rank %d of %d but the engine is shard %d of %d", rk, cnt, e->rank, e->size);
  if (!e->xstream) HIPCHK(hipStreamCreateWithFlags(&e->xstream, hipStreamNonBlocking));
  if (!e->xready) HIPCHK(hipEventCreateWithFlags(&e->xready, hipEventDisableTiming));
  if (!e->xdone) HIPCHK(hipEventCreateWithFlags(&e->xdone, hipEventDisableTiming));
  e->xcomm = comm;
  e->xcomm_owned = owned;
  e->xfn = rccl_exchange;
  e->xctx = e;
  return MCX_OK;
}
}  // namespace

extern "C" int mcx_rccl_available(void)
{
  if (rccl_load()) return 1;
  (void)fail(MCX_ERR_EXCHANGE, "%s", g_rccl.why.c_str());
  return 0;
}

extern "C" int mcx_rccl_unique_id(void *id)
{
  if (!id) return fail(MCX_ERR_INVALID, "id is NULL");
  if (!rccl_load()) return fail(MCX_ERR_EXCHANGE, "%s", g_rccl.why.c_str());
  static_assert(sizeof(ncclUniqueId) == MCX_RCCL_ID_BYTES, "MCX_RCCL_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
  ncclUniqueId u;
  NCCLCHK(g_rccl.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return MCX_OK;
}

extern "C" int mcx_exchange_rccl_init(mcx_engine *e, const void *id)
{
  MCXCHK(enter(e));
  if (!id) return fail(MCX_ERR_INVALID, "id is NULL");
  if (!rccl_load()) return fail(MCX_ERR_EXCHANGE, "%s", g_rccl.why.c_str());
  MCXCHK(mcx_exchange_rccl_destroy(e));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  NCCLCHK(g_rccl.CommInitRank(&comm, e->size, u, e->rank));  // collective over the nshards engines
  const int rc = rccl_install(e, comm, true);
  if (rc != MCX_OK) (void)g_rccl.CommDestroy(comm);
  return rc;
}

extern "C" int mcx_exchange_rccl_adopt(mcx_engine *e, void *nccl_comm)
{
  MCXCHK(enter(e));
  if (!nccl_comm) return fail(MCX_ERR_INVALID, "communicator is NULL");
  if (!rccl_load()) return fail(MCX_ERR_EXCHANGE, "%s", g_rccl.why.c_str());
  MCXCHK(mcx_exchange_rccl_destroy(e));
  return rccl_install(e, (ncclComm_t)nccl_comm, false);
}

extern "C" int mcx_exchange_rccl_destroy(mcx_engine *e)
{
  if (!e) return fail(MCX_ERR_INVALID, "engine is NULL");
  if (e->xcomm) {
    (void)hipSetDevice(e->device);
    (void)finish_tail(e);
    if (e->xstream) (void)hipStreamSynchronize(e->xstream);
    if (e->xcomm_owned && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(e->xcomm);
    if (e->xfn == rccl_exchange) { e->xfn = nullptr; e->xctx = nullptr; }
    e->xcomm = nullptr;
    e->xcomm_owned = false;
    e->xchg_pending = false;
  }
  if (e->xready) { (void)hipEventDestroy(e->xready); e->xready = nullptr; }
  if (e->xdone) { (void)hipEventDestroy(e->xdone); e->xdone = nullptr; }
  if (e->xstream) { (void)hipStreamDestroy(e->xstream); e->xstream = nullptr; }
  return MCX_OK;
}

extern "C" int mcx_exchange_rccl_info(mcx_engine *e, int *nranks, int *rank)
{
  if (!e || !nranks || !rank) return fail(MCX_ERR_INVALID, "bad arguments");
  if (!e->xcomm) return fail(MCX_ERR_EXCHANGE, "no RCCL exchange installed");
  NCCLCHK(g_rccl.CommCount(e->xcomm, nranks));
  NCCLCHK(g_rccl.CommUserRank(e->xcomm, rank));
  return MCX_OK;
}

// One exchange right now (publish is the caller's business): BEGIN + WAIT + drain.  Lets a test (or a
// start-up self-check) push the installed hook through the device without running a job.
extern "C" int mcx_debug_exchange(mcx_engine *e)
{
  MCXCHK(enter(e));
  if (!e->xfn) return fail(MCX_ERR_EXCHANGE, "no exchange hook installed");
  MCXCHK(finish_tail(e));
  MCXCHK(exchange_begin(e));
  MCXCHK(exchange_wait(e));
  HIPCHK(hipStreamSynchronize(e->stream));
  return MCX_OK;
}

static __global__ void k_fill(float *p, size_t n, float v)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

extern "C" int mcx_debug_fill_slot(mcx_engine *e, float value)
{
  MCXCHK(enter(e));
  MCXCHK(finish_tail(e));
  MCXCHK(exchange_wait(e));  // (a gather in flight still reads the slot)
  const size_t slot = 2 * (size_t)e->ntot;
  hipLaunchKernelGGL(k_fill, dim3(nblocks(slot)), dim3(BLOCK), 0, e->stream, e->musigall.p + slot * (size_t)e->rank, slot, value);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return MCX_OK;
}


// May mcx_run return with the run's last gather still in flight (MCX_OPT_ASYNC_TAIL)?  2: with any exchange; 1 (default):
// with the library's RCCL exchange on a communicator of its OWN -- on one the caller handed over (mcx_exchange_rccl_adopt)
// the caller may issue collectives of its own next, and their order against a gather still pending here could differ
// from rank to rank: there mcx_run waits, as with a hook.
bool exchange_tail_may_stay_in_flight(const mcx_engine *e)
{
  return e->opt_async_tail == 2 || (e->opt_async_tail == 1 && e->xfn == rccl_exchange && e->xcomm_owned);
}
