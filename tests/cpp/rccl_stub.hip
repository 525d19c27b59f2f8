// rccl_stub.hip -- TEST shim, never shipped: the seven RCCL entry points libmcx's exchange uses
// (mcx_exchange.hip: rccl_load), for "ranks" that are THREADS of one process on ONE GPU.  Loaded through
// MCX_RCCL_LIB=<this .so>, it lets the library's own hook (BEGIN / WAIT events, side stream, ASYNC_TAIL,
// destroy-while-pending: mcx_exchange.hip:142-157) run with a real peer on a one-GPU box, where RCCL itself
// refuses two ranks on one device.  What it replaces there is MPI_Allgather of src/mcpar.cc:127-140.
//
// ncclAllGather(send, recv, count, ..., stream): every rank posts (send pointer, an event recorded on its
// stream) under a sequence number and waits ON THE HOST until all ranks of the communicator have posted
// (bounded: RCCL_STUB_TIMEOUT_MS, default 20 s -> ncclSystemError); then its stream waits for each peer's
// event and copies the peer's `count` elements into recv + peer*count.  A second host meeting hands every
// rank the peers' copy-done events, which its stream waits for before anything later may overwrite the
// slot the peers were reading.  Ranks in OTHER processes never arrive: ncclCommInitRank then times out --
// which is how the tests rehearse a communicator that does not come up.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
}

namespace {
struct Group {
  int nranks = 0;
  int arrived = 0;  // ranks that called ncclCommInitRank
  std::mutex m;
  std::condition_variable cv;
  // one collective at a time per group (the engines issue them in the same order): what each rank posted
  struct Round {
    std::vector<const void *> send;
    std::vector<hipEvent_t> ready, copied;
    int posted = 0, posted2 = 0, left = 0;
  };
  std::map<uint64_t, Round> rounds;
  uint64_t calls_total = 0;
  std::vector<hipEvent_t> garbage;  // events of finished rounds: destroyed when the group's last communicator goes
  int comms_alive = 0;
};

struct Comm {
  Group *g;
  int rank;
  uint64_t seq = 0;
};

std::mutex g_table_m;
std::map<std::string, Group *> g_table;
uint64_t g_next_id = 1;

int timeout_ms()
{
  const char *s = std::getenv("RCCL_STUB_TIMEOUT_MS");
  return s ? std::atoi(s) : 20000;
}

template <class Pred> bool wait_for(Group *g, std::unique_lock<std::mutex> &lk, Pred p)
{
  return g->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms()), p);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
  std::lock_guard<std::mutex> lk(g_table_m);
  std::memset(id, 0, sizeof *id);
  std::snprintf(id->internal, sizeof id->internal, "rccl-stub-%llu", (unsigned long long)g_next_id++);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(Comm **comm, int nranks, ncclUniqueId id, int rank)
{
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Group *g;
  {
    std::lock_guard<std::mutex> lk(g_table_m);
    std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
    auto it = g_table.find(key);
    if (it == g_table.end()) it = g_table.emplace(key, new Group).first;
    g = it->second;
  }
  std::unique_lock<std::mutex> lk(g->m);
  if (g->nranks == 0) g->nranks = nranks;
  if (g->nranks != nranks) return ncclInvalidArgument;
  g->arrived++;
  g->cv.notify_all();
  if (!wait_for(g, lk, [&] { return g->arrived >= g->nranks; })) {
    g->arrived--;
    return ncclSystemError;  // the other ranks never came (e.g. they live in other processes)
  }
  g->comms_alive++;
  *comm = new Comm{g, rank};
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(Comm *c)
{
  Group *g = c->g;
  std::vector<hipEvent_t> ev;
  {
    std::lock_guard<std::mutex> lk(g->m);
    if (--g->comms_alive == 0) ev.swap(g->garbage);
  }
  for (hipEvent_t e : ev) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
  delete c;  // groups stay in the table for the life of the process: a test shim
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const Comm *c, int *n) { *n = c->g->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const Comm *c, int *r) { *r = c->rank; return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r)
{
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "rccl stub: the other ranks did not arrive in time";
    case ncclInvalidArgument: return "rccl stub: invalid argument";
    default: return "rccl stub: HIP error";
  }
}

// calls seen by the whole process (the tests check the hook really came through here)
uint64_t rccl_stub_allgather_calls(void)
{
  std::lock_guard<std::mutex> lk(g_table_m);
  uint64_t n = 0;
  for (auto &kv : g_table) {
    std::lock_guard<std::mutex> lk2(kv.second->m);
    n += kv.second->calls_total;
  }
  return n;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, Comm *c, hipStream_t st)
{
  if (dt != ncclFloat) return ncclInvalidArgument;
  Group *g = c->g;
  const int n = g->nranks, me = c->rank;
  const uint64_t seq = c->seq++;
  const size_t bytes = count * sizeof(float);
  hipEvent_t ready = nullptr, copied = nullptr;
  if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  if (hipEventCreateWithFlags(&copied, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  if (hipEventRecord(ready, st) != hipSuccess) return ncclUnhandledCudaError;  // my slot is final at this point of my stream
  std::vector<const void *> psend;
  std::vector<hipEvent_t> pready, pcopied;
  {
    std::unique_lock<std::mutex> lk(g->m);
    g->calls_total++;
    Group::Round &r = g->rounds[seq];
    if (r.send.empty()) { r.send.assign(n, nullptr); r.ready.assign(n, nullptr); r.copied.assign(n, nullptr); r.left = n; }
    r.send[me] = send;
    r.ready[me] = ready;
    r.posted++;
    g->cv.notify_all();
    if (!wait_for(g, lk, [&] { return g->rounds[seq].posted >= n; })) return ncclSystemError;
    psend = g->rounds[seq].send;
    pready = g->rounds[seq].ready;
  }
  for (int p = 0; p < n; ++p) {
    char *dst = static_cast<char *>(recv) + (size_t)p * bytes;
    if (p == me) {
      if (dst != send && hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
      continue;
    }
    if (hipStreamWaitEvent(st, pready[p], 0) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(dst, psend[p], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
  }
  if (hipEventRecord(copied, st) != hipSuccess) return ncclUnhandledCudaError;
  {
    std::unique_lock<std::mutex> lk(g->m);
    Group::Round &r = g->rounds[seq];
    r.copied[me] = copied;
    r.posted2++;
    g->cv.notify_all();
    if (!wait_for(g, lk, [&] { return g->rounds[seq].posted2 >= n; })) return ncclSystemError;
    pcopied = g->rounds[seq].copied;
  }
  // nothing later on my stream may touch my slot before every peer has read it
  for (int p = 0; p < n; ++p)
    if (p != me && hipStreamWaitEvent(st, pcopied[p], 0) != hipSuccess) return ncclUnhandledCudaError;
  {
    std::unique_lock<std::mutex> lk(g->m);
    Group::Round &r = g->rounds[seq];
    if (--r.left == 0) {  // every rank has enqueued its waits: nobody looks at this round again
      for (int p = 0; p < n; ++p) { g->garbage.push_back(r.ready[p]); g->garbage.push_back(r.copied[p]); }
      g->rounds.erase(seq);
    }
  }
  return ncclSuccess;
}
}  // extern "C"
