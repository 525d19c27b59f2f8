// facade.cc -- MCPar / MCout with the reference's public interface (src/mcpar.hh, src/mcout.hh),
// implemented over the C ABI of libmcx.  Host-only C++: no HIP types cross this file.
//
// Error behaviour follows the reference: constructor argument errors throw a string literal
// (src/mcpar.cc:268), device/runtime failures print and abort() (VSL_CALL_CHK, src/mcpar.hh:93),
// a sample store that does not fit prints "Unable to allocate space for output samples." and exits
// with status 2 (src/mcpar.cc:34-40), run() returns 0 (src/mcpar.cc:213).
#include "../../include/mcpar/mcpar.hh"
#include "../../include/mcpar/mcutil.hh"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cassert>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <locale>
#include <new>
#include <sstream>
#include <thread>
#include <vector>

#include "fmt_g6.hpp"

const float MCPar::FPEPS = 1.0e-14f;

static void die(const char *where)
{
  std::cerr << "mcpar: " << where << ": " << mcx_last_error() << std::endl;
  abort();
}

// ---------------------------------------------------------------------------------------------
// MCout: behaviour of src/mcout.cc
// ---------------------------------------------------------------------------------------------
MCout::MCout(int np, std::ostream *aoutstream, MPI_Comm acomm)
    : nparam_(np), width_(np + 1), fill_(0), flushed_(0), stored_rows_(0), capacity_rows_(0),
      best_l_(-std::numeric_limits<float>::infinity()), best_p_(static_cast<size_t>(np), 0.0f), sink_(0), binary_(false), text_only_(false),
      text_fd_(-1), text_pos_(0)
{
#ifdef MCX_WITH_MPI
  if (MPI_Comm_dup(acomm, &comm_) != MPI_SUCCESS) {
    std::cerr << "MCout: unable to duplicate input communicator in constructor." << std::endl;
    MPI_Abort(acomm, 1);
  }
#else
  comm_ = acomm;
#endif
  MPI_Comm_rank(comm_, &rank_);
  MPI_Comm_size(comm_, &nranks_);
  if (rank_ == 0) sink_ = aoutstream;  // all text goes through rank 0, like the reference
}

void MCout::note_row(const float *row)
{
  if (row[nparam_] > best_l_) {  // first strict maximum wins (src/mcout.cc:140-144)
    best_l_ = row[nparam_];
    best_p_.assign(row, row + nparam_);
  }
}

bool MCout::all_ranks_agree(bool mine)
{
  int ok = mine ? 1 : 0;
#ifdef MCX_WITH_MPI
  if (nranks_ > 1) {
    int all = 0;
    if (MPI_Allreduce(&ok, &all, 1, MPI_INT, MPI_MIN, comm_) != MPI_SUCCESS) MPI_Abort(MPI_COMM_WORLD, 1);
    ok = all;
  }
#endif
  return ok != 0;
}

bool MCout::text_file(const char *path)
{
  if (text_fd_ >= 0) close(text_fd_);
  text_fd_ = -1;
  text_pos_ = 0;
  if (!path || !*path) return true;
  // rank 0 creates (and empties) the file, then everybody opens it for writing at offsets of its own
  int fd = -1;
  if (rank_ == 0) fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
  if (!all_ranks_agree(rank_ != 0 || fd >= 0)) {
    if (fd >= 0) close(fd);
    return false;
  }
  if (rank_ != 0) fd = open(path, O_WRONLY | O_CLOEXEC);
  if (!all_ranks_agree(fd >= 0)) {
    if (fd >= 0) close(fd);
    return false;
  }
  text_fd_ = fd;
  return true;
}

static bool write_all_at(int fd, const char *p, size_t n, unsigned long long at)
{
  while (n > 0) {
    const ssize_t w = pwrite(fd, p, std::min<size_t>(n, size_t(1) << 30), static_cast<off_t>(at));
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += w;
    n -= static_cast<size_t>(w);
    at += static_cast<unsigned long long>(w);
  }
  return true;
}

// COLLECTIVE with several ranks: the ranks' texts of one block reach the stream in rank order -- the row order of a dump
// (src/mcout.cc:62-69: rank-major, then step, then chain).  Two ways: through rank 0's stream, each rank's share sent
// to it in pieces of at most 1 GiB (an MPI count is an int: a 65 536-chain rank's block passes 2^31 characters from
// nsamp ~ 1900 on); or, text_file(), every rank writing its share itself where an exclusive scan of the sizes puts it.
void MCout::write_text(const char *text, size_t nbytes)
{
  if (text_fd_ >= 0) {
    unsigned long long mine = nbytes, before = 0, total = nbytes;
    (void)mine;
#ifdef MCX_WITH_MPI
    if (nranks_ > 1) {
      if (MPI_Exscan(&mine, &before, 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, comm_) != MPI_SUCCESS ||
          MPI_Allreduce(&mine, &total, 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, comm_) != MPI_SUCCESS)
        MPI_Abort(MPI_COMM_WORLD, 1);
      if (rank_ == 0) before = 0;  // (MPI_Exscan leaves rank 0's result undefined)
    }
#endif
    if (nbytes && !write_all_at(text_fd_, text, nbytes, text_pos_ + before)) {
      std::cerr << "MCout::write_text: writing the sample text failed on rank " << rank_ << ".  Aborting.\n";
#ifdef MCX_WITH_MPI
      MPI_Abort(MPI_COMM_WORLD, 1);
#endif
      abort();
    }
    text_pos_ += total;
    return;
  }
#ifdef MCX_WITH_MPI
  if (nranks_ > 1) {
    const size_t piece = size_t(1) << 30;
    if (rank_ != 0) {
      long long mine = static_cast<long long>(nbytes);
      bool ok = MPI_Send(&mine, 1, MPI_LONG_LONG, 0, 7101, comm_) == MPI_SUCCESS;
      for (size_t at = 0; ok && at < nbytes; at += piece)
        ok = MPI_Send(const_cast<char *>(text + at), static_cast<int>(std::min(piece, nbytes - at)), MPI_CHAR, 0, 7102, comm_) == MPI_SUCCESS;
      if (!ok) {
        std::cerr << "Unable to send output text.  Aborting.\n";
        MPI_Abort(MPI_COMM_WORLD, 1);
      }
      return;
    }
    if (sink_ && nbytes) sink_->write(text, static_cast<std::streamsize>(nbytes));
    std::vector<char> theirs;
    for (int r = 1; r < nranks_; ++r) {
      long long n = 0;
      if (MPI_Recv(&n, 1, MPI_LONG_LONG, r, 7101, comm_, MPI_STATUS_IGNORE) != MPI_SUCCESS) MPI_Abort(MPI_COMM_WORLD, 1);
      theirs.resize(std::min(static_cast<size_t>(n), piece) + 1);
      for (size_t at = 0; at < static_cast<size_t>(n); at += piece) {
        const size_t len = std::min(piece, static_cast<size_t>(n) - at);
        if (MPI_Recv(theirs.data(), static_cast<int>(len), MPI_CHAR, r, 7102, comm_, MPI_STATUS_IGNORE) != MPI_SUCCESS)
          MPI_Abort(MPI_COMM_WORLD, 1);
        if (sink_) sink_->write(theirs.data(), static_cast<std::streamsize>(len));
      }
    }
    return;
  }
#endif
  if (rank_ == 0 && sink_ && nbytes) sink_->write(text, static_cast<std::streamsize>(nbytes));
}

static bool stream_prints_like_printf(const std::ostream &os)
{
  const std::ios_base::fmtflags special =
      std::ios_base::floatfield | std::ios_base::showpoint | std::ios_base::showpos | std::ios_base::uppercase;
  static const bool through_the_stream = getenv("MCPAR_TEXT") && !strcmp(getenv("MCPAR_TEXT"), "stream");  // (escape hatch)
  return !through_the_stream && os.precision() == 6 && !(os.flags() & special) && os.width() == 0 &&
         os.getloc() == std::locale::classic();
}

bool MCout::prints_plain_text(void) const
{
  if (binary_ || text_only_ || fill_ != flushed_) return false;
  // (the stream lives on rank 0; the others follow its answer in MCPar::run)
  return rank_ != 0 || text_fd_ >= 0 || (sink_ && stream_prints_like_printf(*sink_));
}

void MCout::note_best(float lval, const float *params)
{
  if (lval > best_l_) {
    best_l_ = lval;
    best_p_.assign(params, params + nparam_);
  }
}

void MCout::add(const float *pv, float lval)
{
  assert(fill_ + width_ <= rows_.size());
  float *dst = &rows_[fill_];
  std::memcpy(dst, pv, sizeof(float) * nparam_);
  dst[nparam_] = lval;
  note_row(dst);
  fill_ += width_;
  ++stored_rows_;
}

void MCout::add_rows(const float *rows, size_t nrows, bool track_best)
{
  assert(fill_ + nrows * width_ <= rows_.size());
  const size_t bytes = nrows * width_ * sizeof(float);
  char *dst = reinterpret_cast<char *>(&rows_[fill_]);
  const char *src = reinterpret_cast<const char *>(rows);
  if (bytes < (size_t(64) << 20)) {
    std::memcpy(dst, src, bytes);
  } else {  // hundreds of megabytes into pages that are touched for the first time: one thread faults them in at ~3 GB/s
    const unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 8u));
    const size_t per = ((bytes / nt) + 4095) & ~size_t(4095);
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) {
      const size_t a = std::min(bytes, t * per), b = std::min(bytes, a + per);
      if (b > a) th.emplace_back([=] { std::memcpy(dst + a, src + a, b - a); });
    }
    std::memcpy(dst, src, std::min(bytes, per));
    for (auto &x : th) x.join();
  }
  if (track_best)
    for (size_t r = 0; r < nrows; ++r) note_row(rows + r * width_);
  fill_ += nrows * width_;
  stored_rows_ += static_cast<int>(nrows);
}

// rows [r0, r0 + batch) of `all` (w columns) as the reference's text -- every field followed by two blanks, "%g" -- cut
// into pieces for the host's threads: text[t][0 .. used[t]) for t < the returned count, in row order
static unsigned format_batch(const float *all, size_t r0, size_t batch, size_t w, unsigned nthreads,
                             std::vector<std::vector<char> > &text, std::vector<size_t> &used)
{
  const unsigned nt = batch * w < (size_t(1) << 16) ? 1u : nthreads;  // (small dumps: not worth the threads)
  const size_t per = (batch + nt - 1) / nt;
  auto work = [&](unsigned t) {
    const size_t a = std::min(batch, t * per), b = std::min(batch, a + per);
    text[t].resize((b - a) * (w * 18 + 1) + 16);
    char *p = text[t].data();
    for (size_t r = r0 + a; r < r0 + b; ++r) {
      const float *row = all + r * w;
      for (size_t c = 0; c < w; ++c) {
        p = fmtg6::append(p, row[c]);
        *p++ = ' ';
        *p++ = ' ';
      }
      *p++ = '\n';
    }
    used[t] = static_cast<size_t>(p - text[t].data());
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  return nt;
}

// text_file(): a dump goes to the FILE, whoever asks for it -- MCPar::run's row path when a block's GPU text is not to be
// had, a driver's own output() call, binary() rows too.  Nothing is gathered: every rank turns its own fresh rows into
// bytes and write_text() puts them where the ranks' sizes say, i.e. in the rank-major row order of a gathered dump
// (src/mcout.cc:62-69).  COLLECTIVE like the gather it replaces; the ranks hold equal numbers of rows (src/mcpar.cc:225),
// hence make the same number of write_text() calls.  The text is printf("%g")'s -- the stream, whatever state a caller
// left it in, is not involved.
void MCout::output_to_file()
{
  const size_t count = fill_ > flushed_ ? fill_ - flushed_ : 0;
  const float *all = count ? &rows_[flushed_] : 0;
  flushed_ = fill_;
  if (binary_) {
    write_text(reinterpret_cast<const char *>(all), count * sizeof(float));
    return;
  }
  const size_t w = static_cast<size_t>(width_), nrows = count / w;
  const size_t piece_rows = std::max<size_t>(1, (size_t(1) << 21) / w);
  const unsigned nthreads = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
  std::vector<std::vector<char> > text(nthreads);
  std::vector<size_t> used(nthreads, 0);
  std::vector<char> joined;
  if (nrows == 0) write_text(0, 0);  // (still collective)
  for (size_t r0 = 0; r0 < nrows; r0 += piece_rows * nthreads) {
    const size_t batch = std::min(nrows - r0, piece_rows * nthreads);
    const unsigned nt = format_batch(all, r0, batch, w, nthreads, text, used);
    size_t total = 0;
    for (unsigned t = 0; t < nt; ++t) total += used[t];
    joined.resize(total);
    size_t at = 0;
    for (unsigned t = 0; t < nt; ++t) {
      std::memcpy(joined.data() + at, text[t].data(), used[t]);
      at += used[t];
    }
    write_text(joined.data(), total);
  }
}

// text format of the reference: every field followed by two blanks, stream-default precision
void MCout::output()
{
  if (text_fd_ >= 0) {
    output_to_file();
    return;
  }
  size_t count = 0;
  float *owned = 0;
  const float *all = 0;
  if (nranks_ == 1) {  // nothing to gather: print from the store itself (collect() would copy the rows first)
    count = fill_ > flushed_ ? fill_ - flushed_ : 0;
    if (count) all = &rows_[flushed_];
    flushed_ = fill_;
  } else {
    owned = collect(&count);
    all = owned;
  }
  if (rank_ != 0 || count == 0) return;
  std::ostream &os = *sink_;
  if (binary_) {
    os.write(reinterpret_cast<const char *>(all), (std::streamsize)(count * sizeof(float)));
    delete[] owned;
    return;
  }
  // The reference prints every number through the stream (src/mcout.cc:41-45): 65-87 % of its wall time, and all of
  // a driver's time here, where the chain steps are on the GPU.  A stream in its default state (precision 6, no
  // floatfield / showpoint / showpos / uppercase, no pending width, "C" locale) prints a float as printf("%g"): those
  // characters come from fmtg6 (exact, tests/cpp/fmt_check.cc), rows cut into pieces for the host's threads.
  if (stream_prints_like_printf(os)) {
    const size_t w = static_cast<size_t>(width_), nrows = count / w;
    const size_t piece_rows = std::max<size_t>(1, (size_t(1) << 21) / w);  // ~2 M numbers (36 MB of text at most) per piece
    const unsigned nthreads = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    std::vector<std::vector<char> > text(nthreads);
    std::vector<size_t> used(nthreads, 0);
    for (size_t r0 = 0; r0 < nrows; r0 += piece_rows * nthreads) {
      const size_t batch = std::min(nrows - r0, piece_rows * nthreads);
      const unsigned nt = format_batch(all, r0, batch, w, nthreads, text, used);
      for (unsigned t = 0; t < nt; ++t) os.write(text[t].data(), static_cast<std::streamsize>(used[t]));
    }
    for (size_t i = nrows * w; i < count; ++i) os << all[i] << "  ";  // (a ragged tail: never, rows are whole)
    delete[] owned;
    return;
  }
  for (size_t i = 0; i < count; ++i) {
    os << all[i] << "  ";
    if ((i + 1) % width_ == 0) os << "\n";
  }
  delete[] owned;
}

// rows added since the last flush, all ranks', rank-major; buffer exists on rank 0 only
float *MCout::collect(size_t *ntot)
{
  const size_t fresh = fill_ > flushed_ ? fill_ - flushed_ : 0;
  if (fresh == 0) {
    *ntot = 0;
    return 0;
  }
  // (like the reference, ranks other than 0 leave *ntot alone when they had rows to send: src/mcout.cc:80-92;
  // pinned by tests/golden/mcout_reference.json)
  float *gathered = 0;
  if (rank_ == 0) {
    *ntot = fresh * static_cast<size_t>(nranks_);
    gathered = new float[*ntot];
  }
#ifdef MCX_WITH_MPI
  if (nranks_ > 1) {
    const int st = MPI_Gather((void *)&rows_[flushed_], (int)fresh, MPI_FLOAT, (void *)gathered, (int)fresh,
                              MPI_FLOAT, 0, comm_);
    if (st != MPI_SUCCESS) {
      std::cerr << "Unable to gather output data.  Aborting.\n";
      MPI_Abort(MPI_COMM_WORLD, st);
    }
    flushed_ = fill_;
    return gathered;
  }
#endif
  if (gathered) std::memcpy(gathered, &rows_[flushed_], fresh * sizeof(float));
  flushed_ = fill_;
  return gathered;
}

const std::vector<float> &MCout::maxlike(float *lmax)
{
#ifdef MCX_WITH_MPI
  if (nranks_ > 1) {
    struct { float v; int r; } mine = {best_l_, rank_}, top;
    if (MPI_Allreduce(&mine, &top, 1, MPI_FLOAT_INT, MPI_MAXLOC, comm_) != MPI_SUCCESS ||
        MPI_Bcast(&best_p_[0], nparam_, MPI_FLOAT, top.r, comm_) != MPI_SUCCESS) {
      std::cerr << "rank " << rank_ << ":  MPI failure in MCout::maxlike().\n";
      MPI_Abort(MPI_COMM_WORLD, 1);
    }
    best_l_ = top.v;
  }
#endif
  *lmax = best_l_;
  return best_p_;
}

// ---------------------------------------------------------------------------------------------
// MCPar (src/mcpar.cc)
// ---------------------------------------------------------------------------------------------
// The duplicated communicator of src/mcpar.cc:228 and the exchange that replaces its MPI_Allgather.
struct MCPar::Comm {
#ifdef MCX_WITH_MPI
  MPI_Comm comm;
  std::vector<float> host;  // staging of the MPI fallback
#endif
  bool rccl;
  const char *backend;
};

#ifdef MCX_WITH_MPI
namespace {

// One MPI rank drives one GPU: ranks of a node take its devices round-robin by node-local rank
// (MCPAR_DEVICE=<index> overrides).  Must run before mcx_create, which binds the engine to the current device.
void bind_rank_to_device(MPI_Comm comm, int rank)
{
  int ndev = 0;
  if (mcx_device_count(&ndev) != MCX_OK || ndev < 1) return;  // mcx_create reports the missing device
  int dev = -1;
  if (const char *ov = getenv("MCPAR_DEVICE")) dev = atoi(ov);
  if (dev < 0) {
    MPI_Comm node;
    int local = 0;
    if (MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node) == MPI_SUCCESS) {
      MPI_Comm_rank(node, &local);
      MPI_Comm_free(&node);
    }
    dev = local % ndev;
  }
  if (mcx_set_device(dev % ndev) != MCX_OK) die("MCPar::MCPar (device binding)");
}

// true when no two ranks of the communicator sit on the same GPU (RCCL refuses such a communicator)
bool one_gpu_per_rank(MPI_Comm comm, int size)
{
  char mine[96];
  std::memset(mine, 0, sizeof mine);
  if (gethostname(mine, 48) != 0) mine[0] = '?';
  mine[47] = 0;
  const size_t hl = std::strlen(mine);
  mine[hl] = '/';
  if (mcx_device_pci_bus_id(mine + hl + 1, sizeof mine - hl - 1) != MCX_OK) return false;
  std::vector<char> all((size_t)size * sizeof mine);
  if (MPI_Allgather(mine, (int)sizeof mine, MPI_CHAR, all.data(), (int)sizeof mine, MPI_CHAR, comm) != MPI_SUCCESS) return false;
  for (int a = 0; a < size; ++a)
    for (int b = a + 1; b < size; ++b)
      if (std::memcmp(&all[(size_t)a * sizeof mine], &all[(size_t)b * sizeof mine], sizeof mine) == 0) return false;
  return true;
}

}  // namespace
#endif

MCPar::MCPar(int np, int nc, int mpisiz, int mpirank, float pl, float armin, float armax, float dfac,
             float ifac, int sync)
    : TGT_ARATE_MIN(armin), TGT_ARATE_MAX(armax), SCALE_DEC(dfac), SCALE_INC(ifac), PLOCAL(pl),
      SYNCSTEP(sync), logging(false), logstep(1000), nparam(np), nchain(nc), rank(mpirank),
      size(mpisiz), rng_t(0), eng(0), comm(0)
{
  ntot = np * nc;
  ncov = np * np;
  tchains = mpisiz * nchain;
  mpi = mpisiz > 1;
  std::memset(&counters, 0, sizeof counters);
#ifdef MCX_WITH_MPI
  if (mpi) {
    comm = new Comm();
    comm->rccl = false;
    comm->backend = "mpi-staged";
    if (MPI_Comm_dup(MPI_COMM_WORLD, &comm->comm) != MPI_SUCCESS) {
      std::cerr << "rank = " << rank << ":  Unable to create mcpar communicator (fatal)\n";
      MPI_Abort(MPI_COMM_WORLD, 3);
    }
    bind_rank_to_device(comm->comm, rank);
  }
#else
  if (mpi) throw("MCPar built without MPI: mpisiz must be 1 (rebuild with -DMCX_WITH_MPI)");
#endif
  create(8675309u);  // the reference's seed literal (src/mcpar.cc:271)
}

const char *MCPar::exchange_backend() const { return comm ? comm->backend : "none"; }

void MCPar::create(uint32_t seed)
{
  if (eng) mcx_destroy(eng);
  eng = 0;
  const int st = mcx_create(&eng, nparam, nchain, size, rank, PLOCAL, TGT_ARATE_MIN, TGT_ARATE_MAX,
                            SCALE_DEC, SCALE_INC, SYNCSTEP, seed);
  if (st == MCX_ERR_INVALID || st == MCX_ERR_UNSUPPORTED) throw("Invalid MCPar configuration");
  if (st != MCX_OK) die("MCPar::MCPar");
#ifdef MCX_WITH_MPI
  // Exchange of the (mu, sig^2) slots (src/mcpar.cc:127-140).  Preferred: the library's in-place
  // ncclAllGather on device memory (RCCL over xGMI) -- MPI only ships the communicator id, once, here
  // (collective, like the MPI_Comm_dup above).  Fallback, decided collectively: MPI_Allgather through host
  // memory, when two ranks share a GPU, RCCL is not loadable, its initialisation fails on any rank, or the
  // user asks for it (MCPAR_EXCHANGE=mpi).
  if (comm) {
    comm->rccl = false;
    comm->backend = "mpi-staged";
    const char *want = getenv("MCPAR_EXCHANGE");
    int ok = !(want && std::strcmp(want, "mpi") == 0) && mcx_rccl_available();
    int all_ok = 0;
    MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm->comm);
    if (all_ok && one_gpu_per_rank(comm->comm, size)) {
      unsigned char id[MCX_RCCL_ID_BYTES];
      std::memset(id, 0, sizeof id);
      ok = rank != 0 || mcx_rccl_unique_id(id) == MCX_OK;
      MPI_Bcast(id, (int)sizeof id, MPI_BYTE, 0, comm->comm);
      MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm->comm);
      if (all_ok) {
        // one real gather with known contents before the communicator is trusted: every rank fills its slot with
        // rank + 1, the gather runs, slot r must then be full of r + 1 on every rank
        ok = mcx_exchange_rccl_init(eng, id) == MCX_OK && mcx_debug_fill_slot(eng, (float)(rank + 1)) == MCX_OK &&
             mcx_debug_exchange(eng) == MCX_OK;
        if (!ok) std::cerr << "rank = " << rank << ":  RCCL exchange unavailable (" << mcx_last_error() << "), using MPI\n";
        if (ok) {
          std::vector<float> all((size_t)2 * size * nchain * nparam);
          const size_t slot = (size_t)2 * nchain * nparam;
          ok = mcx_get_musigall(eng, all.data()) == MCX_OK;
          for (int r = 0; ok && r < size; ++r)
            for (size_t i = 0; i < slot; ++i)
              if (all[(size_t)r * slot + i] != (float)(r + 1)) { ok = 0; break; }
          if (!ok) std::cerr << "rank = " << rank << ":  RCCL all-gather self-check failed (a slot did not arrive as sent), using MPI\n";
          (void)mcx_debug_fill_slot(eng, 0.0f);  // leave the slots as a fresh engine has them (collective, like the check)
          (void)mcx_debug_exchange(eng);
        }
        MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm->comm);
        if (all_ok) {
          comm->rccl = true;
          comm->backend = "rccl";
        } else {
          mcx_exchange_rccl_destroy(eng);
        }
      }
    }
  }
#endif
}

void MCPar::set_seed_and_recreate(uint32_t seed) { create(seed); }

MCPar::~MCPar()
{
  if (eng) mcx_destroy(eng);
#ifdef MCX_WITH_MPI
  if (comm) MPI_Comm_free(&comm->comm);
#endif
  delete comm;
}

void MCPar::covar_setup(const float *incov, float *restrict cov)
{
  if (mcx_covar_setup(eng, incov, cov) != MCX_OK) die("MCPar::covar_setup");
}

int MCPar::genLocal(const float pvals[], float *restrict ptrial, float *restrict cfac)
{
  if (mcx_gen_local(eng, rng_t++, pvals, ptrial, cfac) != MCX_OK) die("MCPar::genLocal");
  return 0;
}

int MCPar::genRemote(const float pvals[], float *restrict musigall, float *restrict ptrial,
                     float *restrict cfac)
{
  int npass = 0;
  if (mcx_gen_remote(eng, rng_t++, pvals, musigall, ptrial, cfac, 0, 0, &npass) != MCX_OK)
    die("MCPar::genRemote");
  return 0;
}

namespace {

struct HostL {
  VLFunc *L;
};
int host_tramp(void *ctx, int npset, const float *x, float *y) { return (*static_cast<HostL *>(ctx)->L)(npset, x, y); }

struct RunCtx {
  mcx_engine *eng;
  MCout *out;
  std::ofstream *log;
  int nchain, ncol, copied_steps, nsamp;
  std::vector<float> stage;
  // optional per-step diagnostics of the reference (src/mcpar.cc:121-126, 129-139)
  bool logging, mpi;
  int logstep, syncstep, logged_upto, size0;
  bool gpu_text;  // every block's text comes with its rows (MCX_OPT_SINK_TEXT): printed from it, the rows only stored
};

// The reference writes its `logging` diagnostics inside the step loop.  The steps run on the GPU in
// long launches here, so the same lines are written -- in the reference's order -- when the host
// regains control: everything of iterations [logged_upto, upto) that precedes iteration upto's dump.
void write_step_diagnostics(RunCtx *c, int upto)
{
  if (!c->logging || c->logstep < 1) {
    c->logged_upto = upto;
    return;
  }
  std::ofstream &lf = *c->log;
  for (int isamp = c->logged_upto; isamp < upto; ++isamp) {
    if (isamp % c->logstep == 0)
      lf << "sample step " << isamp << ":\toutsamples size= " << (c->size0 + isamp * c->nchain)
         << "  maxsize = " << c->out->maxsize() << "  ncol= " << c->out->ncol() << "\n\tvsize = " << c->out->vsize()
         << "  offset = " << (isamp * c->ncol) << std::endl;
    if (c->mpi && isamp % c->syncstep == 0) {
      lf << "\tisamp = " << isamp << "  entering allgather " << std::endl;
      lf << "\tisamp = " << isamp << "  exiting allgather " << std::endl;
    }
  }
  c->logged_upto = upto;
}

// The engine streams the samples out in blocks of `outstep` main-loop steps -- the interval at which the
// reference dumps output (src/mcpar.cc:110-119) -- through a small ring in HBM and pinned staging memory, while
// the chains keep stepping (mcx_set_sink).  Each block goes into MCout exactly as the reference's per-step
// MCout::add calls would have filled it (src/mcpar.cc:176-182), and MCout prints it where the reference does.
int sample_sink(void *vctx, int first_step, int nsteps, const float *rows)
{
  RunCtx *c = static_cast<RunCtx *>(vctx);
  const int steps_done = first_step + nsteps;
  c->out->add_rows(rows, (size_t)nsteps * c->nchain, false);  // (the running maximum comes from the engine: MCPar::run)
  c->copied_steps = steps_done;
  write_step_diagnostics(c, steps_done);  // iterations before this dump point
  const char *text = 0;
  size_t nbytes = 0;
  // (collective: a rank whose text could not be made -- an allocation failure inside the engine's sink -- must not go
  // down the row path, MPI_Gather, while the others send text: one "no" sends every rank of this block to the rows)
  const bool have_text = c->gpu_text && c->out->all_ranks_agree(mcx_sink_text(c->eng, &text, &nbytes) == MCX_OK);
  if (steps_done < c->nsamp) (*c->log) << "Beginning output at step " << steps_done << std::endl;
  if (have_text) {  // the characters output() would produce for these rows, made on the GPU (the last block's too:
    c->out->write_text(text, nbytes);  // its text exists only now; the run's final output() then finds nothing new)
    c->out->mark_flushed();
  } else if (steps_done < c->nsamp) {
    c->out->output();
  }
  if (steps_done < c->nsamp) (*c->log) << "Output finished\n" << std::endl;
  return 0;
}

// MCout::text_only: the same blocks, as text from the device, straight to the stream (mcx_set_text_sink)
int text_sink(void *vctx, int first_step, int nsteps, const char *text, size_t nbytes)
{
  RunCtx *c = static_cast<RunCtx *>(vctx);
  const int steps_done = first_step + nsteps;
  c->copied_steps = steps_done;
  write_step_diagnostics(c, steps_done);
  if (steps_done < c->nsamp) (*c->log) << "Beginning output at step " << steps_done << std::endl;
  c->out->write_text(text, nbytes);
  if (steps_done < c->nsamp) (*c->log) << "Output finished\n" << std::endl;
  return 0;
}

#ifdef MCX_WITH_MPI
struct XchgCtx {
  MPI_Comm comm;
  std::vector<float> &host;
};
// MPI_Allgather of src/mcpar.cc:127-140, staged through host memory (the fallback: see MCPar::create)
int mpi_exchange(void *vctx, int phase, void *dev, size_t slot, int shard, int nshards, void *stream)
{
  if (phase != MCX_XCHG_BEGIN) return 0;
  XchgCtx *c = static_cast<XchgCtx *>(vctx);
  c->host.resize(slot * nshards);
  float *d = static_cast<float *>(dev);
  if (mcx_copy_to_host(&c->host[slot * shard], d + slot * shard, slot * sizeof(float), stream) != MCX_OK) return 1;
  if (MPI_Allgather(MPI_IN_PLACE, 0, MPI_DATATYPE_NULL, c->host.data(), (int)slot, MPI_FLOAT, c->comm) != MPI_SUCCESS) {
    std::cerr << "Error in MPI_Allgather.  Aborting.\n";
    MPI_Abort(c->comm, 1);
  }
  for (int r = 0; r < nshards; ++r)
    if (r != shard &&
        mcx_copy_to_device(d + slot * r, &c->host[slot * r], slot * sizeof(float), stream) != MCX_OK)
      return 1;
  return 0;
}
#endif

}  // namespace

int MCPar::run(int nsamp, int nburn, const float *pinit, VLFunc &L, MCout &outsamples, float *incov)
{
  // log file (src/mcpar.cc:22-28)
  std::stringstream logname;
  if (rank == 0) logname << "mcpar-log." << std::setfill('0') << std::setw(3) << rank << ".txt";
  else logname << "/dev/null";
  std::ofstream logfile(logname.str().c_str());

  const bool as_text = outsamples.text_only();  // (several ranks: MCout::write_text gathers the texts in rank order)
  try {
    if (!as_text) outsamples.newsamps(nsamp * nchain);  // src/mcpar.cc:31
  } catch (std::bad_alloc &) {
    logfile << "Unable to allocate space for output samples.  Exiting.\n";
    exit(2);
  }

  mcx_vlfunc f;
  HostL hl = {&L};
  if (!L.device_descriptor(nparam, &f)) f = mcx_vlfunc{MCX_VL_HOST, nparam, 0, 0, host_tramp, &hl};

  // the text of every block from the GPU next to its rows, when MCout::output would print plain printf("%g") text anyway
  // (rank 0 knows its stream; with several ranks every rank follows its answer)
  int gpu_text = (!as_text && outsamples.prints_plain_text()) ? 1 : 0;
#ifdef MCX_WITH_MPI
  if (mpi && comm) MPI_Bcast(&gpu_text, 1, MPI_INT, 0, comm->comm);
#endif
  mcx_set_option(eng, MCX_OPT_SINK_TEXT, gpu_text);
  // MCPAR_REFERENCE_CALLS=1: a host functor also gets the reference's discarded per-chain calls L(1, pvals_j, &y) after
  // every main-loop step (src/mcpar.cc:177-182) -- for functors that count or cache their calls; results do not change
  {
    const char *rc_env = getenv("MCPAR_REFERENCE_CALLS");
    mcx_set_option(eng, MCX_OPT_REFERENCE_CALLS, (rc_env && *rc_env && *rc_env != '0') ? 1 : 0);
  }
  RunCtx ctx = {eng,   &outsamples, &logfile, nchain,   nparam + 1, 0, nsamp, std::vector<float>(),
                logging, mpi,        logstep,  SYNCSTEP, 0,          outsamples.size(), gpu_text != 0};
  const int outstep = nsamp > 50 ? nsamp / 10 : 5;  // src/mcpar.cc:110
  if (as_text) mcx_set_text_sink(eng, text_sink, &ctx, outstep);
  else mcx_set_sink(eng, sample_sink, &ctx, outstep);
#ifdef MCX_WITH_MPI
  std::vector<float> nohost;
  XchgCtx xc = {comm ? comm->comm : MPI_COMM_WORLD, comm ? comm->host : nohost};
  const bool staged = mpi && comm && !comm->rccl;
  if (staged) mcx_set_exchange(eng, mpi_exchange, &xc);
#endif
  logfile << "Starting burn-in.  Samples = " << nburn << std::endl;  // src/mcpar.cc:56
  logfile << "Starting main sample loop:  nsamp = " << nsamp << std::endl;  // :111-112
  logfile << "Output after each " << outstep << " steps." << std::endl;

  const int st = mcx_run(eng, nsamp, nburn, pinit, &f, incov);
  mcx_set_sink(eng, 0, 0, 0);
  mcx_set_option(eng, MCX_OPT_SINK_TEXT, 0);
  if (st == MCX_OK) write_step_diagnostics(&ctx, nsamp);
#ifdef MCX_WITH_MPI
  if (staged) mcx_set_exchange(eng, 0, 0);  // its context lives on this stack frame
#endif
  if (st == MCX_ERR_ALLOC) {
    logfile << "Unable to allocate space for output samples.  Exiting.\n";
    exit(2);
  }
  if (st != MCX_OK) die("MCPar::run");
  mcx_get_counters(eng, &counters);
  if (nsamp > 0) {  // MCout's maxlike() answers from the engine's running maximum (first strict maximum, like MCout::add)
    float lbest = 0.0f;
    std::vector<float> pbest(static_cast<size_t>(nparam));
    if (mcx_samples_maxlike(eng, &lbest, pbest.data()) == MCX_OK) outsamples.note_best(lbest, pbest.data());
  }
  outsamples.output();  // output remaining samples (src/mcpar.cc:212)
  return 0;
}

// ---------------------------------------------------------------------------------------------
// mcutil::qriguess (role of src/mcutil.cc:3-34): Sobol points scaled into [plo, phi]
// ---------------------------------------------------------------------------------------------
namespace {

// Joe & Kuo (2008) "new-joe-kuo-6" primitive polynomials and initial direction numbers, dimensions
// 2..21: {degree s, coefficients a, m_1..m_s}.  Dimension 1 is the van der Corput sequence.
const unsigned kSobolInit[20][9] = {
    {1, 0, 1},          {2, 1, 1, 3},          {3, 1, 1, 3, 1},        {3, 2, 1, 1, 1},
    {4, 1, 1, 1, 3, 3}, {4, 4, 1, 3, 5, 13},   {5, 2, 1, 1, 5, 5, 17}, {5, 4, 1, 1, 5, 5, 5},
    {5, 7, 1, 1, 7, 11, 19},  {5, 11, 1, 1, 5, 1, 1},   {5, 13, 1, 1, 1, 3, 11}, {5, 14, 1, 3, 5, 5, 31},
    {6, 1, 1, 3, 3, 9, 7, 49}, {6, 13, 1, 1, 1, 15, 21, 21}, {6, 16, 1, 3, 1, 13, 27, 49},
    {6, 19, 1, 1, 1, 15, 7, 5}, {6, 22, 1, 3, 1, 15, 13, 25}, {6, 25, 1, 1, 5, 5, 19, 61},
    {7, 1, 1, 3, 7, 11, 23, 15, 103}, {7, 4, 1, 3, 7, 13, 13, 15, 69}};

// direction numbers v[j] (32 bits, bit 31 = 1/2) of one dimension
void sobol_directions(int dim, unsigned v[32])
{
  if (dim == 0) {
    for (int j = 0; j < 32; ++j) v[j] = 1u << (31 - j);
    return;
  }
  const unsigned *row = kSobolInit[dim - 1];
  const int s = (int)row[0];
  const unsigned a = row[1];
  unsigned m[32];
  for (int j = 0; j < 32; ++j) {
    if (j < s) m[j] = row[2 + j];
    else {
      unsigned x = m[j - s] ^ (m[j - s] << s);
      for (int k = 1; k < s; ++k)
        if ((a >> (s - 1 - k)) & 1u) x ^= m[j - k] << k;
      m[j] = x;
    }
    v[j] = m[j] << (31 - j);
  }
}

}  // namespace

void mcutil::qriguess(int rank, int npset, int nparam, const float plo[], const float phi[], float *restrict pout)
{
  if (nparam < 1 || nparam > MAXDIM) throw("mcutil::qriguess supports 1 <= nparam <= 21");
  if (npset < 0 || rank < 0) throw("mcutil::qriguess: bad rank or npset");
  const unsigned long long first = (unsigned long long)rank * (unsigned long long)npset;
  for (int i = 0; i < nparam; ++i) {
    unsigned v[32];
    sobol_directions(i, v);
    for (int j = 0; j < npset; ++j) {
      // point number n (counting from 1, so that the all-zero point is skipped) in Gray-code order
      const unsigned long long n = first + (unsigned long long)j + 1ull;
      const unsigned long long g = n ^ (n >> 1);
      unsigned x = 0;
      for (int b = 0; b < 32; ++b)
        if ((g >> b) & 1ull) x ^= v[b];
      const float u = (float)(x >> 8) * (1.0f / 16777216.0f);  // [0, 1) on a 2^-24 grid
      pout[(size_t)j * nparam + i] = plo[i] + u * (phi[i] - plo[i]);
    }
  }
}
