// facade.cc -- MCPar / MCout with the reference's public interface (src/mcpar.hh, src/mcout.hh),
// implemented over the C ABI of libmcx.  Host-only C++: no HIP types cross this file.
//
// Error behaviour follows the reference: constructor argument errors throw a string literal
// (src/mcpar.cc:268), device/runtime failures print and abort() (VSL_CALL_CHK, src/mcpar.hh:93),
// a sample store that does not fit prints "Unable to allocate space for output samples." and exits
// with status 2 (src/mcpar.cc:34-40), run() returns 0 (src/mcpar.cc:213).
#include "../../include/mcpar/mcpar.hh"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <new>
#include <sstream>

const float MCPar::FPEPS = 1.0e-14f;

static void die(const char *where)
{
  std::cerr << "mcpar: " << where << ": " << mcx_last_error() << std::endl;
  abort();
}

// ---------------------------------------------------------------------------------------------
// MCout (src/mcout.cc)
// ---------------------------------------------------------------------------------------------
MCout::MCout(int anparam, std::ostream *aoutstream, MPI_Comm acomm)
    : mnparam(anparam), mncol(anparam + 1), next(0), npset(0), maxsamps(0), nextout(0)
{
#ifdef MCX_WITH_MPI
  if (MPI_Comm_dup(acomm, &mComm) != MPI_SUCCESS) {
    std::cerr << "MCout: unable to duplicate input communicator in constructor." << std::endl;
    MPI_Abort(acomm, 1);
  }
#else
  mComm = acomm;
#endif
  MPI_Comm_rank(mComm, &mrank);
  MPI_Comm_size(mComm, &msize);
  maxlparams.resize(mnparam);
  maxlval = -(std::numeric_limits<float>::infinity());
  outstream = mrank == 0 ? aoutstream : 0;  // all output through rank 0, like the reference
}

void MCout::add(const float *pv, float lval)
{
  assert(next + mncol <= pvals.size());
  float *strt = &pvals[next];
  for (int i = 0; i < mnparam; ++i) strt[i] = pv[i];
  strt[mnparam] = lval;
  next += mncol;
  npset++;
  if (lval > maxlval) {
    maxlval = lval;
    for (int i = 0; i < mnparam; ++i) maxlparams[i] = pv[i];
  }
}

void MCout::add_rows(const float *rows, size_t nrows)
{
  assert(next + nrows * mncol <= pvals.size());
  std::memcpy(&pvals[next], rows, nrows * mncol * sizeof(float));
  for (size_t r = 0; r < nrows; ++r) {
    const float *row = rows + r * mncol;
    if (row[mnparam] > maxlval) {
      maxlval = row[mnparam];
      for (int i = 0; i < mnparam; ++i) maxlparams[i] = row[i];
    }
  }
  next += nrows * mncol;
  npset += (int)nrows;
}

// rows as "v0  v1  ...  LL  \n": two spaces after every field, stream default precision
void MCout::output()
{
  size_t ntot = 0;
  float *buf = collect(&ntot);
  if (mrank == 0 && ntot > 0) {
    const size_t nrow = ntot / mncol;
    size_t indx = 0;
    for (size_t i = 0; i < nrow; ++i) {
      for (int j = 0; j < mncol; ++j) (*outstream) << buf[indx++] << "  ";
      (*outstream) << "\n";
    }
    delete[] buf;
  }
}

// newly allocated buffer on rank 0 (caller deletes), NULL elsewhere; rank-major order
float *MCout::collect(size_t *ntot)
{
  float *buf = 0;
  const size_t nout = next > nextout ? next - nextout : 0;
  if (nout == 0) {
    *ntot = 0;
    return buf;
  }
  if (mrank == 0) {
    *ntot = (size_t)msize * nout;
    buf = new float[*ntot];
  } else {
    *ntot = 0;
  }
#ifdef MCX_WITH_MPI
  if (msize > 1) {
    int st = MPI_Gather((void *)&pvals[nextout], (int)nout, MPI_FLOAT, (void *)buf, (int)nout, MPI_FLOAT, 0, mComm);
    if (st != MPI_SUCCESS) {
      std::cerr << "Unable to gather output data.  Aborting.\n";
      MPI_Abort(MPI_COMM_WORLD, st);
    }
    nextout = next;
    return buf;
  }
#endif
  if (mrank == 0) std::memcpy(buf, &pvals[nextout], nout * sizeof(float));
  nextout = next;
  return buf;
}

const std::vector<float> &MCout::maxlike(float *lmax)
{
#ifdef MCX_WITH_MPI
  if (msize > 1) {
    struct { float val; int rank; } snd, rcv;
    snd.val = maxlval;
    snd.rank = mrank;
    if (MPI_Allreduce(&snd, &rcv, 1, MPI_FLOAT_INT, MPI_MAXLOC, mComm) != MPI_SUCCESS ||
        MPI_Bcast(&maxlparams[0], mnparam, MPI_FLOAT, rcv.rank, mComm) != MPI_SUCCESS) {
      std::cerr << "rank " << mrank << ":  MPI failure in MCout::maxlike().\n";
      MPI_Abort(MPI_COMM_WORLD, 1);
    }
    maxlval = rcv.val;
  }
#endif
  *lmax = maxlval;
  return maxlparams;
}

// ---------------------------------------------------------------------------------------------
// MCPar (src/mcpar.cc)
// ---------------------------------------------------------------------------------------------
MCPar::MCPar(int np, int nc, int mpisiz, int mpirank, float pl, float armin, float armax, float dfac,
             float ifac, int sync)
    : TGT_ARATE_MIN(armin), TGT_ARATE_MAX(armax), SCALE_DEC(dfac), SCALE_INC(ifac), PLOCAL(pl),
      SYNCSTEP(sync), logging(false), logstep(1000), nparam(np), nchain(nc), rank(mpirank),
      size(mpisiz), rng_t(0), eng(0)
{
  ntot = np * nc;
  ncov = np * np;
  tchains = mpisiz * nchain;
  mpi = mpisiz > 1;
  std::memset(&counters, 0, sizeof counters);
#ifdef MCX_WITH_MPI
  if (mpi && MPI_Comm_dup(MPI_COMM_WORLD, &mcparComm) != MPI_SUCCESS) {
    std::cerr << "rank = " << rank << ":  Unable to create mcpar communicator (fatal)\n";
    MPI_Abort(MPI_COMM_WORLD, 3);
  }
#else
  if (mpi) throw("MCPar built without MPI: mpisiz must be 1 (rebuild with -DMCX_WITH_MPI)");
#endif
  create(8675309u);  // the reference's seed literal (src/mcpar.cc:271)
}

void MCPar::create(uint32_t seed)
{
  if (eng) mcx_destroy(eng);
  eng = 0;
  const int st = mcx_create(&eng, nparam, nchain, size, rank, PLOCAL, TGT_ARATE_MIN, TGT_ARATE_MAX,
                            SCALE_DEC, SCALE_INC, SYNCSTEP, seed);
  if (st == MCX_ERR_INVALID || st == MCX_ERR_UNSUPPORTED) throw("Invalid MCPar configuration");
  if (st != MCX_OK) die("MCPar::MCPar");
}

void MCPar::set_seed_and_recreate(uint32_t seed) { create(seed); }

MCPar::~MCPar()
{
  if (eng) mcx_destroy(eng);
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
};

// where the reference dumps output (src/mcpar.cc:115-119): move the new rows from the HBM sample
// store into MCout, then let MCout print them
int output_hook(void *vctx, int steps_done)
{
  RunCtx *c = static_cast<RunCtx *>(vctx);
  const int ns = steps_done - c->copied_steps;
  if (ns > 0) {
    c->stage.resize((size_t)ns * c->nchain * c->ncol);
    if (mcx_samples_copy(c->eng, c->copied_steps, ns, c->stage.data()) != MCX_OK) return 1;
    c->out->add_rows(c->stage.data(), (size_t)ns * c->nchain);
    c->copied_steps = steps_done;
  }
  if (steps_done < c->nsamp) {
    (*c->log) << "Beginning output at step " << steps_done << std::endl;
    c->out->output();
    (*c->log) << "Output finished\n" << std::endl;
  }
  return 0;
}

#ifdef MCX_WITH_MPI
struct XchgCtx {
  MPI_Comm comm;
  std::vector<float> host;
};
// MPI_Allgather of src/mcpar.cc:127-140, staged through host memory
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

  try {
    outsamples.newsamps(nsamp * nchain);  // src/mcpar.cc:31
  } catch (std::bad_alloc &) {
    logfile << "Unable to allocate space for output samples.  Exiting.\n";
    exit(2);
  }

  mcx_vlfunc f;
  HostL hl = {&L};
  if (!L.device_descriptor(nparam, &f)) f = mcx_vlfunc{MCX_VL_HOST, nparam, 0, 0, host_tramp, &hl};

  RunCtx ctx = {eng, &outsamples, &logfile, nchain, nparam + 1, 0, nsamp, std::vector<float>()};
  mcx_set_output_hook(eng, output_hook, &ctx);
#ifdef MCX_WITH_MPI
  XchgCtx xc;
  if (mpi) {
    xc.comm = mcparComm;
    mcx_set_exchange(eng, mpi_exchange, &xc);
  }
#endif
  logfile << "Starting burn-in.  Samples = " << nburn << std::endl;  // src/mcpar.cc:56
  const int outstep = nsamp > 50 ? nsamp / 10 : 5;
  logfile << "Starting main sample loop:  nsamp = " << nsamp << std::endl;  // :111-112
  logfile << "Output after each " << outstep << " steps." << std::endl;

  const int st = mcx_run(eng, nsamp, nburn, pinit, &f, incov);
  mcx_set_output_hook(eng, 0, 0);
  if (st == MCX_ERR_ALLOC) {
    logfile << "Unable to allocate space for output samples.  Exiting.\n";
    exit(2);
  }
  if (st != MCX_OK) die("MCPar::run");
  mcx_get_counters(eng, &counters);
  outsamples.output();  // output remaining samples (src/mcpar.cc:212)
  return 0;
}
