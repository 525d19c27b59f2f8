// mcpar.hh -- the reference's driver class (src/mcpar.hh:1-95), same public interface, running on
// an MI355X through libmcx (include/mcx.h).  Code written against the reference:
//
//     MCout rslts(np, &std::cout, MPI_COMM_WORLD);
//     MCPar mcpar(np, nchain, size, rank);
//     mcpar.run(nsamp, nburn, pinit, L, rslts);
//
// compiles and runs unchanged; one MPI rank drives one GPU (shard).
#ifndef MCPAR_AMD_MCPAR_HH_
#define MCPAR_AMD_MCPAR_HH_

#include <stdint.h>
#include <stdlib.h>

#include "../mcx.h"
#include "mcout.hh"
#include "mpi_compat.hh"
#include "vlfunc.hh"

MCPAR_ABI_NAMESPACE_BEGIN

class MCPar {
public:
  enum { OK, INVALID, ERROR };

  const float TGT_ARATE_MIN;  // target acceptance rate minimum
  const float TGT_ARATE_MAX;  // target acceptance rate maximum
  const float SCALE_DEC;      // decrement factor when acceptance rate is too low
  const float SCALE_INC;      // increment factor when acceptance rate is too high
  const float PLOCAL;         // probability of taking a local (instead of remote) proposal
  const int SYNCSTEP;         // steps between synchronisations of the Gaussian posterior estimates
  static const float FPEPS;

  // logging switches.  Users may set and reset these as desired
  bool logging;
  int logstep;

  MCPar(int np, int nc = 1, int mpisiz = 1, int mpirank = 0, float pl = 0.9, float armin = 0.2,
        float armax = 0.5, float dfac = 0.2, float ifac = 1.5, int sync = 10);
  ~MCPar();

  int run(int nsamp, int nburn, const float *pinit, VLFunc &L, MCout &outsamples, float *incov = 0);
  void covar_setup(const float incov[], float *restrict cov);
  int genLocal(const float pvals[], float *restrict ptrial, float *restrict cfac);
  int genRemote(const float pvals[], float *restrict musigall, float *restrict ptrial,
                float *restrict cfac);

  // ---- additions (the reference never reports these: SURVEY fact 7) ----
  mcx_engine *engine() { return eng; }
  uint64_t naccept_burn() const { return counters.naccept_burn; }
  uint64_t naccept_main() const { return counters.naccept_main; }
  uint64_t remote_passes() const { return counters.remote_passes; }
  void set_seed_and_recreate(uint32_t seed);
  // how the shards exchange their (mu, sig^2) slots (src/mcpar.cc:127-140): "none" (one shard), "rccl"
  // (in-place ncclAllGather on device memory over xGMI; MPI only ships the communicator id) or "mpi-staged"
  // (MPI_Allgather through host memory: ranks sharing a GPU, no RCCL, or MCPAR_EXCHANGE=mpi)
  const char *exchange_backend() const;

private:
  int nparam, nchain, ntot, ncov;
  bool mpi;
  int rank, size, tchains;
  uint32_t rng_t;  // RNG step index used by the public genLocal/genRemote
  mcx_engine *eng;
  mcx_counters counters;
  struct Comm;  // the duplicated communicator (src/mcpar.cc:228) and the exchange it drives; null without MPI
  Comm *comm;
  void create(uint32_t seed);
  MCPar(const MCPar &);
  MCPar &operator=(const MCPar &);
};

MCPAR_ABI_NAMESPACE_END

#endif
