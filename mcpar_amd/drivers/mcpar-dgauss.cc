// mcpar-dgauss -- MI355X build of the reference's two-Gaussian demo: no arguments; 2-D
// DualGaussian(5), 4 chains per rank, 500 burn-in steps + 8 kept steps.  Output surface as the
// reference's: sample rows on stdout, then the maximum-likelihood sample; mcpar-dgauss.RRR.txt with
// the parameters of this rank's samples; mcpar-log.000.txt.
#include "driver_util.hh"
#include "mcpar/mcpar.hh"
#include "mcpar/rosenbrock.hh"

int main(int argc, char *argv[])
{
  drv::Session mpi(argc, argv);
  if (!mpi.ok) return 1;

  enum { NPARAM = 2, NCHAIN = 4, NBURN = 500, NKEEP = 8 };
  DualGaussian target(5.0f);
  MCout store(NPARAM, &std::cout, MPI_COMM_WORLD);
  MCPar sampler(NPARAM, NCHAIN, mpi.nranks, mpi.rank);
  sampler.run(NKEEP, NBURN, drv::demo_start(), target, store);

  drv::dump_params(store, "mcpar-dgauss", mpi.rank);
  drv::report_maxlike(store, std::cout);
  return 0;
}
