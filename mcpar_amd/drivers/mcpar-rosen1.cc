// mcpar-rosen1 [nsamp] -- MI355X build of the reference's Rosenbrock demo: 2-D Rosenbrock1,
// 4 chains per rank, 500 burn-in steps, nsamp kept steps (default 100000).  stdout: "nsamp = N",
// then the sample rows.
#include <cstdlib>

#include "driver_util.hh"
#include "mcpar/mcpar.hh"
#include "mcpar/rosenbrock.hh"

int main(int argc, char *argv[])
{
  drv::Session mpi(argc, argv);
  if (!mpi.ok) return 1;

  const int nkeep = argc > 1 ? std::atoi(argv[1]) : 100000;
  Rosenbrock1 target(2);
  MCout store(2, &std::cout, MPI_COMM_WORLD);
  if (mpi.rank == 0) std::cout << "nsamp = " << nkeep << "\n";

  MCPar sampler(2, 4, mpi.nranks, mpi.rank);
  sampler.run(nkeep, 500, drv::demo_start(), target, store);
  store.output();  // nothing left after run(); kept for parity with the reference demo
  return 0;
}
