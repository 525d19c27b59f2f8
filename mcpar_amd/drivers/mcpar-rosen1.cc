// mcpar-rosen1 [nsamp] -- same command line and output surface as the reference demo
// (src/mcpar-rosen1.cc): 2-D Rosenbrock1, 4 chains per rank, 500 burn-in, nsamp (default 100000)
// samples; stdout = "nsamp = N" then the sample rows.
#include <cstdlib>
#include <iostream>

#include "mcpar/mcout.hh"
#include "mcpar/mcpar.hh"
#include "mcpar/rosenbrock.hh"

int main(int argc, char *argv[])
{
  const int nparam = 2;
  Rosenbrock1 L(2);
  int nsamp = 100000;

  if (MPI_Init(&argc, &argv) != MPI_SUCCESS) {
    std::cerr << "Error on MPI_Init.  Exiting.\n";
    return 1;
  }
  MCout rslts(nparam, &std::cout, MPI_COMM_WORLD);
  int size, rank;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);

  if (argc > 1) nsamp = atoi(argv[1]);
  if (rank == 0) std::cout << "nsamp = " << nsamp << "\n";

  MCPar mcpar(nparam, 4, size, rank);
  float pinit[8] = {0.0f, 0.0f, 2.0f, 2.0f, 0.0f, 1.5f, 0.0f, -2.0f};
  mcpar.run(nsamp, 500, pinit, L, rslts);
  rslts.output();

  MPI_Finalize();
  return 0;
}
