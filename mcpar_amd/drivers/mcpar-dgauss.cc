// mcpar-dgauss -- same command line (none) and output surface as the reference demo
// (src/mcpar-dgauss.cc): 2-D DualGaussian(5), 4 chains per rank, 500 burn-in + 8 samples;
// stdout = sample rows then "max likelihood value: X" and the parameters; per-rank file
// mcpar-dgauss.RRR.txt with tab-separated parameters; log in mcpar-log.000.txt.
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

#include "mcpar/mcout.hh"
#include "mcpar/mcpar.hh"
#include "mcpar/rosenbrock.hh"

int main(int argc, char *argv[])
{
  const int nparam = 2;
  DualGaussian L(5.0f);

  if (MPI_Init(&argc, &argv) != MPI_SUCCESS) {
    std::cerr << "Error on MPI_Init.  Exiting.\n";
    return 1;
  }
  int size, rank;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);

  MCout rslts(nparam, &std::cout, MPI_COMM_WORLD);
  MCPar mcpar(nparam, 4, size, rank);  // 2 parameters, 4 chains per process
  float pinit[8] = {0.0f, 0.0f, 2.0f, 2.0f, 0.0f, 1.5f, 0.0f, -2.0f};
  mcpar.run(8, 500, pinit, L, rslts);

  std::stringstream ofname;
  ofname << "mcpar-dgauss." << std::setfill('0') << std::setw(3) << rank << ".txt";
  std::ofstream outfile(ofname.str().c_str());
  for (int i = 0; i < rslts.size(); ++i) {
    const float *pset = rslts.getpset(i);
    for (int j = 0; j < rslts.ncol() - 1; ++j) outfile << pset[j] << "\t";
    outfile << "\n";
  }

  float lmax;
  const std::vector<float> &pmax = rslts.maxlike(&lmax);
  std::cout << "max likelihood value: " << lmax << "\n";
  for (size_t i = 0; i < pmax.size(); ++i) std::cout << pmax[i] << "  ";
  std::cout << "\n";

  MPI_Finalize();
  return 0;
}
