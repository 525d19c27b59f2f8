// driver_util.hh -- small helpers shared by the demo drivers of the MI355X engine.
#ifndef MCPAR_AMD_DRIVER_UTIL_HH_
#define MCPAR_AMD_DRIVER_UTIL_HH_

#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "mcpar/mcout.hh"

namespace drv {

// RAII around MPI_Init / MPI_Finalize (no-ops in a build without MPI): one rank drives one shard
struct Session {
  int nranks = 1, rank = 0;
  bool ok = false;
  Session(int &argc, char **&argv)
  {
    ok = MPI_Init(&argc, &argv) == MPI_SUCCESS;
    if (!ok) {
      std::cerr << "Error on MPI_Init.  Exiting.\n";
      return;
    }
    MPI_Comm_size(MPI_COMM_WORLD, &nranks);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  }
  ~Session()
  {
    if (ok) MPI_Finalize();
  }
};

// "<stem>.RRR.txt": one line per stored sample, parameters only, a tab after each value
inline void dump_params(const MCout &store, const char *stem, int rank)
{
  char name[256];
  std::snprintf(name, sizeof name, "%s.%03d.txt", stem, rank);
  std::ofstream f(name);
  const int np = const_cast<MCout &>(store).ncol() - 1;
  for (int r = 0; r < store.size(); ++r) {
    const float *row = store.getpset(r);
    for (int k = 0; k < np; ++k) f << row[k] << "\t";
    f << "\n";
  }
}

// collective: best sample over all ranks, printed by every rank like the reference demo
inline void report_maxlike(MCout &store, std::ostream &os)
{
  float best = 0.0f;
  const std::vector<float> &at = store.maxlike(&best);
  os << "max likelihood value: " << best << "\n";
  for (float v : at) os << v << "  ";
  os << "\n";
}

// the four 2-D starting points the reference demos hard-wire
inline const float *demo_start()
{
  static const float p[8] = {0.0f, 0.0f, 2.0f, 2.0f, 0.0f, 1.5f, 0.0f, -2.0f};
  return p;
}

}  // namespace drv

#endif
