// mcutil_check.cc -- prints Sobol initial guesses for inspection by tests/test_mcutil_cpu.py
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mcpar/mcutil.hh"

int main(int argc, char **argv)
{
  const int rank = atoi(argv[1]), npset = atoi(argv[2]), nparam = atoi(argv[3]);
  std::vector<float> lo(nparam), hi(nparam), out((size_t)npset * nparam);
  for (int i = 0; i < nparam; ++i) { lo[i] = -1.0f - i; hi[i] = 2.0f + 0.5f * i; }
  mcutil u;
  try {
    u.qriguess(rank, npset, nparam, lo.data(), hi.data(), out.data());
  } catch (const char *msg) {
    std::printf("throw: %s\n", msg);
    return 3;
  }
  for (int j = 0; j < npset; ++j) {
    for (int i = 0; i < nparam; ++i) std::printf("%.9g ", out[(size_t)j * nparam + i]);
    std::printf("\n");
  }
  return 0;
}
