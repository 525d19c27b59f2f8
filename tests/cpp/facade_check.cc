// facade_check.cc -- drives the reference-shaped C++ API (MCPar / MCout / VLFunc) the way the
// reference's own drivers do, once with the device-backed Rosenbrock1 and once with a user-written
// VLFunc subclass (host callback path) that evaluates the same function with the same operation
// order.  Prints both runs' samples; tests/test_gpu_facade.py compares them with each other and
// with the oracle.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <vector>

#include "mcpar/mcout.hh"
#include "mcpar/mcpar.hh"
#include "mcpar/rosenbrock.hh"

// a user likelihood exactly as a reference user would write one (src/vlfunc.hh:9-12)
class UserRosen : public VLFunc {
  const int n;
public:
  int ncalls;
  UserRosen(int nc) : n(nc), ncalls(0) {}
  int operator()(int npset, const float *x, float *restrict fx)
  {
    ++ncalls;
    for (int j = 0; j < npset; ++j) {
      float part[64];
      const int nb = (n + 3) / 4;
      for (int q = 0; q < nb; ++q) {
        float acc = 0.0f;
        for (int k = 4 * q; k + 1 < n && k < 4 * q + 4; k += 2) {
          const float *p = x + (size_t)j * n + k;
          float t1 = 1.0f - p[0];
          float t2 = std::fmaf(-p[0], p[0], p[1]);
          acc = acc + std::fmaf(100.0f * t2, t2, t1 * t1);
        }
        part[q] = acc;
      }
      int p2 = 1;
      while (p2 < nb) p2 <<= 1;
      for (int q = nb; q < p2; ++q) part[q] = 0.0f;
      for (int s = 1; s < p2; s <<= 1) {
        float nxt[64];
        for (int q = 0; q < p2; ++q) nxt[q] = part[q] + part[q ^ s];
        for (int q = 0; q < p2; ++q) part[q] = nxt[q];
      }
      fx[j] = 0.0f - part[0];
    }
    return 0;
  }
};

static void run(VLFunc &L, int np, int nc, int nsamp, int nburn, float pl, std::ostream &os)
{
  MCout rslts(np, &os, MPI_COMM_WORLD);
  MCPar mcpar(np, nc, 1, 0, pl);
  std::vector<float> pinit((size_t)nc * np);
  for (int j = 0; j < nc; ++j)
    for (int i = 0; i < np; ++i) pinit[(size_t)j * np + i] = (float)(0.5 * std::sin(0.37 * ((double)j * np + i)));
  mcpar.run(nsamp, nburn, pinit.data(), L, rslts);
  float lmax;
  const std::vector<float> &pm = rslts.maxlike(&lmax);
  os << "size " << rslts.size() << " maxsize " << rslts.maxsize() << " ncol " << rslts.ncol() << " maxlike "
     << lmax << " p0 " << pm[0] << " accepts " << mcpar.naccept_main() << "\n";
}

int main()
{
  const int np = 8, nc = 48, nsamp = 60, nburn = 120;
  std::ostringstream a, b;
  Rosenbrock1 builtin(np);
  UserRosen user(np);
  run(builtin, np, nc, nsamp, nburn, 0.8f, a);
  run(user, np, nc, nsamp, nburn, 0.8f, b);
  std::cout << a.str() << "=====\n" << b.str() << "=====\n" << "user calls " << user.ncalls << "\n";
  {  // the reference's optional step diagnostics (mcpar.logging / logstep)
    std::ostringstream c;
    MCout r2(np, &c, MPI_COMM_WORLD);
    MCPar m2(np, 4, 1, 0, 1.0f);
    m2.logging = true;
    m2.logstep = 7;
    std::vector<float> p0(4 * np, 0.25f);
    m2.run(20, 10, p0.data(), builtin, r2);
  }
  // constructor guard of the reference (src/rosenbrock.hh:13-16)
  try {
    Rosenbrock1 bad(3);
    std::cout << "guard missing\n";
  } catch (const char *msg) {
    std::cout << "guard: " << msg << "\n";
  }
  // the functor itself is a batched device call
  float x[4] = {1, 1, 0, 0}, y[2];
  Rosenbrock1 r2(2);
  r2(2, x, y);
  std::cout << "r2 " << y[0] << " " << y[1] << "\n";
  return 0;
}
