// ref_shim.cc -- ORACLE-SIDE harness (test infrastructure, NOT product code).
//
// extern "C" entry points over the reference's own likelihood classes, compiled together with
// /root/reference/src/rosenbrock.cc where it lies (never copied).  Output: oracle/_ref/.
// Used only to pin oracle/mcx_oracle.c's likelihood restatements and to generate the vectors
// in tests/golden/vlfunc_reference.json (tools: oracle/gen_golden.py).
//
// mcpar.cc itself cannot be built here: it needs <mkl.h>/<mkl_vsl.h>, which this image lacks
// (only the MKL runtime .so files are present), and writing stand-in headers is not allowed.
#include "rosenbrock.hh"

extern "C" {

int ref_rosenbrock1(int n, int npset, const float *x, float *y)
{
  try { Rosenbrock1 f(n); return f(npset, x, y); } catch (const char *) { return -1; }
}
int ref_rosenbrock2(int n, int npset, const float *x, float *y)
{
  try { Rosenbrock2 f(n); return f(npset, x, y); } catch (const char *) { return -1; }
}
int ref_gaussian(int n, const float *mu, const float *sig2, int npset, const float *x, float *y)
{
  try { Gaussian f(n, mu, sig2); return f(npset, x, y); } catch (const char *) { return -1; }
}
int ref_dualgaussian(float w, int npset, const float *x, float *y)
{
  DualGaussian f(w);
  return f(npset, x, y);
}

}
