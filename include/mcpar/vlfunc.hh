// vlfunc.hh -- the reference's plug-in likelihood interface (src/vlfunc.hh:1-15), kept name for name
// so that user VLFunc subclasses compile unchanged against the MI355X engine.
#ifndef MCPAR_AMD_VLFUNC_HH_
#define MCPAR_AMD_VLFUNC_HH_

#include "../mcx.h"

#ifndef restrict
#define restrict __restrict__ /* the reference builds with -Drestrict=__restrict__ (src/Makefile:12) */
#endif

/* Vector likelihood function:
 * npset:   number of full sets of parameters
 *     x:   input values (== npset * [number of function parameters])
 *     y:   output values (== npset)
 */
class VLFunc {
public:
  virtual ~VLFunc() {}
  virtual int operator()(int npset, const float *x, float *restrict y) = 0;

  // Extension point (not in the reference): a functor that has a device implementation fills in an
  // mcx_vlfunc and returns true; MCPar::run then keeps the whole step on the GPU.  The default is
  // the host-callback path: proposals are copied out, operator() runs on the caller's thread,
  // log-likelihoods are copied back -- any reference-style subclass works unmodified.
  virtual bool device_descriptor(int /*np*/, mcx_vlfunc * /*out*/) const { return false; }
};

// A likelihood supplied as a GPU kernel from the user's own code object (MCX_VL_DEVICE):
//   extern "C" __global__ void f(int npset, const float *x, float *y);
// kernel = the hipFunction_t from hipModuleGetFunction.  Keeps the whole step on the device.
class DeviceVLFunc : public VLFunc {
  const int n;
  void *kernel;
public:
  DeviceVLFunc(int np, void *hip_function) : n(np), kernel(hip_function) {}
  bool device_descriptor(int, mcx_vlfunc *o) const
  {
    *o = mcx_vlfunc{MCX_VL_DEVICE, n, 0, 0, 0, kernel};
    return true;
  }
  int operator()(int npset, const float *x, float *restrict y)
  {
    mcx_vlfunc f;
    device_descriptor(n, &f);
    return mcx_vlfunc_eval(&f, npset, x, y) == MCX_OK ? 0 : 1;
  }
};

// A likelihood supplied as HIP SOURCE of device functions (MCX_VL_SOURCE, include/mcx.h): the engine compiles it into its
// own fused step kernels at run time -- one launch per segment of steps, chain state in registers, like the built-in
// functors (a separately compiled DeviceVLFunc kernel costs three launches per step).  The text defines
//   __device__ float mcx_user_loglike(const float *x, int d, const float *par);            // one parameter set
// or the per-block form described in mcx.h.  `params` (np_par floats) arrive as `par`.  The text and the parameters
// are copied; the object may be used for any number of runs.
#include <string>
#include <vector>
class SourceVLFunc : public VLFunc {
  const int n;
  const std::string text;
  const std::vector<float> par;
public:
  SourceVLFunc(int np, const char *hip_source, const float *params = 0, int np_par = 0)
      : n(np), text(hip_source ? hip_source : ""), par(params, params + (params ? np_par : 0)) {}
  bool device_descriptor(int, mcx_vlfunc *o) const
  {
    *o = mcx_vlfunc{MCX_VL_SOURCE, n, (int)par.size(), par.empty() ? 0 : par.data(), 0,
                    const_cast<char *>(text.c_str())};
    return true;
  }
  int operator()(int npset, const float *x, float *restrict y)
  {
    mcx_vlfunc f;
    device_descriptor(n, &f);
    return mcx_vlfunc_eval(&f, npset, x, y) == MCX_OK ? 0 : 1;
  }
};

#endif
