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

#endif
