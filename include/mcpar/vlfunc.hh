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

#endif
