// rosenbrock.hh -- the reference's built-in likelihoods (src/rosenbrock.hh:1-64) as device-backed
// functors.  operator() evaluates on the GPU through mcx_vlfunc_eval; inside MCPar::run the
// functor is fused into the step kernels via device_descriptor().
#ifndef MCPAR_AMD_ROSENBROCK_HH_
#define MCPAR_AMD_ROSENBROCK_HH_

#include <vector>

#include "vlfunc.hh"

namespace mcpar_detail {
inline int eval_builtin(const mcx_vlfunc &f, int npset, const float *x, float *y)
{
  return mcx_vlfunc_eval(&f, npset, x, y) == MCX_OK ? 0 : 1;
}
}  // namespace mcpar_detail

/* Rosenbrock function with non-overlapping components (src/rosenbrock.hh:6-18) */
class Rosenbrock1 : public VLFunc {
  const int n;
public:
  Rosenbrock1(int nc) : n(nc)
  {
    if (n < 2 || n % 2 != 0) throw("N for Rosenbrock1 must be even and >= 2");
  }
  bool device_descriptor(int, mcx_vlfunc *o) const
  {
    *o = mcx_vlfunc{MCX_VL_ROSENBROCK1, n, 0, 0, 0, 0};
    return true;
  }
  int operator()(int npset, const float *x, float *restrict fx)
  {
    mcx_vlfunc f;
    device_descriptor(n, &f);
    return mcpar_detail::eval_builtin(f, npset, x, fx);
  }
};

/* Rosenbrock function with overlapping components, exactly as the reference evaluates it
 * (src/rosenbrock.cc:25-41: '-' on the second term, x[i+1] read across the set boundary) */
class Rosenbrock2 : public VLFunc {
  const int n;
public:
  Rosenbrock2(int nc) : n(nc)
  {
    if (n < 2) throw("N for Rosenbrock2 must be >= 2");
  }
  bool device_descriptor(int, mcx_vlfunc *o) const
  {
    *o = mcx_vlfunc{MCX_VL_ROSENBROCK2, n, 0, 0, 0, 0};
    return true;
  }
  int operator()(int npset, const float *x, float *restrict fx)
  {
    mcx_vlfunc f;
    device_descriptor(n, &f);
    return mcpar_detail::eval_builtin(f, npset, x, fx);
  }
};

/* NOT in the reference -- a flagged variant: the overlapping Rosenbrock function made well-posed.  The reference's
 * Rosenbrock2 subtracts its second term (src/rosenbrock.cc:38; compare the '+' at :18), which leaves log L unbounded
 * above, and its flat loop reads x[i+1] across the boundary of a parameter set (:32-35).  This one is
 * - sum_{k < n-1} (1 - x_k)^2 + 100 (x_{k+1} - x_k^2)^2 per set. */
class Rosenbrock2Fixed : public VLFunc {
  const int n;
public:
  Rosenbrock2Fixed(int nc) : n(nc)
  {
    if (n < 2) throw("N for Rosenbrock2 must be >= 2");
  }
  bool device_descriptor(int, mcx_vlfunc *o) const
  {
    *o = mcx_vlfunc{MCX_VL_ROSENBROCK2_FIXED, n, 0, 0, 0, 0};
    return true;
  }
  int operator()(int npset, const float *x, float *restrict fx)
  {
    mcx_vlfunc f;
    device_descriptor(n, &f);
    return mcpar_detail::eval_builtin(f, npset, x, fx);
  }
};

/* Diagonal Gaussian (src/rosenbrock.hh:35-50).  The reference insists on N == 2; any N works here. */
class Gaussian : public VLFunc {
  const int n;
  std::vector<float> par;  // mu[n], sig2[n]
public:
  Gaussian(int nc, const float muin[] = 0, const float sig2[] = 0) : n(nc), par(2 * (nc > 0 ? nc : 1))
  {
    if (nc < 1) throw("Invalid specification.  N for Gaussian must be >= 1.");
    for (int i = 0; i < n; ++i) {
      par[i] = muin ? muin[i] : 0.0f;
      par[n + i] = sig2 ? sig2[i] : 1.0f;
    }
  }
  bool device_descriptor(int, mcx_vlfunc *o) const
  {
    *o = mcx_vlfunc{MCX_VL_GAUSSIAN, n, 0, par.data(), 0, 0};
    return true;
  }
  int operator()(int npset, const float *x, float *restrict fx)
  {
    mcx_vlfunc f;
    device_descriptor(n, &f);
    return mcpar_detail::eval_builtin(f, npset, x, fx);
  }
};

/* sum of two unit Gaussians at (0,0) and (5,5), weight w on the first (src/rosenbrock.hh:52-62) */
class DualGaussian : public VLFunc {
  const float w;
public:
  DualGaussian(float win) : w(win) {}
  bool device_descriptor(int, mcx_vlfunc *o) const
  {
    *o = mcx_vlfunc{MCX_VL_DUALGAUSS, 2, 0, &w, 0, 0};
    return true;
  }
  int operator()(int npset, const float *x, float *restrict fx)
  {
    mcx_vlfunc f;
    device_descriptor(2, &f);
    return mcpar_detail::eval_builtin(f, npset, x, fx);
  }
};

/* N-D mixture of K unit-variance Gaussians (BASELINE config 5; no counterpart in the reference).
 * means[K*n] row-major, weights[K]. */
class GaussianMixture : public VLFunc {
  const int n, K;
  std::vector<float> par;
public:
  GaussianMixture(int nc, int ncomp, const float *means, const float *weights)
      : n(nc), K(ncomp), par((size_t)ncomp * nc + ncomp)
  {
    if (nc < 1 || ncomp < 1 || ncomp > 64) throw("Invalid specification for GaussianMixture.");
    for (int i = 0; i < K * n; ++i) par[i] = means[i];
    for (int c = 0; c < K; ++c) par[(size_t)K * n + c] = weights[c];
  }
  bool device_descriptor(int, mcx_vlfunc *o) const
  {
    *o = mcx_vlfunc{MCX_VL_GAUSSMIX, n, K, par.data(), 0, 0};
    return true;
  }
  int operator()(int npset, const float *x, float *restrict fx)
  {
    mcx_vlfunc f;
    device_descriptor(n, &f);
    return mcpar_detail::eval_builtin(f, npset, x, fx);
  }
};

#endif
