// mcx_cull_proj.hpp -- exact exclusion of far Gaussians along ONE direction (round 4; MCX_OPT_CULL = 2, never chosen
// automatically), for chain clouds stretched along a line that is diagonal to every coordinate axis, where the boxes of
// mcx_remote.hpp -- four coordinates wide -- see nothing.  Written for the 32-D mixture of BASELINE's C5 after
// tools/murray_pair_screen_probe.py had shown how many (128 neighbouring chains, Q_i) rows hold no pair with arg <= 176
// early in a run (96 % after 30 main-loop steps, 59 % after 50, 21 % after 70, 6 % after 100) -- and measured useless
// there: it keeps 0.999 of the pairs, like the boxes.  Those rows are dead because each chain is ~20 sigma away from a
// still-narrow Gaussian in the 31 directions ACROSS the mixture's axis; along the axis -- the only direction in which
// 128 chains can be neighbours -- the components overlap.  On C3's Rosenbrock shape it keeps 0.58 of the pairs where the
// boxes keep 0.41.  Kept as an option because it is exact, tested, and the right screen for a target that IS a line.
//
// The bound.  For any direction e, Cauchy-Schwarz on the sweep's own sum (src/mcpar.cc:367-373):
//     (sum_k e_k (mu_k - x_k))^2  <=  (sum_k (mu_k - x_k)^2 w_k) (sum_k e_k^2 / w_k)     i.e.  arg >= (e.mu - e.x)^2 / s,
// s = sum_k e_k^2 sigma_k^2.  With the active chains sorted by p = e.x, a group of CULL_W neighbours holds p in a short
// interval [lo, hi], and for every chain of the group arg(chain, Q_i) >= dist(e.mu_i, [lo, hi])^2 / s_i =: B.
// That is an inequality between real numbers; the sweep's float sum is related to them by an error bound, not by
// monotone rounding as in mcx_remote.hpp: all its terms are >= 0, so arg_float >= arg_real (1 - 2^-24)^(d+3)
// > arg_real (1 - 3e-6) for d <= 32.  p, s and B are formed in double from the same float inputs (relative error
// 1e-15).  Hence B (1 - 1e-4) > limit  =>  arg_float > limit for all chains of the group: the row's Q_i is exactly 0
// in their sums (limit 176), or cannot lower their minimum (limit = the group's largest own-Gaussian arg).  A NaN or
// infinite input makes B NaN and the comparison false: nothing is excluded.  The direction is whatever two power
// iterations on the active chains' covariance give (any direction is valid; a good one is merely useful).
// Results do not depend on it: tests/test_gpu_murray_cull.py runs the oracle -- which knows nothing of this -- against it.
#pragma once
#include "mcx_remote.hpp"

namespace mcx {

constexpr int PROJ_ACC = 2 * 32 + 2;  // per iteration: T[32], S1[32], sum of p, spare  (np <= 32)
constexpr double PROJ_SLACK = 1.0e-4;

// direction from an accumulator block: v = T - S1 (sum p) / n, normalised; the all-ones diagonal when it is degenerate
template <int DMAX>
__device__ __forceinline__ void proj_direction(const double *__restrict__ acc, int have, int nact, double e[DMAX], double *len)
{
  double nrm = 0.0;
  if (have) {
    const double sp = acc[2 * 32] / (double)nact;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
      e[k] = acc[k] - acc[32 + k] * sp;
      nrm += e[k] * e[k];
    }
  }
  if (!have || !(nrm > 0.0) || !(nrm < 1.0e300)) {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) e[k] = 1.0;
    nrm = (double)DMAX;
  }
  const double inv = 1.0 / __builtin_sqrt(nrm);
#pragma unroll
  for (int k = 0; k < DMAX; ++k) e[k] *= inv;
  if (len) *len = __builtin_sqrt(nrm);
}

// one power iteration: acc_out += (sum_j x_j (x_j . v), sum_j x_j, sum_j x_j . v) over the active chains, v from acc_in
template <int DMAX>
__global__ __launch_bounds__(BLOCK) void k_proj_moments(const float *__restrict__ xrows, const int *__restrict__ active, int nact,
                                                        const double *__restrict__ acc_in, int have_in, double *__restrict__ acc_out)
{
  double e[DMAX];
  proj_direction<DMAX>(acc_in, have_in, nact, e, nullptr);
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  double x[DMAX], p = 0.0;
#pragma unroll
  for (int k = 0; k < DMAX; ++k) x[k] = 0.0;
  if (i < nact) {
    const float *xr = xrows + (size_t)(active ? active[i] : i) * DMAX;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
      x[k] = (double)xr[k];
      p += e[k] * x[k];
    }
    if (!(p == p) || !(p < 1.0e300 && p > -1.0e300)) {  // a chain at NaN / infinity: left out of the statistics
      p = 0.0;
#pragma unroll
      for (int k = 0; k < DMAX; ++k) x[k] = 0.0;
    }
  }
  __shared__ double part[BLOCK / 64][PROJ_ACC];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < DMAX; ++k) {
    double a = x[k] * p, b = x[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
    }
    if (lane == 0) {
      part[wv][k] = a;
      part[wv][32 + k] = b;
    }
  }
  {
    double a = p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) part[wv][64] = a;
  }
  __syncthreads();
  if (threadIdx.x < 65 && (threadIdx.x < (unsigned)DMAX || (threadIdx.x >= 32 && threadIdx.x < 32u + DMAX) || threadIdx.x == 64)) {
    double t = 0.0;  // (same-address atomics are ~10 ns each: one per workgroup and sum)
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) t += part[w][threadIdx.x];
    atomicAdd(acc_out + threadIdx.x, t);
  }
}

// p = e . x of every active chain (by chain index) and its bin along the line: CULL_BINS bins over mean +- 3 sd
template <int DMAX>
__global__ __launch_bounds__(BLOCK) void k_proj_keys(const float *__restrict__ xrows, const int *__restrict__ active, int nact,
                                                     const double *__restrict__ acc, double *__restrict__ pj,
                                                     unsigned *__restrict__ keys, unsigned *__restrict__ hist)
{
  double e[DMAX], len;
  proj_direction<DMAX>(acc, 1, nact, e, &len);
  double mp = 0.0;
#pragma unroll
  for (int k = 0; k < DMAX; ++k) mp += e[k] * acc[32 + k];
  mp /= (double)nact;
  // (the last iteration's |C v| n is a fine estimate of n times the variance along v)
  const double sd = __builtin_sqrt(len / (double)nact > 1e-300 ? len / (double)nact : 1e-300);
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= nact) return;
  const int j = active ? active[i] : i;
  const float *xr = xrows + (size_t)j * DMAX;
  double p = 0.0;
#pragma unroll
  for (int k = 0; k < DMAX; ++k) p += e[k] * (double)xr[k];
  pj[j] = p;
  const double t = (p - (mp - 3.0 * sd)) * ((double)CULL_BINS / (6.0 * sd));
  const unsigned key = (unsigned)(t > 0.0 ? (t < (double)(CULL_BINS - 1) ? (int)t : CULL_BINS - 1) : 0);  // NaN -> 0
  keys[i] = key;
  atomicAdd(hist + key, 1u);
}

// [lo, hi] of p over every group of CULL_W consecutive positions of the sorted list, and the group's limit (as k_cull_boxes)
template <int DMAX, bool SUMS>
__global__ __launch_bounds__(BLOCK) void k_proj_groups(const float *__restrict__ xrows, const int *__restrict__ order, int nact,
                                                       const float *__restrict__ qpar, int own0, const double *__restrict__ pj,
                                                       double *__restrict__ lohi, float *__restrict__ lim,
                                                       double *__restrict__ acc_done, unsigned *__restrict__ hist_done)
{
  // the sort is over: the power iterations' first accumulator block and the histogram are left zero for the next call
  for (int i = (int)(blockIdx.x * BLOCK + threadIdx.x); i < CULL_BINS; i += (int)(gridDim.x * BLOCK)) hist_done[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x < PROJ_ACC) acc_done[threadIdx.x] = 0.0;
  const int g = (int)blockIdx.x * (BLOCK / 64) + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  if (g * CULL_W >= nact) return;
  double lo = __builtin_inf(), hi = -__builtin_inf();
  bool bad = false;
  float worst = 0.0f;  // !SUMS: the largest own-Gaussian arg of the group
#pragma unroll
  for (int h = 0; h < CULL_W / 64; ++h) {
    const int pos = g * CULL_W + h * 64 + lane;
    if (pos < nact) {
      const int j = order[pos];
      const double p = pj[j];
      bad = bad || !(p == p);
      lo = p < lo ? p : lo;
      hi = p > hi ? p : hi;
      if (!SUMS) {
        const float *x = xrows + (size_t)j * DMAX;
        const float *qo = qpar + 2 * (size_t)(own0 + j) * DMAX;
        float a0 = 0.0f;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
          const float xm = qo[2 * k] - x[k];
          a0 = __builtin_fmaf(xm * xm, qo[2 * k + 1], a0);
        }
        worst = (a0 < __builtin_inff()) ? (a0 > worst ? a0 : worst) : __builtin_inff();
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
    if (!SUMS) {
      const float w2 = __shfl_xor(worst, o);
      worst = w2 > worst ? w2 : worst;
    }
  }
  bad = __ballot(bad) != 0ull;
  if (lane == 0) {
    // (a chain without a position on the line -- NaN -- makes its group's interval the whole line: nothing excluded)
    lohi[2 * (size_t)g] = bad ? -__builtin_inf() : lo;
    lohi[2 * (size_t)g + 1] = bad ? __builtin_inf() : hi;
    lim[g] = SUMS ? ZERO_ARG : (worst < ZERO_ARG ? worst : ZERO_ARG);
  }
}

// excl[w][g] bit b = Q_{64 w + b} may matter to group g (the layout of k_cull_test): one Q_i per lane, its position on the
// line and its spread along it in registers, the wavefront walks over a chunk of the groups
template <int DMAX>
__global__ __launch_bounds__(BLOCK) void k_proj_test(const float *__restrict__ qpar, int N, const double *__restrict__ acc,
                                                     const double *__restrict__ lohi, const float *__restrict__ lim, int ngroups,
                                                     int nact, int gchunk, unsigned long long *__restrict__ excl, int excl_words,
                                                     unsigned long long *__restrict__ nkept)
{
  const int w = (int)blockIdx.x * (BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // mask word
  const int i = w * 64 + (int)(threadIdx.x & 63u);
  if (w >= excl_words) return;
  const bool have = i < N;
  double e[DMAX];
  proj_direction<DMAX>(acc, 1, nact, e, nullptr);
  double pm = 0.0, s = 0.0;
  const float2 *src = reinterpret_cast<const float2 *>(qpar + 2 * (size_t)(have ? i : 0) * DMAX);
#pragma unroll
  for (int k = 0; k < DMAX; ++k) {
    const float2 v = src[k];  // (mu, 1 / sigma^2)
    pm += e[k] * (double)v.x;
    s += e[k] * e[k] / (double)v.y;
  }
  const double inv_s = (1.0 - PROJ_SLACK) / s;  // (s = 0, inf or NaN: the bound below is inf -- right for s = 0 -- or NaN -> not excluded)
  const int g0 = (int)blockIdx.y * gchunk, g1 = g0 + gchunk < ngroups ? g0 + gchunk : ngroups;
  unsigned long long kept = 0;
  for (int g = g0; g < g1; ++g) {
    const double lo = lohi[2 * (size_t)g], hi = lohi[2 * (size_t)g + 1];
    const double a = lo - pm, b = pm - hi;
    const double dist = a > b ? (a > 0.0 ? a : 0.0) : (b > 0.0 ? b : 0.0);
    const double bound = dist > 0.0 ? dist * dist * inv_s : 0.0;  // (inside the interval: 0, whatever s is)
    const unsigned long long m = __ballot(have && !(bound > (double)lim[g]));  // (a NaN bound excludes nothing)
    if ((threadIdx.x & 63u) == 0) excl[(size_t)w * ngroups + g] = m;
    const int members = nact - g * CULL_W < CULL_W ? nact - g * CULL_W : CULL_W;
    kept += (unsigned long long)__popcll(m) * (unsigned long long)members;
  }
  if ((threadIdx.x & 63u) == 0 && kept) atomicAdd(nkept + ((blockIdx.x + blockIdx.y) & (CULL_NCOUNT - 1)), kept);
}

}  // namespace mcx
