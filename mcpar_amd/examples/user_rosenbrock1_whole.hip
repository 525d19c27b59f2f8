// A user's likelihood as SOURCE (MCX_VL_SOURCE, include/mcx.h), whole-vector form -- VLFunc::operator() of
// src/vlfunc.hh:9-12 for one parameter set: Rosenbrock1 of src/rosenbrock.cc:4-21 with the sums taken in the
// engine's order (blocks of four left to right, block partials by an xor-butterfly over the block index, DESIGN.md
// section 3) so that the result equals the built-in's bit for bit; `par[0]` scales nothing (1.0), it only proves that
// the parameter block arrives.  MCX_USER_NP (= np) and MCX_USER_LPC (= blocks of four, rounded up to a power of two) are
// defined when the text is compiled: loops over them unroll and `part` stays in registers.
__device__ float mcx_user_loglike(const float *x, int d, const float *par)
{
  constexpr int NB = MCX_USER_LPC;
  float part[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 4 * q; k < 4 * q + 4; k += 2)
      if (k + 1 < MCX_USER_NP) {
        const float t1 = 1.0f - x[k];
        const float t2 = __builtin_fmaf(-x[k], x[k], x[k + 1]);
        acc = acc + __builtin_fmaf(100.0f * t2, t2, t1 * t1);
      }
    part[q] = acc;
  }
#pragma unroll
  for (int s = 1; s < NB; s <<= 1)  // the butterfly as seen from block 0: part[q] += part[q + s] for q a multiple of 2 s
#pragma unroll
    for (int q = 0; q < NB; q += 2 * s) part[q] = part[q] + part[q + s];
  return (0.0f - part[0]) * par[0];
}
