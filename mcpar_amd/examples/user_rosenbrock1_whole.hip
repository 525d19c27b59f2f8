// A user's likelihood as SOURCE (MCX_VL_SOURCE, include/mcx.h), whole-vector form -- VLFunc::operator() of
// src/vlfunc.hh:9-12 for one parameter set: Rosenbrock1 of src/rosenbrock.cc:4-21 with the sums taken in the
// engine's order (blocks of four left to right, block partials by an xor-butterfly over the block index, DESIGN.md
// section 3) so that the result equals the built-in's bit for bit; `par[0]` scales nothing (1.0), it only proves that
// the parameter block arrives.
__device__ float mcx_user_loglike(const float *x, int d, const float *par)
{
  float part[64];
  int nb = (d + 3) / 4, lpc = 1;
  while (lpc < nb) lpc <<= 1;
  for (int q = 0; q < lpc; ++q) {
    float acc = 0.0f;
    for (int k = 4 * q; k + 1 < d && k < 4 * q + 4; k += 2) {
      const float t1 = 1.0f - x[k];
      const float t2 = __builtin_fmaf(-x[k], x[k], x[k + 1]);
      acc = acc + __builtin_fmaf(100.0f * t2, t2, t1 * t1);
    }
    part[q] = acc;
  }
  for (int s = 1; s < lpc; s <<= 1)  // xor-butterfly: after step s every index holds the sum of its 2s-group
    for (int q = 0; q < lpc; q += 2 * s)
      for (int r = 0; r < s; ++r) {
        const float a = part[q + r], b = part[q + r + s];
        part[q + r] = a + b;
        part[q + r + s] = b + a;
      }
  return (0.0f - part[0]) * par[0];
}
